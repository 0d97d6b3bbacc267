// mfma_repro_probe.hip -- is v_mfma_f32_32x32x16_bf16 bit-reproducible from launch to launch on operands whose sum rounds?
// Pseudo-random bf16 operands (fixed seed) of mixed magnitude and sign, many products per wave, XOR/ADD checksum of all result
// bits; the same launch five times, 1 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

__global__ void __launch_bounds__(256) repro(unsigned long long *out, int iters)
{
    uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long chk = 0;
    const f32x16 zero = {0};
    for (int it = 0; it < iters; ++it) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) {
            const float m1 = (float)((int)(lcg(seed) >> 9) - (1 << 22)) * (1.0f / (1 << 14)), m2 = (float)((int)(lcg(seed) >> 9) - (1 << 22)) * (1.0f / (1 << 18));
            a[i] = (__bf16)(m1 * ((i & 1) ? 1.0f : 64.0f)); b[i] = (__bf16)(m2 * ((i & 2) ? 1.0f : 1024.0f));
        }
        f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
        for (int i = 0; i < 16; ++i) chk += (unsigned long long)__float_as_uint(c[i]) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned)i);
    }
    atomicAdd(out, chk);
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    unsigned long long *d; (void)hipMalloc(&d, 8);
    for (int wps = 1; wps <= 3; wps += 2)
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(d, 0, 8);
            hipLaunchKernelGGL(repro, dim3(prop.multiProcessorCount * wps), dim3(256), 0, 0, d, 4000);
            unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("waves/SIMD %d launch %d: checksum %016llx\n", wps, rep, h);
        }
    return 0;
}
