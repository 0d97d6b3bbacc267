// mfma_valu_rate.hip -- can the f32 matrix pipe (v_mfma_f32_32x32x2_f32, exact f32) and the f32 VALU run
// at the same time on one SIMD?  512-thread blocks, one per CU: waves 0-3 (one per SIMD) issue MFMAs,
// waves 4-7 (their SIMD partners) issue v_fma_f32.  T(both) ~ max(T(mfma), T(valu)) means the pipes overlap;
// ~ sum means they serialise.  Design input for the ray x triangle scan.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) mix_kernel(float *out, int mfma_iters, int valu_iters, float seed)
{
    const int wave = threadIdx.x >> 6;
    float res = 0;
    if (wave < 4) {
        f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        float a = seed + threadIdx.x, b = seed * 0.5f + 1.0f;
        for (int it = 0; it < mfma_iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) res += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    } else {
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
        float m = seed * 0.5f + 1.0f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = __builtin_fmaf(v[i], m, 0.25f);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) res += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

static float run(float *out, int grid, int mi, int vi)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(512), 0, 0, out, mi, vi, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    float *out; (void)hipMalloc(&out, 512 * 4096 * sizeof(float));
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = prop.multiProcessorCount * bpc;
        const int mi = 20000;                       // 4 MFMA x 64 cycles = 256 cycles per iteration
        for (int vi : {0, 10000, 20000, 27000, 40000}) {      // 48 fma per iteration (192 cycles alone at 4 cyc/instr)
            float tm = run(out, grid, mi, 0), tv = vi ? run(out, grid, 0, vi) : 0.f, tb = run(out, grid, mi, vi);
            double mf = (double)grid * 4 * mi * 4.0 * 4096, vf = (double)grid * 4 * vi * 48.0 * 128;
            printf("blocks/CU %d  valu_iters %5d: mfma-only %.3f ms (%.0f TF)  valu-only %.3f ms (%.0f TF)  both %.3f ms (%.0f TF total)  overlap: both/max = %.2f, both/sum = %.2f\n",
                   bpc, vi, tm, mf / tm / 1e9, tv, tv > 0 ? vf / tv / 1e9 : 0.0, tb, (mf + vf) / tb / 1e9, tb / (tm > tv ? tm : tv), tb / (tm + tv));
        }
    }
    return 0;
}
