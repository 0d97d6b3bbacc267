// valu_rate.hip -- microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (design input
// for the triangle inner loop; results recorded in DESIGN.md).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float *out, int iters, float seed)
{
    float a[12];
    f2 p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 0.5f}; }
    float m = seed * 0.5f + 1.0f;
    f2 m2 = f2{m, m};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) a[i] = __builtin_fmaf(a[i], m, 0.25f);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) p[i] = __builtin_elementwise_fma(p[i], m2, f2{0.25f, 0.25f});
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += (MODE == 0) ? a[i] : (p[i].x + p[i].y);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float *out; hipMalloc(&out, 256 * 2048 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            int grid = prop.multiProcessorCount * blocks_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
                else hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    double inst = (double)grid * 4 /*waves*/ * iters * 48.0;   // wave-instructions
                    double lane_fma = inst * 64 * (mode ? 2 : 1);
                    printf("mode %s waves/SIMD %d: %.3f ms  %.2f TFLOP/s  wave-instr/s/SIMD %.3e (=> cycles/instr at 2.4GHz: %.2f)\n",
                           mode ? "v_pk_fma_f32" : "v_fma_f32  ", blocks_per_cu, ms, lane_fma * 2 / ms / 1e9,
                           inst / (ms * 1e-3) / (prop.multiProcessorCount * 4), 2.4e9 / (inst / (ms * 1e-3) / (prop.multiProcessorCount * 4)));
                }
            }
        }
    return 0;
}
