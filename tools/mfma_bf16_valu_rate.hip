// mfma_bf16_valu_rate.hip -- how the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) and the f32 VALU share a SIMD.
//  A: waves 0-3 of a 512-thread block issue MFMAs, waves 4-7 (their SIMD partners) issue v_min3_f32:
//     T(both) ~ max => the pipes overlap across waves, ~ sum => they serialise.
//  B: the broad-phase pattern of rt_mfma.hpp in one wave -- MFMA, then 5 v_min3 + 2 v_max3 on its result -- with
//     1..4 waves per SIMD: time per MFMA per SIMD is what the scan can reach at best.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) mix_kernel(float *out, int mfma_iters, int valu_iters, float seed)
{
    const int wave = threadIdx.x >> 6;
    float res = 0;
    if (wave < 4) {
        f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x + i); b[i] = (__bf16)(seed * 0.5f + i); }
        for (int it = 0; it < mfma_iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
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
                for (int i = 0; i < 12; ++i) v[i] = __builtin_fminf(__builtin_fminf(v[i], m), v[(i + 1) % 12] + 0.0f);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) res += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int PAIR>
__global__ void __launch_bounds__(1024) pattern_kernel(float *out, int iters, float seed)
{
    bf16x8 a, b0, b1;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x + i); b0[i] = (__bf16)(seed * 0.5f + i); b1[i] = (__bf16)(seed * 0.25f + i); }
    float res = -1e30f;
    const f16v zero = {0};
    for (int it = 0; it < iters; ++it) {
        f16v acc[2];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, zero, 0, 0, 0);
        if (PAIR) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, zero, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 1 + PAIR; ++s) {
            float mn[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) mn[u] = __builtin_fminf(__builtin_fminf(acc[s][3 * u], acc[s][3 * u + 1]), acc[s][3 * u + 2]);
            res = __builtin_fmaxf(res, __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mn[0], mn[1]), mn[2]), mn[3]), mn[4]));
        }
        // keep the operands changing so that nothing is hoisted
        a[0] = (__bf16)res;
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

template <int PAIR>
static float run_pattern(float *out, int grid, int threads, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(pattern_kernel<PAIR>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    float *out; (void)hipMalloc(&out, 1024 * 4096 * sizeof(float));
    const int cus = prop.multiProcessorCount;
    printf("A: bf16 MFMA waves + v_min3 waves on the same SIMDs (%d CUs)\n", cus);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = cus * bpc;
        const int mi = 20000;                       // 4 MFMA per iteration
        for (int vi : {0, 5000, 10000, 20000}) {    // 48 x 2 VALU per iteration
            float tm = run(out, grid, mi, 0), tv = vi ? run(out, grid, 0, vi) : 0.f, tb = run(out, grid, mi, vi);
            printf("blocks/CU %d  valu_iters %5d: mfma-only %.3f ms (%.1f ns/MFMA/SIMD)  valu-only %.3f ms  both %.3f ms  both/max = %.2f, both/sum = %.2f\n",
                   bpc, vi, tm, tm * 1e6 / (mi * 4.0 * bpc), tv, tb, tb / (tm > tv ? tm : tv), tb / (tm + tv));
        }
    }
    printf("B: MFMA -> 5 v_min3 + 2 v_max3 on its result, per wave; ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n");
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {
        float t1 = run_pattern<0>(out, cus, 256 * wps, iters), t2 = run_pattern<1>(out, cus, 256 * wps, iters);
        printf("waves/SIMD %d: single %.3f ms = %.1f ns/MFMA/SIMD   paired %.3f ms = %.1f ns/MFMA/SIMD\n",
               wps, t1, t1 * 1e6 / ((double)iters * wps), t2, t2 * 1e6 / ((double)iters * 2 * wps));
    }
    return 0;
}
