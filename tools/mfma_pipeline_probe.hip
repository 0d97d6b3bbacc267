// mfma_pipeline_probe.hip -- the scan's tile loop with known operands, to look for the lost-survivor fault of rt_mfma.hpp
// outside the path tracer.  Every product is A(stage) x B with all entries of B = 1 and all entries of A = 1 (even stages) or
// 1 + (stage & 3), K = 16, so every accumulator entry of a stage is 16, 32, 48 or 64 and differs from what its set held before.  Two accumulator sets alternate exactly as in
// the kernel; the examination (5 v_min3 + 2 v_max3 per ray set) of the set produced one stage earlier must return the value
// of THAT stage (a stale set would show the value of three stages earlier).  Modes: 0 = shipped order (both products, gap, examination), 1 = interleaved (product, examination of one
// ray set, product, examination of the other).  Counts lanes/iterations whose examination returned anything else.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int GAP>
__global__ void __launch_bounds__(256) pipeline(uint32_t *out, int iters)
{
    bf16x8 a1, a2, a3, a4, b0, b1;
    for (int i = 0; i < 8; ++i) { a1[i] = (__bf16)1.0f; a2[i] = (__bf16)2.0f; a3[i] = (__bf16)3.0f; a4[i] = (__bf16)4.0f; b0[i] = (__bf16)(float)(1 + (threadIdx.x & 31)); b1[i] = (__bf16)(float)(33 - (int)(threadIdx.x & 31)); }     // per-column values: a lane mix-up shows
    const f32x16 zero = {0};
    f32x16 X0, X1, Y0, Y1;
    uint32_t bad = 0, first_bad = 0xffffffffu;
    auto minmax = [&](const f32x16 &acc) {
        float mn[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) mn[u] = __builtin_fminf(__builtin_fminf(acc[3 * u], acc[3 * u + 1]), acc[3 * u + 2]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mn[0], mn[1]), mn[2]), mn[3]), mn[4]);
        const float lo = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fminf(mn[0], mn[1]), mn[2]), mn[3]), mn[4]);
        return (mx == lo) ? mx : -1.0f;
    };
    // stage(k): products with A = (k odd ? a2 : a1) into `nxt`; examination of `pend`, which must hold 16 * (1 + ((k-1) & 1))
    auto stage = [&](int k, f32x16 &n0, f32x16 &n1, f32x16 &p0, f32x16 &p1, bool examine) {
        const bf16x8 a = (k & 2) ? ((k & 1) ? a4 : a3) : ((k & 1) ? a2 : a1);      // value 1 + (k & 3): every stage differs from the last use of its set
        const float want = 16.0f * (float)(1 + ((k - 1) & 3));
        const float col0 = (float)(1 + (threadIdx.x & 31)), col1 = (float)(33 - (int)(threadIdx.x & 31));
        if (MODE == 0) {
            asm volatile("s_nop 1" ::: "memory");
            n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, zero, 0, 0, 0);
            n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, zero, 0, 0, 0);
            asm volatile("s_nop %4" : "+v"(X0), "+v"(X1), "+v"(Y0), "+v"(Y1) : "n"(GAP));
            if (examine) {
                const float r0 = minmax(p0), r1 = minmax(p1);
                if (r0 != want * col0 || r1 != want * col1) { bad++; if (first_bad == 0xffffffffu) first_bad = (uint32_t)k; }
            }
        } else {
            float r0 = want * col0, r1 = want * col1;
            n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, zero, 0, 0, 0);
            asm volatile("s_nop %4" : "+v"(X0), "+v"(X1), "+v"(Y0), "+v"(Y1) : "n"(GAP));
            if (examine) r0 = minmax(p0);
            asm volatile("" : "+v"(X0), "+v"(X1), "+v"(Y0), "+v"(Y1), "+v"(r0));
            n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, zero, 0, 0, 0);
            asm volatile("s_nop %4" : "+v"(X0), "+v"(X1), "+v"(Y0), "+v"(Y1) : "n"(GAP));
            if (examine) r1 = minmax(p1);
            if (r0 != want * col0 || r1 != want * col1) { bad++; if (first_bad == 0xffffffffu) first_bad = (uint32_t)k; }
        }
    };
    X0 = X1 = Y0 = Y1 = zero;
    stage(0, X0, X1, Y0, Y1, false);
    for (int k = 1; k + 1 < iters; k += 2) {
        stage(k, Y0, Y1, X0, X1, true);
        stage(k + 1, X0, X1, Y0, Y1, true);
    }
    if (bad) {
        atomicAdd(&out[0], 1u);                                   // lanes with at least one wrong examination
        atomicAdd(&out[1], bad);                                  // wrong examinations
        atomicAdd(&out[2 + ((threadIdx.x & 63) >> 4)], 1u);       // by 16-lane quarter of the wave
        atomicMin(&out[6], first_bad);
    }
}

template <int MODE, int GAP> static void run(uint32_t *d, int cus, int blocks_per_cu)
{
    uint32_t init[8] = {0, 0, 0, 0, 0, 0, 0xffffffffu, 0};
    (void)hipMemcpy(d, init, 32, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pipeline<MODE, GAP>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 40000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%s, s_nop %2d behind the products, %d waves/SIMD: %.1f ms, %.1f ns per product per SIMD; lanes with a wrong examination %u (wrong examinations %u; quarters %u %u %u %u; first at stage %d)\n",
           MODE ? "interleaved" : "shipped    ", GAP, blocks_per_cu, ms, ms * 1e6 / (40000.0 * 2 * blocks_per_cu), h[0], h[1], h[2], h[3], h[4], h[5], (int)h[6]);
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    uint32_t *d; (void)hipMalloc(&d, 32);
    const int cus = prop.multiProcessorCount;
    for (int w = 1; w <= 3; ++w) {
        run<0, 7>(d, cus, w); run<0, 0>(d, cus, w);
        run<1, 0>(d, cus, w); run<1, 3>(d, cus, w); run<1, 7>(d, cus, w); run<1, 11>(d, cus, w);
    }
    return 0;
}
