// mfma_hazard_probe.hip -- how many wait states does v_mfma_f32_32x32x16_bf16 need before a VALU instruction may read its
// result registers?  One asm block: pre-fill the 16 result registers with a marker, issue the MFMA (result = known dot
// products, never the marker), s_nop N, then copy result register 15 (written last) and register 0.  A lane that still
// sees the marker read too early.  Run with 1..4 waves per SIMD (the matrix pipe is shared between co-resident waves).
// (A third series -- VALU reads of the result registers directly followed by the next MFMA that overwrites them, the
// pattern rt_mfma.hpp had when it lost hits -- was tried here too: the identical asm block produced tens of thousands of
// wrong lanes in one build of this file and none in another whose surrounding C++ differed, so it proves nothing and was
// removed.  The scan's fix is validated on the scan itself: scripts/dbg_cand.py, scripts/dbg_soak.py, the regression test.)
// Second series: NV independent VALU instructions + s_nop, the mix the compiler's hazard recognizer produces when it counts
// every VALU instruction in between as one wait state ("v_max3, v_max3, v_cmp, s_nop 8" = 12 by its count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__global__ void __launch_bounds__(1024) probe(uint32_t *bad_lanes, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)1.0f; }     // every output = 16
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r15;
        asm volatile(
            "v_mov_b32 v32, 0x7fc00000\n\tv_mov_b32 v33, v32\n\tv_mov_b32 v34, v32\n\tv_mov_b32 v35, v32\n\t"
            "v_mov_b32 v36, v32\n\tv_mov_b32 v37, v32\n\tv_mov_b32 v38, v32\n\tv_mov_b32 v39, v32\n\t"
            "v_mov_b32 v40, v32\n\tv_mov_b32 v41, v32\n\tv_mov_b32 v42, v32\n\tv_mov_b32 v43, v32\n\t"
            "v_mov_b32 v44, v32\n\tv_mov_b32 v45, v32\n\tv_mov_b32 v46, v32\n\tv_mov_b32 v47, v32\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            "v_mfma_f32_32x32x16_bf16 v[32:47], %2, %3, 0\n\t"
            "s_nop %4\n\t"
            "v_mov_b32 %1, v47\n\t"
            "v_mov_b32 %0, v32\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            : "=v"(r0), "=v"(r15) : "v"(a), "v"(b), "n"(N)
            : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        if (r15 != 16.0f) bad |= 1u;
        if (r0 != 16.0f) bad |= 2u;
    }
    if (bad) atomicOr(&bad_lanes[(threadIdx.x & 63) >> 4], bad);       // per 16-lane quarter of the wave
    if (bad) atomicAdd(&bad_lanes[4], 1u);
}

template <int NV, int N>
__global__ void __launch_bounds__(1024) probe_valu(uint32_t *bad_lanes, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)1.0f; }
    uint32_t bad = 0;
    float x = threadIdx.x, y = 1.0f, z = 2.0f;
    for (int it = 0; it < iters; ++it) {
        float r0, r15;
        asm volatile(
            "v_mov_b32 v32, 0x7fc00000\n\tv_mov_b32 v33, v32\n\tv_mov_b32 v34, v32\n\tv_mov_b32 v35, v32\n\t"
            "v_mov_b32 v36, v32\n\tv_mov_b32 v37, v32\n\tv_mov_b32 v38, v32\n\tv_mov_b32 v39, v32\n\t"
            "v_mov_b32 v40, v32\n\tv_mov_b32 v41, v32\n\tv_mov_b32 v42, v32\n\tv_mov_b32 v43, v32\n\t"
            "v_mov_b32 v44, v32\n\tv_mov_b32 v45, v32\n\tv_mov_b32 v46, v32\n\tv_mov_b32 v47, v32\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            "v_mfma_f32_32x32x16_bf16 v[32:47], %2, %3, 0\n\t"
            ".rept %5\n\tv_max3_f32 %6, %6, %7, %8\n\t.endr\n\t"
            "s_nop %4\n\t"
            "v_mov_b32 %1, v47\n\t"
            "v_mov_b32 %0, v32\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            : "=v"(r0), "=v"(r15) : "v"(a), "v"(b), "n"(N), "n"(NV), "v"(x), "v"(y), "v"(z)
            : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        if (r15 != 16.0f) bad |= 1u;
        if (r0 != 16.0f) bad |= 2u;
    }
    if (bad) atomicOr(&bad_lanes[(threadIdx.x & 63) >> 4], bad);
    if (bad) atomicAdd(&bad_lanes[4], 1u);
}

template <int NV, int N> static void run_valu(uint32_t *d, int cus)
{
    for (int wps = 1; wps <= 4; ++wps) {
        (void)hipMemset(d, 0, 32);
        hipLaunchKernelGGL((probe_valu<NV, N>), dim3(cus), dim3(256 * wps), 0, 0, d, 20000);
        uint32_t h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("%2d x v_max3 + s_nop %2d (= %2d by the compiler's count)  waves/SIMD %d: lanes reading stale data %u  quarters [%x %x %x %x]\n",
               NV, N, NV + N + 1, wps, h[4], h[0], h[1], h[2], h[3]);
    }
}

template <int N> static void run(uint32_t *d, int cus)
{
    for (int wps = 1; wps <= 4; ++wps) {
        (void)hipMemset(d, 0, 32);
        hipLaunchKernelGGL(probe<N>, dim3(cus), dim3(256 * wps), 0, 0, d, 20000);
        uint32_t h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("s_nop %2d (%2d wait states)  waves/SIMD %d: lanes reading stale data %u  quarters [0-15 %x] [16-31 %x] [32-47 %x] [48-63 %x] (bit0: last register, bit1: first)\n",
               N, N + 1, wps, h[4], h[0], h[1], h[2], h[3]);
    }
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    uint32_t *d; (void)hipMalloc(&d, 32);
    const int cus = prop.multiProcessorCount;
    run<9>(d, cus); run<10>(d, cus); run<11>(d, cus); run<12>(d, cus);
    run_valu<3, 8>(d, cus); run_valu<6, 5>(d, cus); run_valu<11, 0>(d, cus); run_valu<3, 9>(d, cus); run_valu<3, 10>(d, cus); run_valu<3, 11>(d, cus);
    run_valu<6, 8>(d, cus); run_valu<6, 11>(d, cus); run_valu<12, 5>(d, cus); run_valu<12, 11>(d, cus);
    return 0;
}
