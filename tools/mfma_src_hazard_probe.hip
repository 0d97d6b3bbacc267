// mfma_src_hazard_probe.hip -- does v_mfma_f32_32x32x16_bf16 see a source operand (SrcA) that a VALU instruction wrote N wait states
// before it?  The compiler's hazard recognizer inserts what is needed for its own code, but it does not look into inline asm: the
// scan's trip statements (rt_scan.hpp) start with an MFMA, and the compiler may place its own v_mov of the A rows right in front
// of them (round 3: the list-driven tile loop lost hits until the statement got wait states at its head).
// Series: the A operand holds 1.0; a v_mov rewrites all four registers of it with 2.0 (every output becomes 32 instead of 16),
// N independent instructions (s_nop N-1, or N v_max3 on other registers), then the MFMA.  A lane whose result is 16 read the old
// operand.  1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N, bool kValu>
__global__ void __launch_bounds__(1024) probe(uint32_t *out, int iters)
{
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)1.0f;
    const uint32_t one2 = 0x3f803f80u, two2 = 0x40004000u;            // bf16 pairs (1, 1), (2, 2)
    uint32_t stale = 0, other = 0;
    float x = threadIdx.x, y = 1.0f, z = 2.0f;
    (void)x; (void)y; (void)z;
    for (int it = 0; it < iters; ++it) {
        float r0, r15;
        // A operand in v[48:51]: first all 1.0 (and settled), then v49..v51 rewritten with 2.0 right before the MFMA
        if (kValu)
            asm volatile("v_mov_b32 v48, %2\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %2\n\tv_mov_b32 v51, %2\n\ts_nop 7\n\ts_nop 7\n\t"
                         "v_mov_b32 v49, %3\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %3\n\t"
                         ".rept %5\n\tv_max3_f32 %6, %6, %7, %8\n\t.endr\n\t"
                         "v_mfma_f32_32x32x16_bf16 v[32:47], v[48:51], %4, 0\n\t"
                         "s_nop 15\n\ts_nop 15\n\t"
                         "v_mov_b32 %1, v47\n\tv_mov_b32 %0, v32\n\t"
                         : "=&v"(r0), "=&v"(r15) : "v"(one2), "v"(two2), "v"(b), "n"(N), "v"(x), "v"(y), "v"(z)
                         : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        else
            asm volatile("v_mov_b32 v48, %2\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %2\n\tv_mov_b32 v51, %2\n\ts_nop 7\n\ts_nop 7\n\t"
                         "v_mov_b32 v49, %3\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %3\n\t"
                         ".if %5 > 0\n\ts_nop %5 - 1\n\t.endif\n\t"
                         "v_mfma_f32_32x32x16_bf16 v[32:47], v[48:51], %4, 0\n\t"
                         "s_nop 15\n\ts_nop 15\n\t"
                         "v_mov_b32 %1, v47\n\tv_mov_b32 %0, v32\n\t"
                         : "=&v"(r0), "=&v"(r15) : "v"(one2), "v"(two2), "v"(b), "n"(N)
                         : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        // a = (1,1 | 2.. ) : k slots 0,1 hold 1.0, slots 2..7 hold 2.0 when the moves are seen: sum = 2 + 12 = 14 per lane half -> 28; all old: 16
        if (r0 == 16.0f || r15 == 16.0f) stale++;
        else if (r0 != 28.0f || r15 != 28.0f) other++;
    }
    if (stale) atomicAdd(&out[0], 1u);
    if (other) atomicAdd(&out[1], 1u);
    atomicAdd(&out[2], stale);
}

// Which output lanes are wrong when the B operand (ray columns: lane l holds column l % 32, K half l / 32) is rewritten by ONE v_mov
// directly in front of the MFMA?  A = all ones; B goes from all 1.0 to all 2.0 in register `which` (K slots 2 which, 2 which + 1): a
// correct lane reads 16 + 4 = 20... every output of column j is 16 + 2 * (new values of column j); stale lanes give 16 (or 18: one half).
template <int WHICH>
__global__ void __launch_bounds__(256) probe_b_lanes(unsigned long long *masks, int iters)
{
    bf16x8 a;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)1.0f;
    const uint32_t one2 = 0x3f803f80u, two2 = 0x40004000u;
    unsigned long long bad16 = 0, bad18 = 0;
    for (int it = 0; it < iters; ++it) {
        float r0;
        asm volatile("v_mov_b32 v48, %1\n\tv_mov_b32 v49, %1\n\tv_mov_b32 v50, %1\n\tv_mov_b32 v51, %1\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 v[48+%4], %2\n\t"
                     "v_mfma_f32_32x32x16_bf16 v[32:47], %3, v[48:51], 0\n\t"
                     "s_nop 15\n\ts_nop 15\n\t"
                     "v_mov_b32 %0, v32\n\t"
                     : "=&v"(r0) : "v"(one2), "v"(two2), "v"(a), "n"(WHICH)
                     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        // column j sums 16 K slots: 14 ones + the two rewritten slots of ONE K half (2.0 each when seen) = 14 + 4 = 18 when the lane that
        // holds that half was seen new... both halves of a column are different lanes: K slots of half 0 come from lane j, half 1 from lane j + 32
        if (r0 == 16.0f) bad16 |= 1ull << (threadIdx.x & 63);         // neither half's rewrite was seen
        else if (r0 == 18.0f) bad18 |= 1ull << (threadIdx.x & 63);    // one half's was
    }
    if (threadIdx.x < 64) { atomicOr(&masks[0], bad16); atomicOr(&masks[1], bad18); }
}
template <int WHICH> static void run_lanes(unsigned long long *d)
{
    (void)hipMemset(d, 0, 32);
    hipLaunchKernelGGL(probe_b_lanes<WHICH>, dim3(64), dim3(64), 0, 0, d, 2000);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("one v_mov of B register %d, then mfma (0 wait states): output lanes (= ray columns, both halves) with NEITHER K half updated %016llx, with ONE of two %016llx   (expected value 20)\n", WHICH, h[0], h[1]);
}

template <int N, bool kValu> static void run(uint32_t *d, int cus)
{
    for (int wps = 1; wps <= 4; wps *= 2) {
        (void)hipMemset(d, 0, 32);
        hipLaunchKernelGGL((probe<N, kValu>), dim3(cus), dim3(256 * wps), 0, 0, d, 20000);
        uint32_t h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("v_mov of SrcA, %d %s, mfma   waves/SIMD %d: lanes that saw the OLD operand %u (events %u), lanes with any other value %u\n",
               N, kValu ? "x v_max3" : "wait states (s_nop)", wps, h[0], h[2], h[1]);
    }
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    uint32_t *d; (void)hipMalloc(&d, 32);
    const int cus = prop.multiProcessorCount;
    run<0, false>(d, cus); run<1, false>(d, cus); run<2, false>(d, cus); run<3, false>(d, cus); run<4, false>(d, cus); run<6, false>(d, cus);
    run<1, true>(d, cus); run<2, true>(d, cus); run<3, true>(d, cus);
    run_lanes<0>((unsigned long long *)d); run_lanes<1>((unsigned long long *)d); run_lanes<2>((unsigned long long *)d); run_lanes<3>((unsigned long long *)d);
    return 0;
}
