// scan_stage_rate.hip -- cycles per pipeline stage of the kernel-4 scan loop (rt_scan.hpp), one wave per SIMD, and what each
// ingredient of the stage costs.  A stage = 4 x (v_mfma_f32_32x32x16_bf16 into block N_s, 8 VALU examining block P_s).
// Variants knock out one ingredient at a time; cycles come from s_memtime around the loop (clock independent).
//   0 full      the stage as shipped: MFMA -> VGPR block, 5 v_min3 + 2 v_max3 + v_cmp on the MFMA-written block of the last stage
//   1 other     the same VALU instructions read registers no MFMA ever writes
//   2 nocmp     v_cmp (SGPR destination) replaced by a third v_max3
//   3 valu      no MFMAs at all
//   4 mfma      no VALU at all
//   (5 agpr     MFMA destinations in AGPRs: measured 184, the same as 0; removed so that the kernel fits 256 registers for the x2 arms)
//   6 split     MFMA, 4 VALU of set s, 4 VALU of set s-1 ... (distance between dependent VALU instructions >= 4)
//   7 copy      per block: 8 v_mov of the MFMA-written block into scratch registers, nothing else (pure read cost)
//   8..10       how the v_cmp results are consumed by scalar code (see the STAGE_SOR* macros)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EXAMINE(P, M, TH) \
    "v_min3_f32 v[" #M "+0], v[" #P "+0], v[" #P "+1], v[" #P "+2]\n\t" \
    "v_min3_f32 v[" #M "+1], v[" #P "+3], v[" #P "+4], v[" #P "+5]\n\t" \
    "v_min3_f32 v[" #M "+2], v[" #P "+6], v[" #P "+7], v[" #P "+8]\n\t" \
    "v_min3_f32 v[" #M "+3], v[" #P "+9], v[" #P "+10], v[" #P "+11]\n\t" \
    "v_min3_f32 v[" #M "+4], v[" #P "+12], v[" #P "+13], v[" #P "+14]\n\t" \
    "v_max3_f32 v[" #M "+5], v[" #M "+0], v[" #M "+1], v[" #M "+2]\n\t" \
    "v_max3_f32 v[" #M "+5], v[" #M "+5], v[" #M "+3], v[" #M "+4]\n\t"
#define CMP(M, K) "v_cmp_nle_f32_e64 s[" #K ":" #K "+1], v[" #M "+5], %[th]\n\t"
#define MAX3(M, K) "v_max3_f32 v[" #M "+6], v[" #M "+5], v[" #M "+3], %[th]\n\t"
#define MFMA(N) "v_mfma_f32_32x32x16_bf16 v[" #N ":" #N "+15], %[a], %[b], 0\n\t"
#define MFMA_A(N) "v_mfma_f32_32x32x16_bf16 a[" #N ":" #N "+15], %[a], %[b], 0\n\t"
#define COPY15(P, M) \
    "v_mov_b32 v[" #M "+0], v[" #P "+0]\n\tv_mov_b32 v[" #M "+1], v[" #P "+1]\n\tv_mov_b32 v[" #M "+2], v[" #P "+2]\n\tv_mov_b32 v[" #M "+3], v[" #P "+3]\n\t" \
    "v_mov_b32 v[" #M "+4], v[" #P "+4]\n\tv_mov_b32 v[" #M "+5], v[" #P "+5]\n\tv_mov_b32 v[" #M "+6], v[" #P "+6]\n\tv_mov_b32 v[" #M "+7], v[" #P "+7]\n\t"

// blocks: X = v[128..191], Y = v[192..255]; M = v[96..127]; "other" registers v[32..95]
#define STAGE_FULL(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, 20) \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, 24) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) CMP(120, 26)
#define STAGE_OTHER(NB, PB) \
    MFMA(NB+0)  EXAMINE(32, 96, th)  CMP(96, 20) \
    MFMA(NB+16) EXAMINE(48, 104, th) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(64, 112, th) CMP(112, 24) \
    MFMA(NB+48) EXAMINE(80, 120, th) CMP(120, 26)
#define STAGE_NOCMP(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  MAX3(96, 20) \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) MAX3(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) MAX3(112, 24) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) MAX3(120, 26)
#define SOR(K) "s_or_b64 s[28:29], s[28:29], s[" #K ":" #K "+1]\n\t"
#define SBR "s_cmp_eq_u64 s[28:29], 0\n\ts_cbranch_scc0 9f\n\t"
// 8: every v_cmp result consumed by an s_or right behind its block (what the compiler generated in round 2's first build)
#define STAGE_SOR(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, 20) "s_mov_b64 s[28:29], s[20:21]\n\t" \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, 22) SOR(22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, 24) SOR(24) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) CMP(120, 26) SOR(26) SBR
// 9: the three s_or, the compare and the (never taken) branch at the end of the stage
#define STAGE_SOREND(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, 20) \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, 24) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) CMP(120, 26) "s_or_b64 s[28:29], s[20:21], s[22:23]\n\t" SOR(24) SOR(26) SBR
// 10: the scalar work on the masks of the PREVIOUS stage, behind the first block of this one (masks alternate between two SGPR sets)
#define STAGE_SORDEFER(NB, PB, KA, KB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, KA) "s_or_b64 s[28:29], s[" #KB ":" #KB "+1], s[" #KB "+2:" #KB "+3]\n\t" SOR(KB+4) SOR(KB+6) SBR \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, KA+2) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, KA+4) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) CMP(120, KA+6)
// 11: five s_nop 0 at the end of the stage (issue cost of five scalar instructions, no dependency)
#define STAGE_NOP5(NB, PB) STAGE_FULL(NB, PB) "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
// 12: the three s_or of the fresh masks, no compare, no branch
#define STAGE_OR3(NB, PB) STAGE_FULL(NB, PB) "s_or_b64 s[28:29], s[20:21], s[22:23]\n\t" SOR(24) SOR(26)
// 13: compare + never-taken branch on a scalar no vector instruction wrote
#define STAGE_BR(NB, PB) STAGE_FULL(NB, PB) "s_cmp_eq_u64 s[36:37], 0\n\ts_cbranch_scc0 9f\n\t"
// 14: one s_or of the LAST block's fresh mask only
#define STAGE_OR1(NB, PB) STAGE_FULL(NB, PB) "s_or_b64 s[28:29], s[36:37], s[26:27]\n\t"
// 15: one s_or of the FIRST block's mask (written three blocks earlier)
#define STAGE_OR1OLD(NB, PB) STAGE_FULL(NB, PB) "s_or_b64 s[28:29], s[36:37], s[20:21]\n\t"
// 16: s_cmp early (behind block 0), the never-taken s_cbranch_scc0 at the end of the stage
#define STAGE_BRSCC(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, 20) "s_cmp_eq_u64 s[36:37], 0\n\t" \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, 24) \
    MFMA(NB+48) EXAMINE(PB+48, 120, th) CMP(120, 26) "s_cbranch_scc0 9f\n\t"
// 17: never-taken s_cbranch_vccnz at the end of the stage, vcc written before the loop
#define STAGE_BRVCC(NB, PB) STAGE_FULL(NB, PB) "s_cbranch_vccnz 9f\n\t"
// 18: never-taken s_cbranch_vccnz right behind the LAST MFMA of the stage (inside its shadow)
#define STAGE_BRMID(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96, th)  CMP(96, 20) \
    MFMA(NB+16) EXAMINE(PB+16, 104, th) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112, th) CMP(112, 24) \
    MFMA(NB+48) "s_cbranch_vccnz 9f\n\t" EXAMINE(PB+48, 120, th) CMP(120, 26)
// 19: margin folded into the product: 20 v_min3, 10 v_max3, ONE v_cmp (vcc) per stage, consumed by the branch of the NEXT stage
#define MIN5(P, M) \
    "v_min3_f32 v[" #M "+0], v[" #P "+0], v[" #P "+1], v[" #P "+2]\n\t" \
    "v_min3_f32 v[" #M "+1], v[" #P "+3], v[" #P "+4], v[" #P "+5]\n\t" \
    "v_min3_f32 v[" #M "+2], v[" #P "+6], v[" #P "+7], v[" #P "+8]\n\t" \
    "v_min3_f32 v[" #M "+3], v[" #P "+9], v[" #P "+10], v[" #P "+11]\n\t" \
    "v_min3_f32 v[" #M "+4], v[" #P "+12], v[" #P "+13], v[" #P "+14]\n\t"
#define MX(D, A, B, C) "v_max3_f32 v[" #D "], v[" #A "], v[" #B "], v[" #C "]\n\t"
#define STAGE_FOLD(NB, PB) \
    MFMA(NB+0)  "s_cbranch_vccnz 9f\n\t" MIN5(PB+0, 96) \
    MFMA(NB+16) MIN5(PB+16, 104) MX(101, 96, 97, 98) MX(102, 99, 100, 104) \
    MFMA(NB+32) MIN5(PB+32, 112) MX(103, 105, 106, 107) MX(109, 108, 112, 113) MX(110, 101, 102, 103) \
    MFMA(NB+48) MIN5(PB+48, 120) MX(111, 114, 115, 116) MX(117, 120, 121, 122) MX(118, 123, 124, 109) MX(119, 110, 111, 117) MX(119, 119, 118, 118) \
    "v_cmp_gt_f32_e32 vcc, 0, v119\n\t"       /* (inverted for the probe: its operands are positive, the branch must not be taken) */
// 20: shared-edge tiles: 12 triangles per tile as 6 pairs of 5 rows per lane half (the shared edge's row serves both triangles, negated
// for the second): 6 v_min3 (three with a negated operand), 3 v_max, 1 v_cmp per block
#define EXAM_PAIRS(P, M) \
    "v_min3_f32 v[" #M "+0], v[" #P "+0], v[" #P "+1], v[" #P "+2]\n\t" \
    "v_min3_f32 v[" #M "+1], v[" #P "+3], v[" #P "+4], -v[" #P "+2]\n\t" \
    "v_min3_f32 v[" #M "+2], v[" #P "+5], v[" #P "+6], v[" #P "+7]\n\t" \
    "v_min3_f32 v[" #M "+3], v[" #P "+8], v[" #P "+9], -v[" #P "+7]\n\t" \
    "v_min3_f32 v[" #M "+4], v[" #P "+10], v[" #P "+11], v[" #P "+12]\n\t" \
    "v_min3_f32 v[" #M "+5], v[" #P "+13], v[" #P "+14], -v[" #P "+12]\n\t" \
    "v_max3_f32 v[" #M "+6], v[" #M "+0], v[" #M "+1], v[" #M "+2]\n\t" \
    "v_max3_f32 v[" #M "+7], v[" #M "+3], v[" #M "+4], v[" #M "+5]\n\t" \
    "v_max_f32 v[" #M "+6], v[" #M "+6], v[" #M "+7]\n\t"
#define CMP6(M, K) "v_cmp_nle_f32_e64 s[" #K ":" #K "+1], v[" #M "+6], %[th]\n\t"
#define STAGE_PAIRS(NB, PB) \
    MFMA(NB+0)  EXAM_PAIRS(PB+0, 96)  CMP6(96, 20) \
    MFMA(NB+16) EXAM_PAIRS(PB+16, 104) CMP6(104, 22) \
    MFMA(NB+32) EXAM_PAIRS(PB+32, 112) CMP6(112, 24) \
    MFMA(NB+48) EXAM_PAIRS(PB+48, 120) CMP6(120, 26)
// 21: the same with the shipped scalar spot (three s_or + compare + never-taken branch at the end of the stage)
#define STAGE_PAIRS_SOR(NB, PB) STAGE_PAIRS(NB, PB) "s_or_b64 s[28:29], s[20:21], s[22:23]\n\t" SOR(24) SOR(26) SBR
#define STAGE_VALU(NB, PB) \
    EXAMINE(PB+0, 96, th)  CMP(96, 20) EXAMINE(PB+16, 104, th) CMP(104, 22) EXAMINE(PB+32, 112, th) CMP(112, 24) EXAMINE(PB+48, 120, th) CMP(120, 26)
#define STAGE_MFMA(NB, PB) MFMA(NB+0) MFMA(NB+16) MFMA(NB+32) MFMA(NB+48)
#define STAGE_AGPR(NB, PB) \
    MFMA_A(NB-128+0)  EXAMINE(32, 96, th)  CMP(96, 20) \
    MFMA_A(NB-128+16) EXAMINE(48, 104, th) CMP(104, 22) \
    MFMA_A(NB-128+32) EXAMINE(64, 112, th) CMP(112, 24) \
    MFMA_A(NB-128+48) EXAMINE(80, 120, th) CMP(120, 26)
#define STAGE_COPY(NB, PB) \
    MFMA(NB+0)  COPY15(PB+0, 96) \
    MFMA(NB+16) COPY15(PB+16, 104) \
    MFMA(NB+32) COPY15(PB+32, 112) \
    MFMA(NB+48) COPY15(PB+48, 120)

#define CLOB_V(a) "v" #a
#define C8(a) CLOB_V(a##0), CLOB_V(a##1), CLOB_V(a##2), CLOB_V(a##3), CLOB_V(a##4), CLOB_V(a##5), CLOB_V(a##6), CLOB_V(a##7), CLOB_V(a##8), CLOB_V(a##9)

template <int V, int WPS = 1>
__global__ void __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) stage_rate(unsigned long long *out, int iters, float seed)
{
    u32x4 a, b;
    a.x = 0x3f803f80u + threadIdx.x; a.y = 0x3f003e80u; a.z = 0x40003f80u; a.w = 0x3f803f00u;
    b.x = 0x3f803f80u; b.y = 0x3e803f00u + threadIdx.x; b.z = 0x3f803f80u; b.w = 0x3f003f80u;
    float th = seed == 1.5f ? __builtin_inff() : seed;       // +inf: no compare ever passes, the branches are never taken
    f32x16 X0, X1, X2, X3, Y0, Y1, Y2, Y3;
    for (int i = 0; i < 16; ++i) { X0[i] = X1[i] = X2[i] = X3[i] = seed + i; Y0[i] = Y1[i] = Y2[i] = Y3[i] = seed - i; }
    asm volatile("s_mov_b64 s[20:21], 0\n\ts_mov_b64 s[22:23], 0\n\ts_mov_b64 s[24:25], 0\n\ts_mov_b64 s[26:27], 0\n\ts_mov_b64 s[28:29], 0\n\t"
                 "s_mov_b64 s[30:31], 0\n\ts_mov_b64 s[32:33], 0\n\ts_mov_b64 s[34:35], 0\n\ts_mov_b64 s[36:37], 0"
                 ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37");
    asm volatile("s_mov_b64 vcc, 0" ::: "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define RUN(STAGE) RUN2(STAGE(192, 128) STAGE(128, 192) "9:\n\t")
#define RUN2(TEXT) \
        asm volatile(TEXT \
                     : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), \
                       "+{v[192:207]}"(Y0), "+{v[208:223]}"(Y1), "+{v[224:239]}"(Y2), "+{v[240:255]}"(Y3) \
                     : [a] "v"(a), [b] "v"(b), [th] "v"(th) \
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "scc", "vcc", \
                       "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
                       "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
                       "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
                       "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
                       "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
                       "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127")
        if (V == 0) RUN(STAGE_FULL);
        else if (V == 1) RUN(STAGE_OTHER);
        else if (V == 2) RUN(STAGE_NOCMP);
        else if (V == 3) RUN(STAGE_VALU);
        else if (V == 4) RUN(STAGE_MFMA);
        else if (V == 7) RUN(STAGE_COPY);
        else if (V == 8) RUN(STAGE_SOR);
        else if (V == 9) RUN(STAGE_SOREND);
        else if (V == 11) RUN(STAGE_NOP5);
        else if (V == 12) RUN(STAGE_OR3);
        else if (V == 13) RUN(STAGE_BR);
        else if (V == 14) RUN(STAGE_OR1);
        else if (V == 15) RUN(STAGE_OR1OLD);
        else if (V == 16) RUN(STAGE_BRSCC);
        else if (V == 17) RUN(STAGE_BRVCC);
        else if (V == 18) RUN(STAGE_BRMID);
        else if (V == 19) RUN(STAGE_FOLD);
        else if (V == 20) RUN(STAGE_PAIRS);
        else if (V == 21) RUN(STAGE_PAIRS_SOR);
        else if (V == 10) RUN2(STAGE_SORDEFER(192, 128, 20, 30) STAGE_SORDEFER(128, 192, 30, 20) "9:\n\t");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
    if (seed == 12345.0f) out[1] = (unsigned long long)(X0[0] + X1[1] + X2[2] + X3[3] + Y0[0] + Y1[1] + Y2[2] + Y3[3]);
}

template <int V, int WPS = 1> static void run(const char *name, unsigned long long *d, int cus)
{
    const int iters = 20000;
    (void)hipMemset(d, 0, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stage_rate<V, WPS>), dim3(cus), dim3(256 * WPS), 0, 0, d, 1000, 1.5f);       // warm-up
    (void)hipMemset(d, 0, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((stage_rate<V, WPS>), dim3(cus), dim3(256 * WPS), 0, 0, d, iters, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / (cus * 4.0 * WPS) / (iters * 2.0) / WPS;      // cycles of a SIMD per stage: two waves share it
    printf("%-11s %7.1f cycles per stage (4 products), %6.1f per product; %.2f ms => %.2f GHz\n", name, cyc, cyc / 4.0, ms, (double)h[0] / (cus * 4.0) / (ms * 1e6));
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    unsigned long long *d; (void)hipMalloc(&d, 16);
    const int cus = prop.multiProcessorCount;
    run<0>("full", d, cus); run<1>("other", d, cus); run<2>("nocmp", d, cus); run<3>("valu", d, cus);
    run<4>("mfma", d, cus); run<7>("copy", d, cus);
    run<8>("sor", d, cus); run<9>("sorend", d, cus); run<10>("sordefer", d, cus);
    run<11>("nop5", d, cus); run<12>("or3", d, cus); run<13>("br", d, cus); run<14>("or1", d, cus); run<15>("or1old", d, cus);
    run<16>("brscc", d, cus); run<17>("brvcc", d, cus); run<18>("brmid", d, cus); run<19>("fold", d, cus);
    run<0>("full", d, cus); run<13>("br", d, cus);
    // two waves per SIMD (256 registers each): does a second wave fill the ~40-cycle holes scalar instructions tear into the stream?
    run<20>("pairs", d, cus); run<21>("pairs+sor", d, cus); run<9>("sorend", d, cus);
    run<20, 2>("pairs x2", d, cus); run<21, 2>("pairs+sor x2", d, cus); run<19, 2>("fold x2", d, cus);
    run<0, 2>("full x2", d, cus); run<13, 2>("br x2", d, cus); run<8, 2>("sor x2", d, cus); run<9, 2>("sorend x2", d, cus); run<10, 2>("sordefer x2", d, cus);
    return 0;
}
