// mfma_load_return_probe.hip -- does a vector-memory load whose data comes back WHILE the wave issues the kernel-4 instruction stream
// lose part of its data?
//
// Where this comes from: several path-tracing pipelines rendering concurrently on one MI355X produced frames in which
// 16 consecutive rays of one scan launch had lost their mesh hit.  Narrowed down with builds of the round-1/2 scan (one wave per SIMD,
// rays of the next ray block prefetched by ordinary loads that are in flight across the hand-ordered MFMA stream, partly into AGPRs):
// the fault goes away (20/300 -> 0-1/300 wrong images) when those loads are issued and WAITED FOR before the stream starts -- with
// ordinary or with system-scope loads alike, so it is the overlap, not a cache.  16 lanes x 16 bytes is one return beat of a
// global_load_dwordx4.  Other kernels running at the same time only make the loads come back later, i.e. during the stream.
//
// The probe reproduces the overlap on known data: every iteration a wave issues a global_load_dwordx4 of a pattern element far
// from anything cached (latency ~2-4 us), immediately runs `stages` pipeline stages of the shipped stream (4 x (MFMA -> VGPR block, 8 VALU
// on the previous block)) WITHOUT waiting, then waits and checks all four dwords of every lane.  Arms: destination in VGPRs / in
// AGPRs; stream = full / VALU half only / none (s_sleep of the same length); a second kernel hammering HBM beside it or not.
// Wrong lanes are counted per quarter of the wave, and whether the wrong value is the one the register held before (a beat that
// never arrived) or something else.
// Result (profiles/r2_mfma_load_return_probe.txt): 0 wrong in 2.6e8 loads per arm -- the overlap alone does not lose data; what the old
// scan kernels did between the prefetch and its use is still open.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EXAMINE(P, M) \
    "v_min3_f32 v[" #M "+0], v[" #P "+0], v[" #P "+1], v[" #P "+2]\n\t" \
    "v_min3_f32 v[" #M "+1], v[" #P "+3], v[" #P "+4], v[" #P "+5]\n\t" \
    "v_min3_f32 v[" #M "+2], v[" #P "+6], v[" #P "+7], v[" #P "+8]\n\t" \
    "v_min3_f32 v[" #M "+3], v[" #P "+9], v[" #P "+10], v[" #P "+11]\n\t" \
    "v_min3_f32 v[" #M "+4], v[" #P "+12], v[" #P "+13], v[" #P "+14]\n\t" \
    "v_max3_f32 v[" #M "+5], v[" #M "+0], v[" #M "+1], v[" #M "+2]\n\t" \
    "v_max3_f32 v[" #M "+5], v[" #M "+5], v[" #M "+3], v[" #M "+4]\n\t"
#define CMP(M, K) "v_cmp_nle_f32_e64 s[" #K ":" #K "+1], v[" #M "+5], %[th]\n\t"
#define MFMA(N) "v_mfma_f32_32x32x16_bf16 v[" #N ":" #N "+15], %[a], %[b], 0\n\t"
#define STAGE_FULL(NB, PB) \
    MFMA(NB+0)  EXAMINE(PB+0, 96)  CMP(96, 20) \
    MFMA(NB+16) EXAMINE(PB+16, 104) CMP(104, 22) \
    MFMA(NB+32) EXAMINE(PB+32, 112) CMP(112, 24) \
    MFMA(NB+48) EXAMINE(PB+48, 120) CMP(120, 26)
#define STAGE_VALU(NB, PB) EXAMINE(PB+0, 96) CMP(96, 20) EXAMINE(PB+16, 104) CMP(104, 22) EXAMINE(PB+32, 112) CMP(112, 24) EXAMINE(PB+48, 120) CMP(120, 26)
// 8 stages per statement
#define EIGHT(S) S(192, 128) S(128, 192) S(192, 128) S(128, 192) S(192, 128) S(128, 192) S(192, 128) S(128, 192)

__host__ __device__ inline uint32_t pat(uint32_t i, uint32_t k) { return (i * 2654435761u) ^ (k * 0x9e3779b9u) ^ 0x5a5a5a5au; }

// STREAM: 0 full, 1 VALU half, 2 none.  kAgpr: the load's destination is an AGPR quadruple.
template <int STREAM, bool kAgpr>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
probe(const uint4 *__restrict__ pattern, uint32_t n_elems, int iters, int blocks8, unsigned long long *bad, float seed)
{
    extern __shared__ float hog[];
    if (seed == 777.0f) hog[threadIdx.x] = seed;
    u32x4 a, b;
    a.x = 0x3f803f80u + threadIdx.x; a.y = 0x3f003e80u; a.z = 0x40003f80u; a.w = 0x3f803f00u;
    b.x = 0x3f803f80u; b.y = 0x3e803f00u + threadIdx.x; b.z = 0x3f803f80u; b.w = 0x3f003f80u;
    float th = seed == 1.5f ? __builtin_inff() : seed;
    f32x16 X0, X1, X2, X3, Y0, Y1, Y2, Y3;
    for (int i = 0; i < 16; ++i) { X0[i] = X1[i] = X2[i] = X3[i] = seed + i; Y0[i] = Y1[i] = Y2[i] = Y3[i] = seed - i; }
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t wrong_stale = 0, wrong_other = 0;
    u32x4 prev = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it) {
        // a far-away element: lanes of a wave read 64 consecutive elements (1 KB), the wave jumps pseudo-randomly through a 1 GB buffer
        const uint32_t base = (uint32_t)(((unsigned long long)(tid >> 6) * 2654435761ull + (unsigned long long)it * 40503ull * 64ull) % (n_elems / 64u)) * 64u;
        const uint32_t idx = base + (threadIdx.x & 63u);
        const uint4 *addr = pattern + idx;
        u32x4 d;
#define CLOB "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc", "vcc", "memory", \
             "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
             "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define BLOCKS "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), "+{v[192:207]}"(Y0), "+{v[208:223]}"(Y1), "+{v[224:239]}"(Y2), "+{v[240:255]}"(Y3)
        if constexpr (kAgpr) {
            // preset the AGPRs with the previous value (what a lost beat would leave), load, stream, wait, read back
            asm volatile("v_accvgpr_write_b32 a0, %[p0]\n\tv_accvgpr_write_b32 a1, %[p1]\n\tv_accvgpr_write_b32 a2, %[p2]\n\tv_accvgpr_write_b32 a3, %[p3]\n\t"
                         "s_nop 4\n\tglobal_load_dwordx4 a[0:3], %[addr], off\n\t" : : [p0] "v"(prev.x), [p1] "v"(prev.y), [p2] "v"(prev.z), [p3] "v"(prev.w), [addr] "v"(addr) : "a0", "a1", "a2", "a3", "memory");
        } else {
            d = prev;
            asm volatile("global_load_dwordx4 %[d], %[addr], off" : [d] "+v"(d) : [addr] "v"(addr) : "memory");
        }
        for (int s8 = 0; s8 < blocks8; ++s8) {
            if (STREAM == 0) { if constexpr (kAgpr) asm volatile(EIGHT(STAGE_FULL) : BLOCKS : [a] "v"(a), [b] "v"(b), [th] "v"(th) : CLOB, "a0", "a1", "a2", "a3");
                               else asm volatile(EIGHT(STAGE_FULL) : BLOCKS, "+v"(d) : [a] "v"(a), [b] "v"(b), [th] "v"(th) : CLOB); }
            else if (STREAM == 1) { if constexpr (kAgpr) asm volatile(EIGHT(STAGE_VALU) : BLOCKS : [a] "v"(a), [b] "v"(b), [th] "v"(th) : CLOB, "a0", "a1", "a2", "a3");
                                    else asm volatile(EIGHT(STAGE_VALU) : BLOCKS, "+v"(d) : [a] "v"(a), [b] "v"(b), [th] "v"(th) : CLOB); }
            else { if constexpr (kAgpr) asm volatile("s_sleep 20" ::: "memory", "a0", "a1", "a2", "a3"); else asm volatile("s_sleep 20" : "+v"(d) :: "memory"); }
        }
        if constexpr (kAgpr) asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3"
                                          : "=v"(d.x), "=v"(d.y), "=v"(d.z), "=v"(d.w) :: "a0", "a1", "a2", "a3", "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(d) :: "memory");
        const bool ok = d.x == pat(idx, 0) && d.y == pat(idx, 1) && d.z == pat(idx, 2) && d.w == pat(idx, 3);
        if (!ok) { if (d.x == prev.x && d.y == prev.y && d.z == prev.z && d.w == prev.w) wrong_stale++; else wrong_other++; }
        prev = d;
    }
    const uint32_t q = (threadIdx.x & 63u) >> 4;
    if (wrong_stale) atomicAdd(bad + q, (unsigned long long)wrong_stale);
    if (wrong_other) atomicAdd(bad + 4 + q, (unsigned long long)wrong_other);
    if (seed == 12345.0f) bad[15] = (unsigned long long)(X0[0] + X1[1] + X2[2] + X3[3] + Y0[0] + Y1[1] + Y2[2] + Y3[3]) + (unsigned long long)hog[0];
}

// memory hog: keeps HBM and the fabric busy so that the probe's loads come back late
__global__ void __launch_bounds__(256) hogger(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n, int rounds)
{
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = src[(i * 7919u + (size_t)r * 104729u) % n]; v.x += (uint32_t)r; dst[i] = v; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int STREAM, bool kAgpr> static int run(const char *name, const uint4 *d_pat, uint32_t n_elems, unsigned long long *d_bad, int cus, bool noise, hipStream_t sp, hipStream_t sh, const uint4 *hsrc, uint4 *hdst, size_t hn)
{
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)&probe<STREAM, kAgpr>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(d_bad, 0, 128, sp));
    CK(hipStreamSynchronize(sp));
    if (noise) hipLaunchKernelGGL(hogger, dim3(2048), dim3(256), 0, sh, hsrc, hdst, hn, 6);
    const int iters = 4000, blocks8 = 2;            // 16 stages = 64 products (~3,000 cycles) behind every load
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sp));
    hipLaunchKernelGGL((probe<STREAM, kAgpr>), dim3(cus), dim3(256), lds, sp, d_pat, n_elems, iters, blocks8, d_bad, 1.5f);
    CK(hipEventRecord(e1, sp));
    CK(hipStreamSynchronize(sp));
    const hipError_t still = noise ? hipStreamQuery(sh) : hipSuccess;
    CK(hipStreamSynchronize(sh));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16]; CK(hipMemcpy(h, d_bad, 128, hipMemcpyDeviceToHost));
    printf("%-58s | %s | loads checked %.1e | register kept its old value, by lane quarter: %llu %llu %llu %llu | other wrong value: %llu %llu %llu %llu | %.1f ms%s\n",
           name, noise ? "HBM busy " : "quiet    ", (double)cus * 256.0 * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], ms, noise ? (still == hipErrorNotReady ? " (noise outlasted it)" : " (noise ended first)") : "");
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t n_elems = 1u << 26;                       // 1 GiB of pattern: nothing stays cached
    uint4 *d_pat; CK(hipMalloc(&d_pat, (size_t)n_elems * 16));
    {
        std::vector<uint4> h(1u << 20);
        for (uint32_t c = 0; c < n_elems; c += (1u << 20)) {
            for (uint32_t i = 0; i < (1u << 20); ++i) h[i] = make_uint4(pat(c + i, 0), pat(c + i, 1), pat(c + i, 2), pat(c + i, 3));
            CK(hipMemcpy(d_pat + c, h.data(), (size_t)(1u << 20) * 16, hipMemcpyHostToDevice));
        }
    }
    const size_t hn = (size_t)1 << 26;
    uint4 *hsrc, *hdst; CK(hipMalloc(&hsrc, hn * 16)); CK(hipMalloc(&hdst, hn * 16)); CK(hipMemset(hsrc, 1, hn * 16));
    unsigned long long *d_bad; CK(hipMalloc(&d_bad, 128));
    hipStream_t sp, sh; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    for (int noise = 0; noise < 2; ++noise) {
        if (run<2, false>("no stream (s_sleep), load into VGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<1, false>("VALU half of the stream, load into VGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<0, false>("full stream (MFMA + VALU), load into VGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<2, true>("no stream (s_sleep), load into AGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<1, true>("VALU half of the stream, load into AGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<0, true>("full stream (MFMA + VALU), load into AGPRs", d_pat, n_elems, d_bad, cus, noise, sp, sh, hsrc, hdst, hn)) return 1;
    }
    return 0;
}
