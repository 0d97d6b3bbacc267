// coresident_probe.hip -- what happens to a wave that shares a SIMD with the kernel-4 instruction stream?
//
// Hypothesis that prompted it (round 2, tools/diagnostics/flaky_multi.py): three contexts rendering on ONE device produced, every few
// dozen runs, an image with 16 wrong pixels -- always queue slots 48..63 of a 64-slot block -- while the scan's own counters stayed
// exact, only with the one-wave-per-SIMD scan (which left room on its SIMDs for waves of other kernels), and not with a build that
// claimed all 512 registers.  Is a wave that shares a SIMD with the matrix-instruction stream disturbed by it?
// Result (profiles/r2_coresident_probe.txt): NO -- 0 wrong results in 1e9 checks per arm.  (The fault was later narrowed down to the
// scan's own asynchronous ray prefetch: DESIGN.md 5.2, tools/mfma_load_return_probe.hip.)
//
// This probe isolates it on known operands: an aggressor kernel (one wave per SIMD, ~340 registers, 100 KB of LDS so that one block
// owns a CU) runs a stream of its choice for ~100 ms; beside it, on a second stream, small victim kernels (24-64 registers) run
// loops whose results are known exactly, and count wrong results per lane quarter.
//   aggressors: full (4 x (MFMA -> VGPR block, 8 VALU examining the previous block): the shipped stage), mfma (MFMAs only), valu (the
//   VALU half only), the shipped trip with its two LDS reads, idle (no aggressor)
//   victims: int (v_mul_lo_u32 chain, quarter rate), fma (v_fma_f32 chain, full rate), load (global_load_dwordx4 of a known pattern),
//   lds (ds_write / ds_read of a known pattern), trans (v_sqrt / v_rcp chain against a quiet run's values)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
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
#define STAGE_MFMA(NB, PB) MFMA(NB+0) MFMA(NB+16) MFMA(NB+32) MFMA(NB+48)

// A = which stream; kFull: the wave claims all 512 registers of its SIMD (nothing can run beside it: the control for the control)
template <int A, bool kFull>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) aggressor(unsigned long long *out, int iters, float seed)
{
    extern __shared__ float hog[];
    if (seed == 777.0f) hog[threadIdx.x] = seed;
    if (kFull) asm volatile("" ::: "a255"); else asm volatile("" ::: "a80");
    u32x4 a, b;
    a.x = 0x3f803f80u + threadIdx.x; a.y = 0x3f003e80u; a.z = 0x40003f80u; a.w = 0x3f803f00u;
    b.x = 0x3f803f80u; b.y = 0x3e803f00u + threadIdx.x; b.z = 0x3f803f80u; b.w = 0x3f003f80u;
    float th = seed == 1.5f ? __builtin_inff() : seed;
    u32x4 ny, nx;
    const uint32_t lds_addr = (threadIdx.x & 63u) * 16u + (threadIdx.x >> 6) * 4096u;
    f32x16 X0, X1, X2, X3, Y0, Y1, Y2, Y3;
    for (int i = 0; i < 16; ++i) { X0[i] = X1[i] = X2[i] = X3[i] = seed + i; Y0[i] = Y1[i] = Y2[i] = Y3[i] = seed - i; }
    for (int it = 0; it < iters; ++it) {
#define RUN(STAGE) \
        asm volatile(STAGE(192, 128) STAGE(128, 192) \
                     : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), \
                       "+{v[192:207]}"(Y0), "+{v[208:223]}"(Y1), "+{v[224:239]}"(Y2), "+{v[240:255]}"(Y3) \
                     : [a] "v"(a), [b] "v"(b), [th] "v"(th) \
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc", "vcc", \
                       "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
                       "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127")
        if (A == 0) RUN(STAGE_FULL);
        else if (A == 1) RUN(STAGE_MFMA);
        else if (A == 2) RUN(STAGE_VALU);
        else {
            // the shipped trip: two LDS reads of 16 bytes per lane behind the first matrix instruction, waited for at the end
            asm volatile(MFMA(192) "ds_read_b128 %[ny], %[addr]\n\tds_read_b128 %[nx], %[addr] offset:1024\n\t" EXAMINE(128, 96) CMP(96, 20)
                         MFMA(208) EXAMINE(144, 104) CMP(104, 22) MFMA(224) EXAMINE(160, 112) CMP(112, 24) MFMA(240) EXAMINE(176, 120) CMP(120, 26)
                         STAGE_FULL(128, 192) "s_waitcnt lgkmcnt(0)\n\ts_or_b64 s[20:21], s[20:21], s[22:23]\n\ts_or_b64 s[20:21], s[20:21], s[24:25]\n\t"
                         : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3),
                           "+{v[192:207]}"(Y0), "+{v[208:223]}"(Y1), "+{v[224:239]}"(Y2), "+{v[240:255]}"(Y3), [ny] "=&v"(ny), [nx] "=&v"(nx)
                         : [a] "v"(a), [b] "v"(b), [th] "v"(th), [addr] "v"(lds_addr)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc", "vcc", "memory",
                           "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111",
                           "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127");
            a.x ^= ny.x & 1u; b.y ^= nx.y & 1u;
        }
    }
    if (seed == 12345.0f) out[0] = (unsigned long long)(X0[0] + X1[1] + X2[2] + X3[3] + Y0[0] + Y1[1] + Y2[2] + Y3[3]) + (unsigned long long)hog[0];
}

// ---- victims: wrong results per lane quarter -> bad[quarter]; bad[4] += launches' blocks (to see that they ran)
struct Lcg { static __host__ __device__ uint32_t step(uint32_t x) { return x * 1664525u + 1013904223u; } };

__global__ void __launch_bounds__(256) victim_int(unsigned long long *bad, const uint32_t *expect, int n)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < n; ++i) x = Lcg::step(x);
    uint32_t y = threadIdx.x * 2654435761u;                    // expectation of block 0's seeds comes from the host; others: the chain is run twice
    uint32_t x2 = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < n; ++i) { x2 = Lcg::step(x2); asm volatile("" : "+v"(x2)); }
    (void)y;
    const bool wrong = (x != x2) || (blockIdx.x == 0 && x != expect[threadIdx.x]);
    if (wrong) atomicAdd(bad + ((threadIdx.x & 63) >> 4), 1ull);
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

__global__ void __launch_bounds__(256) victim_fma(unsigned long long *bad, const float *expect, int n)
{
    float x = 1.0f + (float)threadIdx.x * 0.001f, x2 = x;
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 0.99999f, 0.25f);
    for (int i = 0; i < n; ++i) { x2 = __builtin_fmaf(x2, 0.99999f, 0.25f); asm volatile("" : "+v"(x2)); }
    const bool wrong = (__float_as_uint(x) != __float_as_uint(x2)) || (__float_as_uint(x) != __float_as_uint(expect[threadIdx.x]));
    if (wrong) atomicAdd(bad + ((threadIdx.x & 63) >> 4), 1ull);
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

// pattern buffer: element i = (4i, 4i+1, 4i+2, 4i+3) ^ 0x5a5a5a5a
__global__ void __launch_bounds__(256) victim_load(unsigned long long *bad, const uint4 *pattern, uint32_t n_elems, int n)
{
    uint32_t wrong = 0;
    uint32_t i = (blockIdx.x * 256u + threadIdx.x) % n_elems;
    for (int k = 0; k < n; ++k) {
        const uint4 v = pattern[i];
        const uint32_t e = (4u * i) ^ 0x5a5a5a5au;
        wrong += (v.x != e) || (v.y != ((4u * i + 1u) ^ 0x5a5a5a5au)) || (v.z != ((4u * i + 2u) ^ 0x5a5a5a5au)) || (v.w != ((4u * i + 3u) ^ 0x5a5a5a5au));
        i = (i + 256u * 61u) % n_elems;
    }
    if (wrong) atomicAdd(bad + ((threadIdx.x & 63) >> 4), (unsigned long long)wrong);
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

__global__ void __launch_bounds__(256) victim_lds(unsigned long long *bad, int n)
{
    __shared__ uint4 s[256];
    uint32_t wrong = 0;
    for (int k = 0; k < n; ++k) {
        const uint32_t e = threadIdx.x * 977u + (uint32_t)k * 131u + blockIdx.x;
        s[threadIdx.x] = make_uint4(e, e + 1u, e + 2u, e + 3u);
        __syncthreads();
        const uint4 v = s[threadIdx.x ^ 1u];
        const uint32_t f = (threadIdx.x ^ 1u) * 977u + (uint32_t)k * 131u + blockIdx.x;
        wrong += (v.x != f) || (v.y != f + 1u) || (v.z != f + 2u) || (v.w != f + 3u);
        __syncthreads();
    }
    if (wrong) atomicAdd(bad + ((threadIdx.x & 63) >> 4), (unsigned long long)wrong);
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

// mode 0: write the chain's value (quiet run); mode 1: compare with it
__global__ void __launch_bounds__(256) victim_trans(unsigned long long *bad, float *quiet, int n, int mode)
{
    float x = 2.0f + (float)threadIdx.x * 0.01f;
    for (int i = 0; i < n; ++i) x = __builtin_amdgcn_sqrtf(x) + __builtin_amdgcn_rcpf(x + 1.0f) + 1.5f;
    if (mode == 0) { if (blockIdx.x == 0) quiet[threadIdx.x] = x; }
    else if (__float_as_uint(x) != __float_as_uint(quiet[threadIdx.x])) atomicAdd(bad + ((threadIdx.x & 63) >> 4), 1ull);
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

// the shape of the product's own victims: a float4 record per lane loaded, a few dependent VALU operations, stored; verified by a second pass
__global__ void __launch_bounds__(256) victim_stream(unsigned long long *bad, const float4 *in, float4 *outp, uint32_t n_elems, int check)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_elems) return;
    const float4 v = in[i];
    const float4 r = make_float4(v.x * 1.5f + v.y, v.y * 0.5f - v.z, v.z + v.w * 2.0f, v.w - v.x);
    if (!check) outp[i] = r;
    else {
        const float4 o = outp[i];
        if (__float_as_uint(o.x) != __float_as_uint(r.x) || __float_as_uint(o.y) != __float_as_uint(r.y) || __float_as_uint(o.z) != __float_as_uint(r.z) || __float_as_uint(o.w) != __float_as_uint(r.w))
            atomicAdd(bad + ((threadIdx.x & 63) >> 4), 1ull);
    }
    if (threadIdx.x == 0) atomicAdd(bad + 4, 1ull);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_bad, *d_sink; CK(hipMalloc(&d_bad, 64)); CK(hipMalloc(&d_sink, 64));
    hipStream_t sa, sv; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const int chain = 4000;
    std::vector<uint32_t> e_int(256); std::vector<float> e_fma(256);
    for (int t = 0; t < 256; ++t) {
        uint32_t x = (uint32_t)t * 2654435761u; for (int i = 0; i < chain; ++i) x = Lcg::step(x); e_int[t] = x;
        float f = 1.0f + (float)t * 0.001f; for (int i = 0; i < chain; ++i) f = fmaf(f, 0.99999f, 0.25f); e_fma[t] = f;
    }
    uint32_t *d_eint; float *d_efma, *d_quiet; CK(hipMalloc(&d_eint, 1024)); CK(hipMalloc(&d_efma, 1024)); CK(hipMalloc(&d_quiet, 1024));
    CK(hipMemcpy(d_eint, e_int.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d_efma, e_fma.data(), 1024, hipMemcpyHostToDevice));
    const uint32_t n_elems = 1u << 20;
    std::vector<uint4> pat(n_elems);
    for (uint32_t i = 0; i < n_elems; ++i) pat[i] = make_uint4((4u * i) ^ 0x5a5a5a5au, (4u * i + 1u) ^ 0x5a5a5a5au, (4u * i + 2u) ^ 0x5a5a5a5au, (4u * i + 3u) ^ 0x5a5a5a5au);
    uint4 *d_pat; CK(hipMalloc(&d_pat, (size_t)n_elems * 16)); CK(hipMemcpy(d_pat, pat.data(), (size_t)n_elems * 16, hipMemcpyHostToDevice));
    float4 *d_sin, *d_sout; CK(hipMalloc(&d_sin, (size_t)n_elems * 16)); CK(hipMalloc(&d_sout, (size_t)n_elems * 16));
    { std::vector<float4> h(n_elems); for (uint32_t i = 0; i < n_elems; ++i) h[i] = make_float4((float)(i % 977) * 0.25f, (float)(i % 131), 1.0f / (float)(1 + i % 17), (float)(i % 7) - 3.0f); CK(hipMemcpy(d_sin, h.data(), (size_t)n_elems * 16, hipMemcpyHostToDevice)); }
    const size_t hog_bytes = 100 * 1024;
    for (const void *fn : {(const void *)&aggressor<0, false>, (const void *)&aggressor<1, false>, (const void *)&aggressor<2, false>, (const void *)&aggressor<0, true>, (const void *)&aggressor<3, false>})
        CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hog_bytes));
    hipLaunchKernelGGL(victim_trans, dim3(1), dim3(256), 0, sv, d_bad, d_quiet, chain, 0); CK(hipStreamSynchronize(sv));

    const char *agg_names[] = {"idle", "full stream (MFMA + VALU), 337 registers", "MFMAs only, 337 registers", "VALU half only, 337 registers", "full stream, all 512 registers", "full stream + 2 LDS reads per trip, 337 registers"};
    const char *vic_names[] = {"int chain", "fma chain", "load x4", "lds", "sqrt/rcp chain", "load-compute-store"};
    for (int agg = 0; agg < 6; ++agg) {
        for (int vic = 0; vic < 6; ++vic) {
            CK(hipMemset(d_bad, 0, 64));
            const int agg_iters = 400000;                                                // ~100 ms of stream
            if (agg == 1) hipLaunchKernelGGL((aggressor<0, false>), dim3(cus), dim3(256), hog_bytes, sa, d_sink, agg_iters, 1.5f);
            if (agg == 2) hipLaunchKernelGGL((aggressor<1, false>), dim3(cus), dim3(256), hog_bytes, sa, d_sink, agg_iters, 1.5f);
            if (agg == 3) hipLaunchKernelGGL((aggressor<2, false>), dim3(cus), dim3(256), hog_bytes, sa, d_sink, agg_iters, 1.5f);
            if (agg == 5) hipLaunchKernelGGL((aggressor<3, false>), dim3(cus), dim3(256), hog_bytes, sa, d_sink, agg_iters, 1.5f);
            if (agg == 4) hipLaunchKernelGGL((aggressor<0, true>), dim3(cus), dim3(256), hog_bytes, sa, d_sink, agg_iters, 1.5f);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, sv));
            for (int rep = 0; rep < 40; ++rep) {
                const dim3 g(4096);
                if (vic == 0) hipLaunchKernelGGL(victim_int, g, dim3(256), 0, sv, d_bad, d_eint, chain);
                if (vic == 1) hipLaunchKernelGGL(victim_fma, g, dim3(256), 0, sv, d_bad, d_efma, chain);
                if (vic == 2) hipLaunchKernelGGL(victim_load, g, dim3(256), 0, sv, d_bad, d_pat, n_elems, 64);
                if (vic == 3) hipLaunchKernelGGL(victim_lds, g, dim3(256), 0, sv, d_bad, 64);
                if (vic == 4) hipLaunchKernelGGL(victim_trans, g, dim3(256), 0, sv, d_bad, d_quiet, chain, 1);
                if (vic == 5) {
                    hipLaunchKernelGGL(victim_stream, g, dim3(256), 0, sv, d_bad, d_sin, d_sout, n_elems, 0);
                    hipLaunchKernelGGL(victim_stream, g, dim3(256), 0, sv, d_bad, d_sin, d_sout, n_elems, 1);
                }
            }
            CK(hipEventRecord(e1, sv));
            CK(hipStreamSynchronize(sv));
            float vms = 0; CK(hipEventElapsedTime(&vms, e0, e1));
            const hipError_t still = hipStreamQuery(sa);                                 // was the aggressor still running when the victims ended?
            CK(hipStreamSynchronize(sa));
            unsigned long long h[8]; CK(hipMemcpy(h, d_bad, 64, hipMemcpyDeviceToHost));
            printf("%-44s | %-18s | wrong results by lane quarter: %llu %llu %llu %llu | victim blocks %llu in %.1f ms%s\n", agg_names[agg], vic_names[vic], h[0], h[1], h[2], h[3], h[4], vms,
                   agg == 0 ? "" : (still == hipErrorNotReady ? " (aggressor outlasted them)" : " (aggressor ended first)"));
            fflush(stdout);
        }
    }
    return 0;
}
