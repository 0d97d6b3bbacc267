// vmcnt_order_probe.hip -- do vector-memory operations of one wave complete in issue order, as `s_waitcnt vmcnt(N)` with N > 0 assumes?
//
// LLVM's waitcnt insertion treats loads, stores and atomics without return as ONE in-order stream on targets without a separate store
// counter (the gfx9 family, gfx950 included): to use the result of the OLDER of two operations it waits for vmcnt(1).  If a younger
// store (or atomic) could be acknowledged before an older load has delivered its data, the wave would read the load's destination
// registers too early -- the stale-ray picture of DESIGN.md 5.2 (the old scan kernels have prefetch loads in flight while they issue
// candidate stores and atomicMin).  The probe: a far load (1 GiB pattern, ~2-4 us), then a younger operation on a line the wave
// touches all the time (fast), `s_waitcnt vmcnt(1)`, copy the load's destination at once, `s_waitcnt vmcnt(0)`, compare the copy.
// Younger operation: store / 64-bit atomic umin without return / near load (control).  Quiet, and with a second kernel keeping HBM busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline uint32_t pat(uint32_t i, uint32_t k) { return (i * 2654435761u) ^ (k * 0x9e3779b9u) ^ 0x5a5a5a5au; }

#define COPY_OUT "v_mov_b32 v12, v8\n\tv_mov_b32 v13, v9\n\tv_mov_b32 v14, v10\n\tv_mov_b32 v15, v11\n\ts_waitcnt vmcnt(0)"

template <int YOUNG>      // 0 store, 1 atomic umin (64 bit) without return, 2 near load
__global__ void __launch_bounds__(256) probe(const uint4 *__restrict__ pattern, uint32_t n_elems, uint32_t *near32, unsigned long long *near64, int iters, unsigned long long *bad)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t early = 0, other = 0;
    u32x4 prev = {0u, 0u, 0u, 0u};
    uint32_t *np = near32 + tid; unsigned long long *np64 = near64 + tid;
    for (int it = 0; it < iters; ++it) {
        const uint32_t base = (uint32_t)(((unsigned long long)(tid >> 6) * 2654435761ull + (unsigned long long)it * 40503ull * 64ull) % (n_elems / 64u)) * 64u;
        const uint32_t idx = base + (threadIdx.x & 63u);
        const uint4 *addr = pattern + idx;
        u32x4 d = prev, e;
        if (YOUNG == 0)
            asm volatile("global_load_dwordx4 v[8:11], %[addr], off\n\tglobal_store_dword %[np], %[v], off\n\ts_waitcnt vmcnt(1)\n\t" COPY_OUT
                         : "+{v[8:11]}"(d), "=&{v[12:15]}"(e) : [addr] "v"(addr), [np] "v"(np), [v] "v"((uint32_t)it) : "memory");
        else if (YOUNG == 1) {
            u32x2 key = {0xFFFFFFFFu - (uint32_t)it, 0x7FFFFFFFu};
            asm volatile("global_load_dwordx4 v[8:11], %[addr], off\n\tglobal_atomic_umin_x2 %[np], %[v], off\n\ts_waitcnt vmcnt(1)\n\t" COPY_OUT
                         : "+{v[8:11]}"(d), "=&{v[12:15]}"(e) : [addr] "v"(addr), [np] "v"(np64), [v] "v"(key) : "memory");
        } else {
            uint32_t sink;
            asm volatile("global_load_dwordx4 v[8:11], %[addr], off\n\tglobal_load_dword %[s], %[np], off\n\ts_waitcnt vmcnt(1)\n\t" COPY_OUT
                         : "+{v[8:11]}"(d), "=&{v[12:15]}"(e), [s] "=&v"(sink) : [addr] "v"(addr), [np] "v"(np) : "memory");
            other += sink & 0u;
        }
        const bool ok = e.x == pat(idx, 0) && e.y == pat(idx, 1) && e.z == pat(idx, 2) && e.w == pat(idx, 3);
        if (!ok) { if (e.x == prev.x && e.y == prev.y && e.z == prev.z && e.w == prev.w) early++; else other++; }
        prev = d;
    }
    const uint32_t q = (threadIdx.x & 63u) >> 4;
    if (early) atomicAdd(bad + q, (unsigned long long)early);
    if (other) atomicAdd(bad + 4 + q, (unsigned long long)other);
}

__global__ void __launch_bounds__(256) hogger(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n, int rounds)
{
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = src[(i * 7919u + (size_t)r * 104729u) % n]; v.x += (uint32_t)r; dst[i] = v; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int YOUNG> static int run(const char *name, const uint4 *d_pat, uint32_t n_elems, uint32_t *n32, unsigned long long *n64, unsigned long long *d_bad, int blocks, bool noise,
                                    hipStream_t sp, hipStream_t sh, const uint4 *hsrc, uint4 *hdst, size_t hn)
{
    CK(hipMemsetAsync(d_bad, 0, 128, sp)); CK(hipMemsetAsync(n64, 0xFF, (size_t)blocks * 256 * 8, sp));
    CK(hipStreamSynchronize(sp));
    if (noise) hipLaunchKernelGGL(hogger, dim3(2048), dim3(256), 0, sh, hsrc, hdst, hn, 6);
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sp));
    hipLaunchKernelGGL((probe<YOUNG>), dim3(blocks), dim3(256), 0, sp, d_pat, n_elems, n32, n64, iters, d_bad);
    CK(hipEventRecord(e1, sp));
    CK(hipStreamSynchronize(sp));
    const hipError_t still = noise ? hipStreamQuery(sh) : hipSuccess;
    CK(hipStreamSynchronize(sh));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16]; CK(hipMemcpy(h, d_bad, 128, hipMemcpyDeviceToHost));
    printf("far load, then %-28s | %s | pairs %.1e | load's registers read before its data arrived, by lane quarter: %llu %llu %llu %llu | other wrong value: %llu %llu %llu %llu | %.1f ms%s\n",
           name, noise ? "HBM busy" : "quiet   ", (double)blocks * 256.0 * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], ms, noise ? (still == hipErrorNotReady ? " (noise outlasted it)" : " (noise ended first)") : "");
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4;
    const uint32_t n_elems = 1u << 26;
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
    uint32_t *n32; unsigned long long *n64, *d_bad;
    CK(hipMalloc(&n32, (size_t)blocks * 256 * 4)); CK(hipMalloc(&n64, (size_t)blocks * 256 * 8)); CK(hipMalloc(&d_bad, 128));
    hipStream_t sp, sh; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    for (int noise = 0; noise < 2; ++noise) {
        if (run<2>("a near load (control)", d_pat, n_elems, n32, n64, d_bad, blocks, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<0>("a near store", d_pat, n_elems, n32, n64, d_bad, blocks, noise, sp, sh, hsrc, hdst, hn)) return 1;
        if (run<1>("a near atomic umin, no return", d_pat, n_elems, n32, n64, d_bad, blocks, noise, sp, sh, hsrc, hdst, hn)) return 1;
    }
    return 0;
}
