// stale_line_probe.hip -- do plain stores of one kernel survive when several independent pipelines share the device?
//
// Round 2 saw, with three path-tracing pipelines rendering concurrently on one MI355X, 256 bytes of a ray queue holding
// the PREVIOUS frame's values some kernels after ray generation had rewritten them -- in memory, not in a cache.  The working hypothesis
// is about the eight per-XCD L2s: a line written on XCD A in one frame and on XCD B in the next reaching memory in the wrong order.
// This probe strips the scenario down to known values.  Per stream and iteration ("frame"), all enqueued without waiting:
//     gen     writes the whole queue a[0..n) with (index, frame) -- plain 16-byte stores, one element per lane, like generate_rays_kernel
//     filler  256 blocks x 512 threads holding 64 KB of LDS each (one block per CU, like the scan): reads the queue once (its lines
//             enter the L2 of whatever XCD the block runs on), then spins in LDS for ~20 us
//     shade   rewrites a prefix of the queue, a[0..n/2), with (index, frame | flag) -- the compacted rays of the next bounce
//     filler
//     verify  a[i] must be (i, frame | flag) below n/2 and (i, frame) above; a wrong element is recorded with what it holds
// Arms: 1 or 3 streams (separate queues: block -> XCD placement of a stream's kernels no longer repeats from frame to frame), plain or
// written-through stores, queue of 22,304 elements (a third of the 328 x 204 image of tools/diagnostics/flaky_tiled.py) or 691,200.
// Build: hipcc -O2 --offload-arch=gfx950 tools/stale_line_probe.hip -o stale_line_probe ; run: ./stale_line_probe [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Wrong { uint32_t stream, frame, index, found_index, found_frame, where; };

template <bool kThrough>
__device__ __forceinline__ void put(uint4 *p, uint4 v)
{
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v w = {v.x, v.y, v.z, v.w};
    if (kThrough) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
    else *p = v;
}

template <bool kThrough>
__global__ void __launch_bounds__(256) gen_kernel(uint4 *a, uint32_t n, uint32_t tag)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) put<kThrough>(a + i, make_uint4(i, tag, i ^ tag, ~i));
}

// one block per CU, as the scan: touches the queue through the ordinary cached path, then keeps the CU busy without memory traffic
__global__ void __launch_bounds__(512) filler_kernel(const uint4 *a, uint32_t n, uint32_t spins, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 512u + threadIdx.x; i < n; i += gridDim.x * 512u) acc += a[i].y;
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 0; s < spins; ++s) { acc = acc * 1664525u + lds[(threadIdx.x + s) & 511u]; lds[(threadIdx.x * 7u + s) & 511u] = acc; }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void __launch_bounds__(256) verify_kernel(const uint4 *a, uint32_t n, uint32_t tag, uint32_t half, uint32_t stream, uint32_t where, Wrong *log, uint32_t *n_wrong)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint4 v = a[i];
    const uint32_t want = i < half ? (tag | 0x80000000u) : tag;
    if (v.x != i || v.y != want || v.z != (i ^ want) || v.w != ~i) {
        const uint32_t at = atomicAdd(n_wrong, 1u);
        if (at < 256u) log[at] = Wrong{stream, tag, i, v.x, v.y, where};
    }
}

template <bool kThrough>
static int run_arm(int n_streams, uint32_t n, int frames, uint32_t spins, const char *name)
{
    std::vector<hipStream_t> st(n_streams);
    std::vector<uint4 *> q(n_streams);
    Wrong *log; uint32_t *n_wrong, *sink;
    CHECK(hipMalloc(&log, 256 * sizeof(Wrong))); CHECK(hipMalloc(&n_wrong, 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(n_wrong, 0, 4));
    for (int s = 0; s < n_streams; ++s) { CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CHECK(hipMalloc(&q[s], (size_t)n * 16)); CHECK(hipMemset(q[s], 0, (size_t)n * 16)); }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(filler_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipDeviceSynchronize());
    const uint32_t blocks = (n + 255u) / 256u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, st[0]));
    for (int f = 1; f <= frames; ++f)
        for (int s = 0; s < n_streams; ++s) {
            // (the prefix a later "bounce" rewrites differs from stream to stream and frame to frame, like the ray counts of a frame)
            const uint32_t half = (n / 2u + (uint32_t)(f * 37 + s * 101) % 997u) & ~15u;
            hipLaunchKernelGGL(gen_kernel<kThrough>, dim3(blocks), dim3(256), 0, st[s], q[s], n, (uint32_t)f);
            hipLaunchKernelGGL(verify_kernel, dim3(blocks), dim3(256), 0, st[s], q[s], n, (uint32_t)f, 0u, (uint32_t)s, 0u, log, n_wrong);
            hipLaunchKernelGGL(filler_kernel, dim3(256), dim3(512), 65536, st[s], q[s], n, spins, sink);
            hipLaunchKernelGGL(gen_kernel<kThrough>, dim3((half + 255u) / 256u), dim3(256), 0, st[s], q[s], half, (uint32_t)f | 0x80000000u);
            hipLaunchKernelGGL(filler_kernel, dim3(256), dim3(512), 65536, st[s], q[s], n, spins / 2u, sink);
            hipLaunchKernelGGL(verify_kernel, dim3(blocks), dim3(256), 0, st[s], q[s], n, (uint32_t)f, half, (uint32_t)s, 1u, log, n_wrong);
        }
    for (int s = 0; s < n_streams; ++s) CHECK(hipStreamSynchronize(st[s]));
    CHECK(hipEventRecord(e1, st[0])); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t h_wrong = 0; std::vector<Wrong> h(256);
    CHECK(hipMemcpy(&h_wrong, n_wrong, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h.data(), log, 256 * sizeof(Wrong), hipMemcpyDeviceToHost));
    std::printf("%-46s | %d stream(s) x %d frames of %u elements | %.0f ms | wrong elements: %u\n", name, n_streams, frames, n, ms, h_wrong);
    for (uint32_t k = 0; k < h_wrong && k < 24u; ++k)
        std::printf("    stream %u frame %u %s: a[%u] holds (index %u, frame %u%s)\n", h[k].stream, h[k].frame, h[k].where ? "at the end of the frame" : "behind gen",
                    h[k].index, h[k].found_index, h[k].found_frame & 0x7FFFFFFFu, (h[k].found_frame >> 31) ? " rewritten" : "");
    for (int s = 0; s < n_streams; ++s) { CHECK(hipStreamDestroy(st[s])); CHECK(hipFree(q[s])); }
    CHECK(hipFree(log)); CHECK(hipFree(n_wrong)); CHECK(hipFree(sink));
    return 0;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 3000;
    const uint32_t spins = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1500u;
    for (uint32_t n : {22304u, 691200u}) {
        const int fr = n > 100000u ? frames / 4 : frames;
        if (run_arm<false>(1, n, fr, spins, "plain stores, one pipeline")) return 1;
        if (run_arm<false>(3, n, fr, spins, "plain stores, three pipelines")) return 1;
        if (run_arm<false>(6, n, fr, spins, "plain stores, six pipelines")) return 1;
        if (run_arm<true>(3, n, fr, spins, "stores written through, three pipelines")) return 1;
    }
    return 0;
}
