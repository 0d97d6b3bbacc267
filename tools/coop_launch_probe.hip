// coop_launch_probe.hip -- what does replacing two dependent small launches by ONE launch with a grid-wide barrier save on MI355X?
//
// A frame of the path tracer is 26 small-to-medium kernels, each waiting for the one before (no gaps between them, but every
// kernel pays its own start and drain: `narrow_phase` 6-13 us and `shade` 7-17 us per launch for a rank of eight, whatever they process).
// Fusing two of them needs a barrier across the grid between the phases.  Three ways to run "phase 1: x[i] = f(i); phase 2: y[i] = g(x[j(i)])"
// (phase 2 reads what any block of phase 1 wrote), n elements, 256 blocks of 256 threads, grid-stride:
//     two    two ordinary launches on one stream
//     coop   one hipLaunchCooperativeKernel, cooperative_groups grid sync between the phases
//     spin   one ordinary launch, hand-made barrier (one arrive counter in global memory, s_sleep polling).  Only as many blocks as there
//            are CUs, so that all are resident on an idle device; the poll gives up after a bounded number of rounds (and the run is
//            flagged), so every wave reaches its end whatever happens
// Build: hipcc -O2 --offload-arch=gfx950 tools/coop_launch_probe.hip -o tools/coop_launch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void phase1(uint32_t *x, uint32_t n, uint32_t it)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = i * 2654435761u + it;
}
__device__ __forceinline__ void phase2(const uint32_t *x, uint32_t *y, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = x[(uint32_t)(((unsigned long long)i * 7919ull) % n)] ^ i;
}
__global__ void __launch_bounds__(256) k_phase1(uint32_t *x, uint32_t n, uint32_t it) { phase1(x, n, it); }
__global__ void __launch_bounds__(256) k_phase2(const uint32_t *x, uint32_t *y, uint32_t n) { phase2(x, y, n); }
__global__ void __launch_bounds__(256) k_coop(uint32_t *x, uint32_t *y, uint32_t n, uint32_t it)
{
    phase1(x, n, it);
    cg::this_grid().sync();
    phase2(x, y, n);
}
__global__ void __launch_bounds__(256) k_spin(uint32_t *x, uint32_t *y, uint32_t n, uint32_t it, uint32_t *arrive, uint32_t *gave_up)
{
    phase1(x, n, it);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = gridDim.x * (it + 1u);                     // the counter is never reset: launch `it` waits for (it + 1) * blocks arrivals
        uint32_t rounds = 0;
        while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++rounds > 200000u) { atomicAdd(gave_up, 1u); break; }  // (tens of ms: never on an idle device; the exit every wave reaches)
        }
    }
    __syncthreads();
    __threadfence();
    phase2(x, y, n);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    int dev = 0, coop = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    std::printf("device %d: %d CUs, cooperative launch %s\n", dev, cus, coop ? "supported" : "NOT supported");
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    uint32_t *arrive, *gave_up; CHECK(hipMalloc(&arrive, 4)); CHECK(hipMalloc(&gave_up, 4));
    const uint32_t blocks = (uint32_t)std::min(cus, 256);
    for (uint32_t n : {1024u, 16384u, 262144u, 2097152u}) {
        uint32_t *x, *y; CHECK(hipMalloc(&x, (size_t)n * 4)); CHECK(hipMalloc(&y, (size_t)n * 4));
        std::vector<uint32_t> want(n), got(n);
        float ms_two = 0, ms_coop = -1, ms_spin = 0;
        // two launches
        for (int w = 0; w < 2; ++w) {
            CHECK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) {
                hipLaunchKernelGGL(k_phase1, dim3(blocks), dim3(256), 0, st, x, n, (uint32_t)it);
                hipLaunchKernelGGL(k_phase2, dim3(blocks), dim3(256), 0, st, x, y, n);
            }
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_two, e0, e1));
        }
        CHECK(hipMemcpy(want.data(), y, (size_t)n * 4, hipMemcpyDeviceToHost));
        // cooperative
        bool ok_coop = true;
        if (coop) {
            for (int w = 0; w < 2; ++w) {
                CHECK(hipEventRecord(e0, st));
                for (int it = 0; it < iters; ++it) {
                    uint32_t itv = (uint32_t)it, nn = n;
                    void *args[] = {&x, &y, &nn, &itv};
                    CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_coop), dim3(blocks), dim3(256), args, 0, st));
                }
                CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_coop, e0, e1));
            }
            CHECK(hipMemcpy(got.data(), y, (size_t)n * 4, hipMemcpyDeviceToHost));
            ok_coop = got == want;
        }
        // hand-made barrier
        uint32_t h_gave = 0;
        for (int w = 0; w < 2; ++w) {
            CHECK(hipMemsetAsync(arrive, 0, 4, st)); CHECK(hipMemsetAsync(gave_up, 0, 4, st));
            CHECK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, x, y, n, (uint32_t)it, arrive, gave_up);
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_spin, e0, e1));
            CHECK(hipMemcpy(&h_gave, gave_up, 4, hipMemcpyDeviceToHost));
            if (h_gave) break;
        }
        CHECK(hipMemcpy(got.data(), y, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::printf("n = %8u | two launches %6.2f us per pair | cooperative + grid sync %6.2f us%s | one launch + hand-made barrier %6.2f us%s%s\n", n, ms_two * 1e3 / iters,
                    ms_coop * 1e3 / iters, ok_coop ? "" : " (WRONG RESULT)", ms_spin * 1e3 / iters, got == want ? "" : " (WRONG RESULT)", h_gave ? " (barrier gave up)" : "");
        CHECK(hipFree(x)); CHECK(hipFree(y));
    }
    return 0;
}
