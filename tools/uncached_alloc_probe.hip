// uncached_alloc_probe.hip -- does hipExtMallocWithFlags(hipDeviceMallocUncached) change anything on this stack?  (round 2
// tried it for the ray queues against the stale-line fault, saw "same failure rate, same speed" and could not tell whether the flag had
// taken effect.)  Reports the pointer attributes of a plain and a flagged allocation and the time of (1) a streaming read-modify-write
// kernel and (2) a kernel that re-reads ONE 64 KB window 256 times (cache-resident for cached memory, fabric-bound for uncached).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void stream_rmw(float4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.0f; p[i] = v; } }
__global__ void reread(const float4 *p, float *out, int reps)
{
    float acc = 0.0f;
    for (int r = 0; r < reps; ++r) { const float4 v = p[(threadIdx.x + 256u * ((blockIdx.x + r) & 15u))]; acc += v.x + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
static float time_ms(void (*launch)(void *), void *arg)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(arg); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 10; ++i) launch(arg); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
static size_t g_n = (size_t)64 << 20;
static float *g_out;
static void l_stream(void *p) { hipLaunchKernelGGL(stream_rmw, dim3(2048), dim3(256), 0, 0, (float4 *)p, g_n / 16); }
static void l_reread(void *p) { hipLaunchKernelGGL(reread, dim3(4096), dim3(256), 0, 0, (const float4 *)p, g_out, 256); }

int main()
{
    void *plain = nullptr, *unc = nullptr, *fine = nullptr;
    (void)hipMalloc(&g_out, 64);
    hipError_t e0 = hipMalloc(&plain, g_n);
    hipError_t e1 = hipExtMallocWithFlags(&unc, g_n, hipDeviceMallocUncached);
    hipError_t e2 = hipExtMallocWithFlags(&fine, g_n, hipDeviceMallocFinegrained);
    printf("hipMalloc: %s   hipExtMallocWithFlags(Uncached): %s   (Finegrained): %s\n", hipGetErrorString(e0), hipGetErrorString(e1), hipGetErrorString(e2));
    const char *names[3] = {"plain", "uncached", "finegrained"}; void *ptrs[3] = {plain, unc, fine};
    for (int k = 0; k < 3; ++k) {
        if (!ptrs[k]) continue;
        hipPointerAttribute_t at; hipError_t e = hipPointerGetAttributes(&at, ptrs[k]);
        unsigned flags = 0; hipError_t ef = hipPointerGetAttribute(&flags, HIP_POINTER_ATTRIBUTE_MEMORY_TYPE, (hipDeviceptr_t)ptrs[k]);
        printf("%-12s attributes: %s type %d device %d isManaged %d allocationFlags 0x%x | memory-type attribute: %s %u\n", names[k], hipGetErrorString(e), (int)at.type, at.device, at.isManaged, at.allocationFlags, hipGetErrorString(ef), flags);
        (void)hipMemset(ptrs[k], 0, g_n);
        const float s = time_ms(l_stream, ptrs[k]), r = time_ms(l_reread, ptrs[k]);
        printf("%-12s streaming read-modify-write of 64 MiB: %.3f ms = %.0f GB/s     256 re-reads of a 64 KB window by 4096 blocks: %.3f ms\n", names[k], s, 2.0 * g_n / s / 1e6, r);
    }
    return 0;
}
