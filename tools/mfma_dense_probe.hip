// mfma_dense_probe.hip -- pipe-dense probe for the kernel-3 fault: does a VALU read of a
// v_mfma_f32_32x32x16_bf16 result ever see something else than the result when SEVERAL waves keep the matrix pipe of a SIMD
// busy and vector-memory / LDS returns land in the register file at the same time -- and if so, WHAT does it see?
//
// Operands are known: A has a one in K slot 0 of every row, B carries a per-iteration, per-column small integer c in K slot 0,
// so every one of the 16 result registers of a lane must read float(c(it, lane % 32)).  c changes every iteration, so a stale
// register (the previous result of the same block) is distinguishable from a foreign or garbage value.
//
//   RAW<G>   mfma X; s_nop G; read X15, X0           the interleaved order of the faulty builds (G + 1 wait states)
//   PIPE     mfma X; mfma X'; s_nop 7; read Y, Y'    the shipped order of kernel 3: results read one stage later
//   WAR<G>   read X15 (old); s_nop G; mfma X         VALU read directly in front of the overwrite
// each with 1..4 waves per SIMD and with/without a global_load + ds_write + ds_read per iteration in every wave.
// Output: per arm the number of wrong reads, how many of them were stale / zero / other, and the lane quarters hit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Log { uint32_t wrong, stale, zero, other, quarter[4], first[8][6]; };

__device__ __forceinline__ float expect(int it, int col) { return (float)(((it * 3 + col) & 127) + 1); }

__device__ __forceinline__ void record(Log *log, int it, int lane, int which, float got, float want, float prev)
{
    const uint32_t n = atomicAdd(&log->wrong, 1u);
    if (got == prev) atomicAdd(&log->stale, 1u); else if (got == 0.0f) atomicAdd(&log->zero, 1u); else atomicAdd(&log->other, 1u);
    atomicAdd(&log->quarter[lane >> 4], 1u);
    if (n < 8u) { uint32_t *f = log->first[n]; f[0] = (uint32_t)it; f[1] = (uint32_t)lane; f[2] = (uint32_t)which; f[3] = __float_as_uint(got); f[4] = __float_as_uint(want); f[5] = __float_as_uint(prev); }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: RAW<G>, 1: PIPE, 2: WAR<G>.  The accumulator blocks are C++ values bound to physical registers by constraint, so the
// compiler keeps them intact between the asm statements; inside a statement the single registers are named directly.
template <int MODE, int G, bool TRAFFIC>
__global__ void __launch_bounds__(1024) dense(Log *log, const uint4 *__restrict__ buf, int iters)
{
    __shared__ uint32_t lds[1024 * 2];
    const int lane = threadIdx.x & 63, col = lane & 31;
    u32x4 a; a.x = lane < 32 ? 0x00003f80u : 0u; a.y = a.z = a.w = 0u;   // bf16(1.0) in K slot 0 of every row, zeros elsewhere
    uint32_t sink = 0;
    uint4 ld = make_uint4(0u, 0u, 0u, 0u);
    float prev = 0.0f, prev2 = 0.0f;
    f32x16 X0, X1, Y0, Y1;
    for (int i = 0; i < 16; ++i) X0[i] = X1[i] = Y0[i] = Y1[i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const float want = expect(it, col);
        u32x4 b; b.x = __float_as_uint(want) >> 16; b.y = b.z = b.w = 0u;
        if (TRAFFIC) {
            sink += ld.x;                                                                // last iteration's load is consumed here
            ld = buf[(size_t)((it * 977 + (int)blockIdx.x * 131) & 4095) * 16 + (threadIdx.x & 15)];
            lds[threadIdx.x] = sink; sink += lds[(threadIdx.x + 64) & 1023];
        }
        float r0, r15;
        if (MODE == 0) {
            asm volatile("v_mfma_f32_32x32x16_bf16 v[64:79], %3, %4, 0\n\ts_nop %5\n\tv_mov_b32 %1, v79\n\tv_mov_b32 %0, v64"
                         : "=&v"(r0), "=&v"(r15), "={v[64:79]}"(X0) : "v"(a), "v"(b), "n"(G));
            if (r15 != want) record(log, it, lane, 15, r15, want, prev);
            if (r0 != want) record(log, it, lane, 0, r0, want, prev);
        } else if (MODE == 1) {
            // two products into X (even iterations) or Y (odd), then the blocks written one iteration ago are read
            float q0, q15;
            if ((it & 1) == 0)
                asm volatile("v_mfma_f32_32x32x16_bf16 v[64:79], %8, %9, 0\n\tv_mfma_f32_32x32x16_bf16 v[80:95], %8, %9, 0\n\ts_nop 7\n\t"
                             "v_mov_b32 %1, v111\n\tv_mov_b32 %0, v96\n\tv_mov_b32 %3, v127\n\tv_mov_b32 %2, v112"
                             : "=&v"(r0), "=&v"(r15), "=&v"(q0), "=&v"(q15), "={v[64:79]}"(X0), "={v[80:95]}"(X1)
                             : "{v[96:111]}"(Y0), "{v[112:127]}"(Y1), "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 v[96:111], %8, %9, 0\n\tv_mfma_f32_32x32x16_bf16 v[112:127], %8, %9, 0\n\ts_nop 7\n\t"
                             "v_mov_b32 %1, v79\n\tv_mov_b32 %0, v64\n\tv_mov_b32 %3, v95\n\tv_mov_b32 %2, v80"
                             : "=&v"(r0), "=&v"(r15), "=&v"(q0), "=&v"(q15), "={v[96:111]}"(Y0), "={v[112:127]}"(Y1)
                             : "{v[64:79]}"(X0), "{v[80:95]}"(X1), "v"(a), "v"(b));
            if (it > 0) {                          // what is read is the result of iteration it - 1; stale = the one of it - 3
                if (r15 != prev) record(log, it, lane, 15, r15, prev, expect(it - 3, col));
                if (r0 != prev) record(log, it, lane, 0, r0, prev, expect(it - 3, col));
                if (q15 != prev) record(log, it, lane, 31, q15, prev, expect(it - 3, col));
                if (q0 != prev) record(log, it, lane, 16, q0, prev, expect(it - 3, col));
            }
        } else {
            asm volatile("v_mov_b32 %1, v79\n\tv_mov_b32 %0, v64\n\ts_nop %5\n\tv_mfma_f32_32x32x16_bf16 v[64:79], %3, %4, 0\n\ts_nop 15\n\ts_nop 15"
                         : "=&v"(r0), "=&v"(r15), "+{v[64:79]}"(X0) : "v"(a), "v"(b), "n"(G));
            if (r15 != prev) record(log, it, lane, 15, r15, prev, want);                 // "stale" here = already the NEW value
            if (r0 != prev) record(log, it, lane, 0, r0, prev, want);
        }
        prev2 = prev; prev = want;
    }
    if (sink == 0x12345678u && prev2 == -1.0f) log->other += ld.y + (uint32_t)(X0[3] + X1[3] + Y0[3] + Y1[3]);   // keeps everything alive
}

template <int MODE, int G, bool TRAFFIC> static void arm(const char *name, Log *d_log, const uint4 *buf, int cus, int iters)
{
    for (int wps = 1; wps <= 4; ++wps) {
        (void)hipMemset(d_log, 0, sizeof(Log));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((dense<MODE, G, TRAFFIC>), dim3(cus), dim3(256 * wps), 0, 0, d_log, buf, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        Log h; (void)hipMemcpy(&h, d_log, sizeof h, hipMemcpyDeviceToHost);
        const double products = (double)iters * (MODE == 1 ? 2 : 1);
        printf("%-8s G %2d traffic %d waves/SIMD %d: %7.2f ms, %6.1f ns per product per SIMD (pipe busy %3.0f%%); wrong %u (stale %u zero %u other %u) quarters %u %u %u %u\n",
               name, G, (int)TRAFFIC, wps, ms, ms * 1e6 / (products * wps), 100.0 * 32.0 / 2.4 / (ms * 1e6 / (products * wps)),
               h.wrong, h.stale, h.zero, h.other, h.quarter[0], h.quarter[1], h.quarter[2], h.quarter[3]);
        for (uint32_t i = 0; i < (h.wrong < 4u ? h.wrong : 4u); ++i) {
            float got, want, prev; memcpy(&got, &h.first[i][3], 4); memcpy(&want, &h.first[i][4], 4); memcpy(&prev, &h.first[i][5], 4);
            printf("      it %u lane %u reg %u: got %g (0x%08x) want %g stale-value %g\n", h.first[i][0], h.first[i][1], h.first[i][2], got, h.first[i][3], want, prev);
        }
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = argc > 1 ? atoi(argv[1]) : 400000;
    Log *d_log; (void)hipMalloc(&d_log, sizeof(Log));
    uint4 *buf; (void)hipMalloc(&buf, 4096 * 16 * sizeof(uint4)); (void)hipMemset(buf, 1, 4096 * 16 * sizeof(uint4));
    printf("%d CUs, %d iterations per wave\n", cus, iters);
    arm<0, 10, false>("RAW", d_log, buf, cus, 64);            // 11 wait states: must fail (sanity of the probe; every read is logged, so keep it short)
    arm<0, 11, false>("RAW", d_log, buf, cus, iters);
    arm<0, 11, true>("RAW", d_log, buf, cus, iters);
    arm<0, 12, true>("RAW", d_log, buf, cus, iters);
    arm<0, 15, true>("RAW", d_log, buf, cus, iters);
    arm<1, 7, false>("PIPE", d_log, buf, cus, iters);
    arm<1, 7, true>("PIPE", d_log, buf, cus, iters);
    arm<2, 0, false>("WAR", d_log, buf, cus, iters);
    arm<2, 0, true>("WAR", d_log, buf, cus, iters);
    arm<2, 1, true>("WAR", d_log, buf, cus, iters);
    return 0;
}
