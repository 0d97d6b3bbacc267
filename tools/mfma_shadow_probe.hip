// mfma_shadow_probe.hip -- how many INDEPENDENT v_max3_f32 of the SAME wave issue for free behind a v_mfma_f32_32x32x16_bf16?
// One asm block per loop iteration: 4 x (MFMA into its own accumulator, K independent v_max3 on other registers).
// Reports ns and cycles (at the measured clock of an MFMA-only loop = 32 cycles per MFMA) per MFMA for K = 0..10, 1..3 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K>
__global__ void __launch_bounds__(1024) shadow(float *out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x); b[i] = (__bf16)1.0f; }
    float x0 = threadIdx.x, x1 = 1.0f, x2 = 2.0f, x3 = 3.0f, y = 0.5f, z = 0.25f;
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 v[32:47], %4, %5, 0\n\t"
            ".rept %8\n\tv_max3_f32 %0, %0, %6, %7\n\t.endr\n\t"
            "v_mfma_f32_32x32x16_bf16 v[48:63], %4, %5, 0\n\t"
            ".rept %8\n\tv_max3_f32 %1, %1, %6, %7\n\t.endr\n\t"
            "v_mfma_f32_32x32x16_bf16 v[64:79], %4, %5, 0\n\t"
            ".rept %8\n\tv_max3_f32 %2, %2, %6, %7\n\t.endr\n\t"
            "v_mfma_f32_32x32x16_bf16 v[80:95], %4, %5, 0\n\t"
            ".rept %8\n\tv_max3_f32 %3, %3, %6, %7\n\t.endr\n\t"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(y), "v"(z), "n"(K)
            : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47",
              "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63",
              "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",
              "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int K> static double run(float *out, int cus, int wps, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(shadow<K>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e6 / ((double)iters * 4 * wps);      // ns per MFMA per SIMD
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    float *out; (void)hipMalloc(&out, 1024 * 4096 * sizeof(float));
    const int cus = prop.multiProcessorCount, iters = 20000;
    for (int wps = 1; wps <= 3; ++wps) {
        const double base = run<0>(out, cus, wps, iters);
        printf("waves/SIMD %d: MFMA only %.2f ns (= 32 cycles => %.2f GHz)\n", wps, base, 32.0 / base);
        const double t[] = {run<1>(out, cus, wps, iters), run<2>(out, cus, wps, iters), run<3>(out, cus, wps, iters), run<4>(out, cus, wps, iters),
                            run<5>(out, cus, wps, iters), run<6>(out, cus, wps, iters), run<8>(out, cus, wps, iters), run<10>(out, cus, wps, iters)};
        const int ks[] = {1, 2, 3, 4, 5, 6, 8, 10};
        for (int i = 0; i < 8; ++i) printf("   + %2d independent v_max3 per MFMA: %.2f ns = %.1f cycles per MFMA (+%.1f cycles, %.2f per v_max3)\n",
                                           ks[i], t[i], t[i] / base * 32.0, (t[i] / base - 1.0) * 32.0, (t[i] / base - 1.0) * 32.0 / ks[i]);
    }
    return 0;
}
