// mfma_layout_check.hip -- verifies the operand / result lane maps rt_mfma.hpp assumes for
// v_mfma_f32_32x32x16_bf16 with exact small-integer data (asymmetric A and B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D)     // A[32][16], B[16][32], D[32][32]
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}
int main()
{
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) A[i * 16 + kk] = (float)((i * 3 + kk * 5) % 7 - 3);
    for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) B[kk * 32 + j] = (float)((kk * 2 + j * 7) % 5 - 2);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int kk = 0; kk < 16; ++kk) R[i * 32 + j] += A[i * 16 + kk] * B[kk * 32 + j];
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("mfma_f32_32x32x16_bf16 layout check: %d of 1024 elements differ\n", bad);
    return bad != 0;
}
