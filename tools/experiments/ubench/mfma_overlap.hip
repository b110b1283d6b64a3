// GPU box, standalone (no torch): (1) checks the operand / result layout of v_mfma_f32_32x32x2_f32 and the half exchange
// of v_permlane32_swap that the MFMA blend relies on; (2) measures whether f32 MFMAs run beside VALU work of the same
// wave / of other waves of the SIMD (time of the mixed loop vs the two pure loops).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void k_layout(const float* A /*32x2*/, const float* B /*2x32*/, float* D /*32x32*/, unsigned* sw) {
    const unsigned l = threadIdx.x;
    f32x16 c = {0};
    const float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l >> 5) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[i];
    u32x2 r = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    sw[2 * l] = r[0]; sw[2 * l + 1] = r[1];
}

template <int MODE, int NV>
__global__ __launch_bounds__(64) void k_mix(float* out, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    f32x16 acc0 = {0}, acc1 = {0};
    float a = seed * 0.001f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < NV / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-3f);
        }
        if (MODE & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < NV / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-3f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// same question for the bf16 MFMA (a real matrix-pipe instruction): MODE bit 0 = MFMA, bit 1 = NV fma per MFMA
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k_mix_bf16(float* out, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * 0.001f * (i + 1)); b[i] = (__bf16)(0.5f + i); }
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < NV / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-3f);
        }
        if (MODE & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < NV / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-3f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE, int NV> static float run16(int blocks, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix_bf16<MODE, NV>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mix_bf16<MODE, NV>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE, int NV> static float run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *A, *B, *D; unsigned* sw;
    hipMallocManaged(&A, 64 * 4); hipMallocManaged(&B, 64 * 4); hipMallocManaged(&D, 1024 * 4); hipMallocManaged(&sw, 128 * 4);
    for (int i = 0; i < 64; ++i) { A[i] = (float)(rand() % 1000) * 1e-3f + 0.1234567f; B[i] = (float)(rand() % 1000) * 1e-3f - 0.3456789f; }
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, A, B, D, sw);
    hipDeviceSynchronize();
    int bad = 0, bad_alt = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        const float e = fmaf(A[m * 2 + 1], B[32 + n], fmaf(A[m * 2], B[n], 0.0f));      // k = 0 first, then k = 1
        const float e2 = fmaf(A[m * 2], B[n], fmaf(A[m * 2 + 1], B[32 + n], 0.0f));
        if (D[m * 32 + n] != e) ++bad;
        if (D[m * 32 + n] != e2) ++bad_alt;
    }
    printf("layout: %d of 1024 differ from the fmaf chain k=0,1 (%d from k=1,0)\n", bad, bad_alt);
    int swbad = 0;
    for (unsigned l = 0; l < 64; ++l) {
        const unsigned e0 = l < 32 ? 1000u + l : 2000u + (l - 32), e1 = l < 32 ? 1000u + (l + 32) : 2000u + l;
        if (sw[2 * l] != e0 || sw[2 * l + 1] != e1) ++swbad;
    }
    printf("permlane32_swap: %d of 64 lanes differ from {vdst.hi <-> src.lo}; lane0 = (%u, %u), lane32 = (%u, %u)\n", swbad, sw[0], sw[1], sw[64], sw[65]);
    float* out; hipMalloc(&out, 16);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * 4 * wps;
        const float m = run<1, 24>(blocks, iters, out), v24 = run<2, 24>(blocks, iters, out), b24 = run<3, 24>(blocks, iters, out);
        const float v48 = run<2, 48>(blocks, iters, out), b48 = run<3, 48>(blocks, iters, out);
        // per SIMD: wps waves x iters x 2 MFMA (x NV fma each)
        printf("waves/SIMD %d: mfma-only %.3f ms (%.1f cyc/mfma/SIMD) | 24 fma per mfma: valu %.3f, both %.3f | 48 fma per mfma: valu %.3f, both %.3f ms\n",
               wps, m, m * 1e-3 * 2.4e9 / (wps * iters * 2.0), v24, b24, v48, b48);
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * 4 * wps;
        const float m = run16<1, 48>(blocks, iters, out), v48 = run16<2, 48>(blocks, iters, out), b48 = run16<3, 48>(blocks, iters, out);
        const float v96 = run16<2, 96>(blocks, iters, out), b96 = run16<3, 96>(blocks, iters, out);
        printf("bf16 32x32x16, waves/SIMD %d: mfma-only %.3f ms (%.1f cyc/mfma/SIMD) | 48 fma per mfma: valu %.3f, both %.3f | 96 fma per mfma: valu %.3f, both %.3f ms\n",
               wps, m, m * 1e-3 * 2.4e9 / (wps * iters * 2.0), v48, b48, v96, b96);
    }
    return 0;
}
