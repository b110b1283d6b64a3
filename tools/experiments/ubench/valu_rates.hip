// Micro-benchmark (GPU box): issue rate of the VALU instruction kinds the blend kernel is made of, per SIMD.
// Every wave runs ITER trips of 8 independent chains of one instruction kind; waves_per_simd is set by the grid.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float pk2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(float* out, int iters, float seed) {
    float a[8];
    pk2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = pk2{a[i], a[i] + 0.5f}; }
    const float m = 0.999f, c = 1e-3f;
    const pk2 pm = {m, m}, pc = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);                                   // v_fma_f32
            if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pc);                      // v_pk_fma_f32
            if (KIND == 2) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.5f - 1.0f;                   // v_exp_f32 + v_fma
            if (KIND == 3) p[i] = p[i] * pm;                                                    // v_pk_mul_f32
            if (KIND == 4) p[i] = p[i] + pc;                                                    // v_pk_add_f32
            if (KIND == 5) a[i] = fminf(a[i] * m, 0.99f);                                       // v_mul + v_min
            if (KIND == 6) a[i] = __builtin_amdgcn_exp2f(a[i]);                                 // v_exp_f32 alone (saturates fast, still issues)
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
}

// broadcast LDS reads: every lane reads the same 16 bytes (what the blend does 3x per Gaussian)
__global__ __launch_bounds__(64) void k_lds_bcast(float* out, int iters) {
    __shared__ float4 buf[64];
    buf[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = buf[(it + i) & 63];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

extern "C" int ubench_launch(int kind, int blocks, int iters, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(64), 0, s, out, iters, 1.0f); break;
        case 6: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(64), 0, s, out, iters, -1.0f); break;
        case 7: hipLaunchKernelGGL(k_lds_bcast, dim3(blocks), dim3(64), 0, s, out, iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
