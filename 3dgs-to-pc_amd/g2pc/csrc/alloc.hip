// Point allocation kernels: distribute_points (gauss_to_pc.py:73-90) and bincount (gauss_to_pc.py:110).
#include "g2pc_internal.h"

namespace g2pc {

constexpr int AL_T = 256;

// deterministic two-level f64 sum: per-block partials, then one block over the partials
__global__ __launch_bounds__(AL_T) void k_sum_f64_partial(const double* __restrict__ x, long n,
                                                         double* __restrict__ partial) {
    __shared__ double ws[AL_T / kWave];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * AL_T + threadIdx.x; i < n; i += (long)gridDim.x * AL_T) acc += x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < AL_T / kWave; ++i) t += ws[i];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(AL_T) void k_sum_f64_final(const double* __restrict__ partial, int nb,
                                                       double* __restrict__ total) {
    __shared__ double ws[AL_T / kWave];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += AL_T) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < AL_T / kWave; ++i) t += ws[i];
        *total = t;
    }
}

// ppg = round_half_even(sizes * (num_points / total)); zero flags for the fill; per-block sums of ppg
__global__ __launch_bounds__(AL_T) void k_round_ppg(const double* __restrict__ sizes, long n, double num_points,
                                                   const double* __restrict__ total, double* __restrict__ ppg,
                                                   uint32_t* __restrict__ zero_flag,
                                                   double* __restrict__ partial_sum) {
    __shared__ double ws[AL_T / kWave];
    const double ratio = num_points / *total;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * AL_T + threadIdx.x; i < n; i += (long)gridDim.x * AL_T) {
        double v = rint(sizes[i] * ratio);
        ppg[i] = v;
        zero_flag[i] = (v == 0.0) ? 1u : 0u;
        acc += v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < AL_T / kWave; ++i) t += ws[i];
        partial_sum[blockIdx.x] = t;
    }
}

// zero fill with the reference's slice semantics: zero_indices[:k], k = int(min(num_points - sum, #zeros));
// a negative k keeps all but the last |k| zero entries (python negative slice).
__global__ __launch_bounds__(AL_T) void k_fill_zeros(double* __restrict__ ppg, int32_t* __restrict__ ppg_i32,
                                                    double* __restrict__ ppg_out, long n,
                                                    const uint32_t* __restrict__ zero_rank, double num_points,
                                                    const double* __restrict__ sum_ppg,
                                                    int64_t* __restrict__ stats) {
    const double sum = *sum_ppg;
    const long zeros = (long)zero_rank[n];
    double kd = num_points - sum;
    if ((double)zeros < kd) kd = (double)zeros;
    const long k = (long)kd;                       // int(): truncation toward zero
    long fill = k >= 0 ? k : (zeros + k > 0 ? zeros + k : 0);
    int local_max = 0;
    for (long i = (long)blockIdx.x * AL_T + threadIdx.x; i < n; i += (long)gridDim.x * AL_T) {
        double v = ppg[i];
        if (v == 0.0 && (long)zero_rank[i] < fill) v = 1.0;
        if (ppg_out) ppg_out[i] = v;
        int iv = (int)v;                           // .type(torch.int)
        ppg_i32[i] = iv;
        local_max = iv > local_max ? iv : local_max;
    }
    // one atomic per block (a per-wave atomic on a single address serialises: 98 us at 1 M Gaussians)
    __shared__ unsigned wmax[AL_T / kWave];
    unsigned m = wave_max_u32((unsigned)local_max);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned bm = 0;
        for (int i = 0; i < AL_T / kWave; ++i) bm = wmax[i] > bm ? wmax[i] : bm;
        if (bm > 0) atomicMax((unsigned long long*)&stats[3], (unsigned long long)bm);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[0] = (int64_t)sum;
        stats[1] = zeros;
        stats[2] = k;
    }
}

__global__ __launch_bounds__(AL_T) void k_bincount(const int32_t* __restrict__ v, long n,
                                                  uint32_t* __restrict__ hist, long hist_len) {
    constexpr int LOCAL = 2048;
    __shared__ uint32_t lh[LOCAL];
    for (int i = threadIdx.x; i < LOCAL; i += AL_T) lh[i] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * AL_T + threadIdx.x; i < n; i += (long)gridDim.x * AL_T) {
        int x = v[i];
        if (x < 0 || x >= hist_len) continue;
        if (x < LOCAL) atomicAdd(&lh[x], 1u);
        else atomicAdd(&hist[x], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LOCAL && i < hist_len; i += AL_T) {
        uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

static unsigned stream_grid(long n) {
    unsigned nb = cdiv(n, AL_T);
    return nb > 2048 ? 2048 : (nb < 1 ? 1 : nb);
}

}  // namespace g2pc

extern "C" {
size_t g2pc_distribute_points_workspace(int64_t n) {
    using namespace g2pc;
    return align_up(2048 * sizeof(double)) * 2 + align_up(4 * sizeof(double)) + align_up((size_t)(n + 1) * 4) * 2 +
           align_up((size_t)n * sizeof(double)) + scan_workspace(n) + 1024;
}

int g2pc_distribute_points(const double* sizes, int64_t n, int64_t num_points, double* ppg_f64, int32_t* ppg_i32,
                           int64_t* stats, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n > 0 && sizes && ppg_i32 && stats && ws, G2PC_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    double* partial = ar.get<double>(2048);
    double* partial2 = ar.get<double>(2048);
    double* scal = ar.get<double>(4);                 // [0] = total size, [1] = sum ppg
    uint32_t* zflag = ar.get<uint32_t>((size_t)n + 1);
    uint32_t* zrank = ar.get<uint32_t>((size_t)n + 1);
    double* ppg_tmp = ar.get<double>((size_t)n);
    size_t scan_bytes = scan_workspace(n);
    char* scan_ws = ar.get<char>(scan_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    unsigned nb = stream_grid(n);
    hipMemsetAsync(stats, 0, 4 * sizeof(int64_t), s);
    hipLaunchKernelGGL(k_sum_f64_partial, dim3(nb), dim3(AL_T), 0, s, sizes, (long)n, partial);
    hipLaunchKernelGGL(k_sum_f64_final, dim3(1), dim3(AL_T), 0, s, partial, (int)nb, scal);
    hipLaunchKernelGGL(k_round_ppg, dim3(nb), dim3(AL_T), 0, s, sizes, (long)n, (double)num_points, scal, ppg_tmp,
                       zflag, partial2);
    hipLaunchKernelGGL(k_sum_f64_final, dim3(1), dim3(AL_T), 0, s, partial2, (int)nb, scal + 1);
    int rc = scan_exclusive_u32(zflag, zrank, n, scan_ws, scan_bytes, s);
    if (rc) return rc;
    // grid-stride with at most 256 blocks: every block ends in one atomicMax on the same word (2 048 of them took 28 us)
    hipLaunchKernelGGL(k_fill_zeros, dim3(nb > 256 ? 256 : nb), dim3(AL_T), 0, s, ppg_tmp, ppg_i32, ppg_f64, (long)n, zrank,
                       (double)num_points, scal + 1, stats);
    return check_launch("g2pc_distribute_points");
}

int g2pc_bincount_i32(const int32_t* values, int64_t n, uint32_t* hist, int64_t hist_len, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && hist && hist_len > 0, G2PC_ERR_ARG, "bad arguments");
    if (n == 0) return G2PC_OK;
    unsigned nb = stream_grid(n);
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(k_bincount, dim3(nb), dim3(AL_T), 0, (hipStream_t)stream, values, (long)n, hist,
                       (long)hist_len);
    return check_launch("g2pc_bincount_i32");
}
}
