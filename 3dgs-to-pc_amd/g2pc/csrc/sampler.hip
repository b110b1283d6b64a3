// Point sampler: multivariate-normal draws with Mahalanobis rejection and first-k emission.
//
// Reference: gauss_to_pc.py:140-155 (MultivariateNormal: Cholesky + loc + L eps), :92-103 (Mahalanobis via
// the explicit inverse), :157-275 (attempt loop; emits the FIRST min(n - added, accepted) draws of every
// Gaussian, not the accepted ones), :277-371 (bin loop and output order).
//
// The reference walks bins on the host and re-concatenates the whole cloud per bin (torch.cat).  Here every
// bin is processed at once: Gaussians are stably partitioned by bin (radix sort on the bin id), a counting
// pass produces d[attempt][position] (small quotas: one Gaussian per lane -- the quota is uniform inside a bin, so
// the lanes of a wave run in lock-step; large quotas: one Gaussian per wave, 64 lanes striding over the draws,
// ballot/popcount for the accept count), a batched scan + a one-block section table turn d into output offsets
// ON THE DEVICE, and a row-balanced emission kernel (one output row per lane) regenerates the keyed Philox draws
// and writes every point straight to its final place in the reference's order.
#include "g2pc_internal.h"

namespace g2pc {

constexpr int SM_T = 256;

struct GaussSample {
    float mx, my, mz;
    float l00, l10, l11, l20, l21, l22;        // Cholesky factor (lower)
    float i00, i01, i02, i11, i12, i22;        // inverse covariance (symmetric)
};

__device__ __forceinline__ void load_gauss(const float* __restrict__ means, const float* __restrict__ cov9,
                                           unsigned g, GaussSample& s) {
    s.mx = means[3 * (size_t)g + 0];
    s.my = means[3 * (size_t)g + 1];
    s.mz = means[3 * (size_t)g + 2];
    const float* c = cov9 + 9 * (size_t)g;
    float a00 = c[0], a01 = c[1], a02 = c[2], a10 = c[3], a11 = c[4], a12 = c[5], a20 = c[6], a21 = c[7], a22 = c[8];
    // torch.linalg.cholesky reads the lower triangle
    s.l00 = sqrtf(a00);
    s.l10 = a10 / s.l00;
    s.l20 = a20 / s.l00;
    s.l11 = sqrtf(a11 - s.l10 * s.l10);
    s.l21 = (a21 - s.l20 * s.l10) / s.l11;
    s.l22 = sqrtf(a22 - s.l20 * s.l20 - s.l21 * s.l21);
    // torch.inverse (general 3x3): cofactor form
    float c00 = a11 * a22 - a12 * a21;
    float c01 = a02 * a21 - a01 * a22;
    float c02 = a01 * a12 - a02 * a11;
    float c10 = a12 * a20 - a10 * a22;
    float c11 = a00 * a22 - a02 * a20;
    float c12 = a02 * a10 - a00 * a12;
    float c20 = a10 * a21 - a11 * a20;
    float c21 = a01 * a20 - a00 * a21;
    float c22 = a00 * a11 - a01 * a10;
    float det = a00 * c00 + a01 * c10 + a02 * c20;
    float id = 1.0f / det;
    s.i00 = c00 * id;
    s.i01 = 0.5f * (c01 + c10) * id;
    s.i02 = 0.5f * (c02 + c20) * id;
    s.i11 = c11 * id;
    s.i12 = 0.5f * (c12 + c21) * id;
    s.i22 = c22 * id;
}

// draw k of (gid, attempt): sample point and accept flag
__device__ __forceinline__ bool draw(const GaussSample& s, unsigned seed_lo, unsigned seed_hi, unsigned gid_lo,
                                     unsigned gid_hi, unsigned attempt, unsigned k, float std_limit, float& px,
                                     float& py, float& pz) {
    Normal3 e = keyed_normal3(seed_lo, seed_hi, gid_lo, gid_hi, attempt, k);
    float ox = s.l00 * e.x;
    float oy = s.l10 * e.x + s.l11 * e.y;
    float oz = s.l20 * e.x + s.l21 * e.y + s.l22 * e.z;
    px = s.mx + ox;
    py = s.my + oy;
    pz = s.mz + oz;
    float dx = s.mx - px, dy = s.my - py, dz = s.mz - pz;        // reference: delta = means - samples
    float yx = s.i00 * dx + s.i01 * dy + s.i02 * dz;
    float yy = s.i01 * dx + s.i11 * dy + s.i12 * dz;
    float yz = s.i02 * dx + s.i12 * dy + s.i22 * dz;
    float m = dx * yx + dy * yy + dz * yz;
    return sqrtf(m) <= std_limit;                                  // NaN (m < 0) rejects, as in the reference
}

__global__ __launch_bounds__(SM_T) void k_bin_keys(const int32_t* __restrict__ ppg, long g,
                                                  const int32_t* __restrict__ lut, long lut_len, int num_bins,
                                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    long i = (long)blockIdx.x * SM_T + threadIdx.x;
    if (i >= g) return;
    int v = ppg[i];
    int b = (v >= 0 && v < lut_len) ? lut[v] : -1;
    keys[i] = (b < 0 || b >= num_bins) ? (uint32_t)num_bins : (uint32_t)b;
    vals[i] = (uint32_t)i;
}

// ---- counting pass ------------------------------------------------------------------------------------
__global__ __launch_bounds__(SM_T) void k_count_thread(const float* __restrict__ means,
                                                      const float* __restrict__ cov9,
                                                      const uint32_t* __restrict__ perm,
                                                      const uint32_t* __restrict__ pbin,
                                                      const int32_t* __restrict__ quota, long p_end, long gv,
                                                      float std_limit, int attempt0, int num_attempts,
                                                      unsigned seed_lo, unsigned seed_hi, uint64_t gid_base,
                                                      uint32_t* __restrict__ added, uint32_t* __restrict__ dcount,
                                                      uint32_t* __restrict__ remaining, float* __restrict__ stage,
                                                      uint32_t* __restrict__ hb) {
    // stage (draw-once sampling, G2pcSampleStage): every draw that may be emitted -- k < room, the attempt emits its FIRST
    // d <= room draws -- is kept at row have + k of its position, plane-major ([row][p_end][3]: the lanes of a wave write 768
    // contiguous bytes per row); hb[attempt][p] = rows the position had before the attempt.  The emission then copies.
    long p = (long)blockIdx.x * SM_T + threadIdx.x;
    if (p >= p_end) return;
    const unsigned g = perm[p];
    const int n = quota[pbin[p]] - 1;
    unsigned have = added[p];
    if (n <= 0 || have >= (unsigned)n) {
        for (int a = 0; a < num_attempts; ++a) dcount[(size_t)a * gv + p] = 0;
        return;
    }
    GaussSample s;
    load_gauss(means, cov9, g, s);
    const uint64_t gid = gid_base + g;
    const unsigned gid_lo = (unsigned)gid, gid_hi = (unsigned)(gid >> 32);
    for (int a = 0; a < num_attempts; ++a) {
        unsigned d = 0;
        if (hb) hb[(size_t)a * gv + p] = have;
        if (have < (unsigned)n) {
            unsigned acc = 0;
            const unsigned room = (unsigned)n - have;
            // d = min(accepted, room): once `room` draws are accepted the rest of the attempt cannot change d (every attempt
            // after the first has room for a quarter of the quota only)
            for (int k = 0; k < n && acc < room; ++k) {
                float x, y, z;
                acc += draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), (unsigned)k, std_limit, x, y, z) ? 1u : 0u;
                if (stage && (unsigned)k < room) {
                    float* dst = stage + ((size_t)(have + (unsigned)k) * (size_t)p_end + (size_t)p) * 3;
                    dst[0] = x; dst[1] = y; dst[2] = z;
                }
            }
            d = acc < room ? acc : room;
            have += d;
        }
        dcount[(size_t)a * gv + p] = d;
    }
    added[p] = have;
    if (have < (unsigned)n) atomicAdd(remaining, 1u);
}

__global__ __launch_bounds__(SM_T) void k_count_wave(const float* __restrict__ means,
                                                    const float* __restrict__ cov9,
                                                    const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ pbin,
                                                    const int32_t* __restrict__ quota, long p_begin, long gv,
                                                    float std_limit, int attempt0, int num_attempts,
                                                    unsigned seed_lo, unsigned seed_hi, uint64_t gid_base,
                                                    uint32_t* __restrict__ added, uint32_t* __restrict__ dcount,
                                                    uint32_t* __restrict__ remaining, float* __restrict__ stage,
                                                    uint32_t* __restrict__ hb, const uint32_t* __restrict__ bin_start,
                                                    const unsigned long long* __restrict__ wave_row_start) {
    // stage: Gaussian-major rows ([wave_row_start[bin] + (p - bin_start[bin]) * (quota - 1) + row][3]); the lanes of the wave
    // hold consecutive draws, i.e. consecutive rows
    const unsigned lane = threadIdx.x & 63;
    long p = p_begin + (long)blockIdx.x * (SM_T / kWave) + (threadIdx.x >> 6);
    if (p >= gv) return;                                   // whole wave leaves together
    const unsigned g = perm[p];
    const unsigned bin = pbin[p];
    const int n = quota[bin] - 1;
    unsigned have = added[p];
    if (n <= 0 || have >= (unsigned)n) {
        if (lane == 0) for (int a = 0; a < num_attempts; ++a) dcount[(size_t)a * gv + p] = 0;
        return;
    }
    GaussSample s;
    load_gauss(means, cov9, g, s);
    const uint64_t gid = gid_base + g;
    const unsigned gid_lo = (unsigned)gid, gid_hi = (unsigned)(gid >> 32);
    float* rows = stage ? stage + 3 * (size_t)(wave_row_start[bin] + (unsigned long long)(p - (long)bin_start[bin]) * (unsigned long long)n) : nullptr;
    for (int a = 0; a < num_attempts; ++a) {
        unsigned d = 0;
        if (hb && lane == 0) hb[(size_t)a * gv + p] = have;
        if (have < (unsigned)n) {
            unsigned acc = 0;
            const unsigned room = (unsigned)n - have;
            // d = min(accepted, room): once `room` draws are accepted the rest of the attempt cannot change d
            for (int k0 = 0; k0 < n && acc < room; k0 += 64) {
                int k = k0 + (int)lane;
                float x, y, z;
                bool ok = false;
                if (k < n) {
                    ok = draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), (unsigned)k, std_limit, x, y, z);
                    if (rows && (unsigned)k < room) {
                        float* dst = rows + 3 * (size_t)(have + (unsigned)k);
                        dst[0] = x; dst[1] = y; dst[2] = z;
                    }
                }
                acc += (unsigned)__popcll(__ballot(ok));
            }
            d = acc < room ? acc : room;
            have += d;
        }
        if (lane == 0) dcount[(size_t)a * gv + p] = d;
    }
    if (lane == 0) {
        added[p] = have;
        if (have < (unsigned)n) atomicAdd(remaining, 1u);
    }
}

// ---- device-side bin table ----------------------------------------------------------------------------------------
// The bin heuristics of generate_pointcloud / calculate_bin_sizes (gauss_to_pc.py:105-138, :308-337) on the histogram of
// points per Gaussian, in ONE block: until round 3 the histogram went to the host, numpy built the table and the look-up
// table came back (a round trip plus ~0.2 ms of host work in the middle of a 1 ms job).  The arithmetic is numpy's, value
// for value: np.gradient twice in float64 (the second differences of integer counts are multiples of 1/4, so every sum
// below is exact whatever its order), bin_size = max(D // 100, 1), the group sums, cut = max // 50 (floor division),
// start_bin = first group at or after the peak below the cut -- counted FROM the peak, as the reference does --, the
// tail rounded up to multiples of bin_size, quota = floor(s + (e - s) / 2).  Work arrays live in the caller's workspace.
constexpr int BT_T = 1024, BT_PER = 8, BT_MAX = BT_T * BT_PER;      // histograms of up to 8 192 entries
struct BinPlan {            // what the host needs back (pinned memory), all int64
    int64_t num_bins, gv, p_wave, any_sampling, means_rows, rows_ub, error, start_bin, bin_size, distinct;
    int64_t lane_planes, reserved;     // ABI 7: the largest quota - 1 of a lane-mode bin (rows of G2pcSampleStage.thread_rows)
};
// exclusive scan of one value per (thread, k) in thread-major order; returns the total.  s_w: BT_T / 64 + 1 words
__device__ __forceinline__ uint32_t block_excl_scan8(const uint32_t v[BT_PER], uint32_t out[BT_PER], uint32_t* s_w) {
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) sum += v[k];
    const uint32_t incl = wave_incl_scan_u32(sum);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - sum, total = 0;
    for (int w = 0; w < BT_T / kWave; ++w) { const uint32_t t = s_w[w]; if (w < (int)(threadIdx.x >> 6)) off += t; total += t; }
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) { out[k] = off; off += v[k]; }
    return total;
}

__global__ __launch_bounds__(BT_T) void k_bin_table(const uint32_t* __restrict__ hist, int hist_len,
                                                   const int64_t* __restrict__ stats /* [3] = max points per Gaussian */,
                                                   int exact, int emit_means, int wave_min_draws,
                                                   int32_t* __restrict__ lut, int32_t* __restrict__ quota,
                                                   uint32_t* __restrict__ bin_start, int32_t* __restrict__ bin_lo,
                                                   int32_t* __restrict__ val, uint32_t* __restrict__ cnt,
                                                   double* __restrict__ g1, double* __restrict__ g2,
                                                   int32_t* __restrict__ pd, uint32_t* __restrict__ members,
                                                   BinPlan* __restrict__ plan_host) {
    __shared__ uint32_t s_w[BT_T / kWave + 1];
    __shared__ int s_i[8];
    __shared__ unsigned long long s_rows[2];
    __shared__ int s_planes;
    const int tid = (int)threadIdx.x;
    const long max_ppg = stats ? (long)stats[3] : (long)hist_len - 1;
    if (max_ppg >= hist_len || hist_len > BT_MAX) {                 // histogram too short for this job: the host takes over
        if (tid == 0) { BinPlan p{}; p.error = 1; *plan_host = p; }
        return;
    }
    const int HL = (int)max_ppg + 1;
    // (1) the distinct values (ascending) and their counts
    uint32_t f[BT_PER], o[BT_PER];
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) {
        const int v = tid * BT_PER + k;
        f[k] = (v < HL && hist[v] != 0u) ? 1u : 0u;
        if (v < hist_len) lut[v] = -1;             // (clears the look-up table: every v is visited by exactly one (thread, k))
    }
    const int D = (int)block_excl_scan8(f, o, s_w);
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) { const int v = tid * BT_PER + k; if (f[k]) { val[o[k]] = v; cnt[o[k]] = hist[v]; } }
    __syncthreads();
    int start_bin = D, bin_size = 1;                               // exact mode: every distinct value is a bin
    if (!exact) {
        if (D < 2) {                                               // np.gradient needs two samples: the reference raises here
            if (tid == 0) { BinPlan p{}; p.error = 2; p.distinct = D; *plan_host = p; }
            return;
        }
        for (int i = tid; i < D; i += BT_T) {
            const double a = (double)cnt[i > 0 ? i - 1 : 0], b = (double)cnt[i], c = (double)cnt[i < D - 1 ? i + 1 : D - 1];
            g1[i] = i == 0 ? c - b : (i == D - 1 ? b - a : (c - a) / 2.0);
        }
        __syncthreads();
        for (int i = tid; i < D; i += BT_T) {
            const double a = g1[i > 0 ? i - 1 : 0], b = g1[i], c = g1[i < D - 1 ? i + 1 : D - 1];
            g2[i] = fabs(i == 0 ? c - b : (i == D - 1 ? b - a : (c - a) / 2.0));
        }
        __syncthreads();
        bin_size = D / 100 > 1 ? D / 100 : 1;
        const int K = D / bin_size;                                // groups (the remainder of g2 is dropped)
        for (int k = tid; k < K; k += BT_T) {
            double acc = 0.0;
            for (int j = 0; j < bin_size; ++j) acc += g2[k * bin_size + j];
            g1[k] = acc;                                           // (g1 is free again: the group sums)
        }
        __syncthreads();
        if (tid == 0) {                                            // K <= 199: peak, cut, first group below the cut
            int peak = 0;
            double mx = g1[0];
            for (int k = 1; k < K; ++k) if (g1[k] > mx) { mx = g1[k]; peak = k; }
            const double cut = (mx - fmod(mx, 50.0)) / 50.0;       // np.max(sums) // 50 (non-negative operands)
            int sb = 1;
            for (int k = peak; k < K; ++k) if (g1[k] < cut) { sb = k - peak; break; }
            s_i[0] = sb;
        }
        __syncthreads();
        start_bin = s_i[0] < D ? s_i[0] : D;
    }
    // (2) pd = val[:start_bin] ++ unique(ceil(val[start_bin:] / bin_size)) * bin_size
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) {
        const int i = tid * BT_PER + k;
        uint32_t keep = 0;
        if (i < D) {
            if (i < start_bin) keep = 1;
            else {
                const int t = (val[i] + bin_size - 1) / bin_size;
                keep = (i == start_bin || (val[i - 1] + bin_size - 1) / bin_size != t) ? 1u : 0u;
            }
        }
        f[k] = keep;
    }
    const int B = (int)block_excl_scan8(f, o, s_w);
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) {
        const int i = tid * BT_PER + k;
        if (f[k]) pd[o[k]] = i < start_bin ? val[i] : (val[i] + bin_size - 1) / bin_size * bin_size;
    }
    if (tid == 0) { s_i[1] = 0x7FFFFFFF; s_i[2] = 0; s_rows[0] = 0ull; s_rows[1] = 0ull; s_planes = 0; }
    __syncthreads();
    // (3) bins (start, end, quota), their members, the look-up table
    for (int b = tid; b < B; b += BT_T) {
        const int sv = pd[b], ev = b != B - 1 ? pd[b + 1] : sv + 1;
        const int q = sv + (ev - sv) / 2;                          // floor(s + (e - s) / 2), integers with e > s
        int lo = sv, hi = ev < HL ? ev : HL;
        uint32_t m = 0;
        if (q > 0 && hi > lo) for (int v = lo; v < hi; ++v) { lut[v] = b; m += hist[v]; }
        quota[b] = q;
        members[b] = m;
        bin_lo[b] = sv;
    }
    __syncthreads();
    // (4) bin_start = exclusive scan of the members; totals
    uint32_t mv[BT_PER], mo[BT_PER];
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) { const int b = tid * BT_PER + k; mv[k] = b < B ? members[b] : 0u; }
    const uint32_t gv = block_excl_scan8(mv, mo, s_w);
    unsigned long long means = 0ull, rows = 0ull;
    int any_sampling = 0, first_wave = 0x7FFFFFFF;
#pragma unroll
    for (int k = 0; k < BT_PER; ++k) {
        const int b = tid * BT_PER + k;
        if (b < B) {
            bin_start[b] = mo[k];
            const int q = quota[b];
            if (mv[k]) {
                if (q > 0) means += mv[k];
                if (q > 1) { rows += (unsigned long long)mv[k] * (unsigned long long)(q - 1); any_sampling = 1; }
                if (q - 1 >= wave_min_draws && b < first_wave) first_wave = b;
                if (q - 1 < wave_min_draws && q - 1 > 0) atomicMax(&s_planes, q - 1);
            }
        }
    }
    if (means) atomicAdd(&s_rows[0], means);
    if (rows) atomicAdd(&s_rows[1], rows);
    if (any_sampling) atomicOr(&s_i[2], 1);
    if (first_wave != 0x7FFFFFFF) atomicMin(&s_i[1], first_wave);
    __syncthreads();
    if (tid == 0) {
        bin_start[B] = gv; bin_start[B + 1] = gv;
        BinPlan p{};
        p.num_bins = B; p.gv = gv; p.any_sampling = s_i[2];
        p.means_rows = emit_means ? (int64_t)s_rows[0] : 0;
        p.rows_ub = p.means_rows + (int64_t)s_rows[1];
        p.start_bin = start_bin; p.bin_size = bin_size; p.distinct = D;
        p.lane_planes = s_planes;
        *plan_host = p;                                            // p_wave below needs bin_start of another thread's bin
    }
    __syncthreads();
    if (tid == 0) plan_host->p_wave = s_i[1] != 0x7FFFFFFF ? (int64_t)bin_start[s_i[1]] : (int64_t)gv;
}

// ---- device-side section table ------------------------------------------------------------------------------------
// The output is a sequence of sections in the reference's order: bin-major, and inside a bin the means (one row per
// member) followed by the rows attempt 0, 1, ... emitted for the bin's members.  sec_base[b * (1 + A) + s] = first output
// row of section s of bin b (exclusive scan of the section sizes); sec_base[B * (1 + A)] = M, the number of points.
// One block: the table has B * (1 + A) entries (a few thousand at most).
constexpr int SEC_T = 1024;
__global__ __launch_bounds__(SEC_T) void k_sections(const uint32_t* __restrict__ bin_start,
                                                    const int32_t* __restrict__ quota, int B, int A,
                                                    const uint32_t* __restrict__ dscan, long gv, int emit_means,
                                                    int64_t* __restrict__ sec_base, int64_t* __restrict__ info_host,
                                                    const uint32_t* __restrict__ remaining) {
    __shared__ uint32_t wsum[SEC_T / kWave];
    __shared__ unsigned long long carry;
    const int S = B * (1 + A);
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (int base = 0; base < S; base += SEC_T) {
        const int i = base + (int)threadIdx.x;
        uint32_t v = 0;
        if (i < S) {
            const int b = i / (1 + A), sct = i % (1 + A);
            const uint32_t p0 = bin_start[b], p1 = bin_start[b + 1];
            if (sct == 0) v = (quota[b] > 0 && emit_means) ? (p1 - p0) : 0u;
            else { const uint32_t* sc = dscan + (size_t)(sct - 1) * (size_t)(gv + 1); v = sc[p1] - sc[p0]; }
        }
        const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t incl = wave_incl_scan_u32(v);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
        for (int k = 0; k < SEC_T / kWave; ++k) { const uint32_t t = wsum[k]; if (k < (int)w) woff += t; total += t; }
        const unsigned long long c = carry;
        if (i < S) sec_base[i] = (int64_t)(c + woff + incl - v);
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sec_base[S] = (int64_t)carry;
        if (info_host) {                               // pinned host memory through its device mapping
            info_host[0] = (int64_t)carry;
            info_host[1] = remaining ? (int64_t)*remaining : 0;
        }
    }
}

// ---- row-balanced emission ------------------------------------------------------------------------------------------
// One output row per lane: a block owns ER_ROWS consecutive output rows, finds the section(s) they fall into and, inside
// an attempt section, the Gaussian and draw index of every row (rows of a Gaussian are consecutive, the per-attempt scan
// of d gives their start) -- a window of the scan is staged in LDS for the searches.  Every lane then evaluates exactly
// one keyed draw and writes one 12-byte row next to its neighbours': all lanes busy whatever the quotas are (32-draw
// bins and 2000-draw bins alike), stores fully coalesced, no second pass over d for offsets.  The per-Gaussian inputs
// are gathered straight from global memory: neighbouring rows share their Gaussian, so a wave touches a handful of lines.
constexpr int ER_T = 256, ER_ROWS = 1024, ER_WIN = 1024;

struct GaussChol { float mx, my, mz, l00, l10, l11, l20, l21, l22; };
__device__ __forceinline__ void load_chol(const float* __restrict__ means, const float* __restrict__ cov9, unsigned g,
                                          GaussChol& s) {
    s.mx = means[3 * (size_t)g + 0]; s.my = means[3 * (size_t)g + 1]; s.mz = means[3 * (size_t)g + 2];
    const float* c = cov9 + 9 * (size_t)g;
    const float a00 = c[0], a10 = c[3], a11 = c[4], a20 = c[6], a21 = c[7], a22 = c[8];   // the lower triangle, as load_gauss
    s.l00 = sqrtf(a00);
    s.l10 = a10 / s.l00;
    s.l20 = a20 / s.l00;
    s.l11 = sqrtf(a11 - s.l10 * s.l10);
    s.l21 = (a21 - s.l20 * s.l10) / s.l11;
    s.l22 = sqrtf(a22 - s.l20 * s.l20 - s.l21 * s.l21);
}

// largest p in [lo, hi) with sc[p] <= t, for a non-decreasing sc with sc[lo] <= t (all threads of the block, same
// arguments, same result): 256-way narrowing, one probe per thread and step
__device__ __forceinline__ uint32_t block_search_le(const uint32_t* __restrict__ sc, uint32_t lo, uint32_t hi, uint32_t t) {
    while (hi - lo > 1) {
        const uint32_t span = hi - lo;
        const uint32_t step = (span + ER_T - 1) / ER_T;
        const uint32_t probe = lo + (threadIdx.x + 1) * step;               // probes lo+step, lo+2 step, ...
        const int ok = (probe < hi) && (sc[probe] <= t);
        const int cnt = __syncthreads_count(ok);                            // monotone: the first cnt probes hold
        const uint32_t nlo = lo + (uint32_t)cnt * step;
        const uint32_t nhi = nlo + step < hi ? nlo + step : hi;
        lo = nlo; hi = nhi;
    }
    return lo;
}

// 64 consecutive 12-byte rows of one wave -> global memory as three fully coalesced dword stores per array (a lane
// writing its own row would touch 12 cache lines per store instruction, a third of each)
__device__ __forceinline__ void wave_store_rows3(float* __restrict__ s_rows /* [192] wave-private */, float a, float b, float c,
                                                 bool valid, unsigned lane, unsigned cnt, float* __restrict__ dst /* row base */) {
    if (valid) { s_rows[3 * lane + 0] = a; s_rows[3 * lane + 1] = b; s_rows[3 * lane + 2] = c; }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const unsigned idx = j * 64 + lane;
        if (idx < 3 * cnt) dst[idx] = s_rows[idx];
    }
    wave_sync();
}

__global__ __launch_bounds__(ER_T) void k_emit_rows(
    const float* __restrict__ means, const float* __restrict__ cov9, const float* __restrict__ colours,
    const float* __restrict__ normals, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ bin_start, int B,
    int A, long gv, int attempt0, unsigned seed_lo, unsigned seed_hi, uint64_t gid_base,
    const uint32_t* __restrict__ dscan, const int64_t* __restrict__ sec_base, float* __restrict__ out_points,
    float* __restrict__ out_colours, float* __restrict__ out_normals, int32_t* __restrict__ out_gauss, long rows_capacity,
    const float* __restrict__ stage_t, long p_wave, const float* __restrict__ stage_w,
    const unsigned long long* __restrict__ wave_row_start, const uint32_t* __restrict__ hb, const int32_t* __restrict__ quota) {
    __shared__ uint32_t s_win[ER_WIN + 1];
    __shared__ float s_rows[ER_T / kWave][192];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int S = B * (1 + A);
    long M = (long)sec_base[S];
    if (M > rows_capacity) M = rows_capacity;          // never past the caller's arrays (the host compares M with its bound too)
    const long row0 = (long)blockIdx.x * ER_ROWS;
    if (row0 >= M) return;
    const long row1 = row0 + ER_ROWS < M ? row0 + ER_ROWS : M;
    // first section that ends after row0: sec_base is monotone, so a binary search (every thread finds the same index)
    int lo = 0, hi = S;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sec_base[mid + 1] <= (int64_t)row0) lo = mid + 1; else hi = mid;
    }
    int si = lo;
    for (; si < S; ++si) {
        const long sb = (long)sec_base[si], se = (long)sec_base[si + 1];
        if (sb >= row1) break;
        if (se <= sb) continue;
        const long r_lo = sb > row0 ? sb : row0, r_hi = se < row1 ? se : row1;
        const int b = si / (1 + A), sct = si % (1 + A);
        const uint32_t bs0 = bin_start[b], bs1 = bin_start[b + 1];
        const uint32_t* sc = nullptr;
        uint32_t sc0 = 0, p_first = 0, wlen = 0, win_end = 0;
        if (sct > 0) {
            sc = dscan + (size_t)(sct - 1) * (size_t)(gv + 1);
            sc0 = sc[bs0];
            // owner of the first row, then a window of the scan from there (relative to the section)
            const uint32_t t_first = sc0 + (uint32_t)(r_lo - sb);
            p_first = block_search_le(sc, bs0, bs1, t_first);
            wlen = (bs1 - p_first) < (uint32_t)ER_WIN ? (bs1 - p_first) : (uint32_t)ER_WIN;
            __syncthreads();                        // previous section's readers are done with the window
            for (uint32_t j = threadIdx.x; j <= wlen; j += ER_T) s_win[j] = sc[p_first + j];
            __syncthreads();
            win_end = s_win[wlen];                  // first scan value NOT covered by the window's owners
        }
        // every wave takes 64 consecutive rows per step; lanes past the section's end idle but join the staged stores
        for (long rb = r_lo + 64 * (long)w; rb < r_hi; rb += ER_T) {
            const long r = rb + lane;
            const bool valid = r < r_hi;
            const unsigned cnt = (unsigned)((r_hi - rb) < 64 ? (r_hi - rb) : 64);
            unsigned g = 0;
            float x = 0.f, y = 0.f, z = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
            if (valid) {
                if (sct == 0) {                     // the means of the bin's members, in member order
                    g = perm[bs0 + (uint32_t)(r - sb)];
                    x = means[3 * (size_t)g]; y = means[3 * (size_t)g + 1]; z = means[3 * (size_t)g + 2];
                    c0 = colours[3 * (size_t)g]; c1 = colours[3 * (size_t)g + 1]; c2 = colours[3 * (size_t)g + 2];
                    if (out_normals) { n0 = normals[3 * (size_t)g]; n1 = normals[3 * (size_t)g + 1]; n2 = normals[3 * (size_t)g + 2]; }
                } else {
                    const uint32_t t = sc0 + (uint32_t)(r - sb);
                    uint32_t p, start;
                    if (t < win_end) {              // largest j in [0, wlen) with s_win[j] <= t
                        uint32_t lo = 0, hi = wlen;
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_win[mid] <= t) lo = mid; else hi = mid; }
                        p = p_first + lo; start = s_win[lo];
                    } else {                        // long runs of finished Gaussians (d = 0): search the global scan
                        uint32_t lo = p_first + wlen, hi = bs1;
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sc[mid] <= t) lo = mid; else hi = mid; }
                        p = lo; start = sc[lo];
                    }
                    const unsigned k = t - start;
                    g = perm[p];
                    // every per-Gaussian input is requested before the first is used: one round trip, not one per field
                    c0 = colours[3 * (size_t)g]; c1 = colours[3 * (size_t)g + 1]; c2 = colours[3 * (size_t)g + 2];
                    if (out_normals) { n0 = normals[3 * (size_t)g]; n1 = normals[3 * (size_t)g + 1]; n2 = normals[3 * (size_t)g + 2]; }
                    if (hb) {
                        // draw-once: the count pass kept the point (G2pcSampleStage) -- row hb + k of position p
                        const uint32_t row = hb[(size_t)(sct - 1) * (size_t)gv + p] + k;
                        const float* src = (long)p < p_wave
                            ? stage_t + ((size_t)row * (size_t)p_wave + p) * 3
                            : stage_w + 3 * (size_t)(wave_row_start[b] + (unsigned long long)(p - bs0) * (unsigned long long)(quota[b] - 1) + row);
                        x = src[0]; y = src[1]; z = src[2];
                    } else {
                        GaussChol s;
                        load_chol(means, cov9, g, s);
                        const uint64_t gid = gid_base + g;
                        const Normal3 e = keyed_normal3(seed_lo, seed_hi, (unsigned)gid, (unsigned)(gid >> 32),
                                                        (unsigned)(attempt0 + sct - 1), k);
                        x = s.mx + s.l00 * e.x;
                        y = s.my + (s.l10 * e.x + s.l11 * e.y);
                        z = s.mz + (s.l20 * e.x + s.l21 * e.y + s.l22 * e.z);
                    }
                }
            }
            wave_store_rows3(s_rows[w], x, y, z, valid, lane, cnt, out_points + 3 * (size_t)rb);
            wave_store_rows3(s_rows[w], c0, c1, c2, valid, lane, cnt, out_colours + 3 * (size_t)rb);
            if (out_normals) wave_store_rows3(s_rows[w], n0, n1, n2, valid, lane, cnt, out_normals + 3 * (size_t)rb);
            if (out_gauss && valid) out_gauss[(size_t)r] = (int32_t)g;
        }
    }
}

// standalone mahalanobis() of the reference API (gauss_to_pc.py:92-103): one thread per (mean, sample, cov) row
__global__ __launch_bounds__(SM_T) void k_mahalanobis(const float* __restrict__ means,
                                                     const float* __restrict__ samples,
                                                     const float* __restrict__ cov9, long n,
                                                     float* __restrict__ out) {
    long i = (long)blockIdx.x * SM_T + threadIdx.x;
    if (i >= n) return;
    GaussSample s;
    load_gauss(means, cov9, (unsigned)i, s);
    float dx = s.mx - samples[3 * i], dy = s.my - samples[3 * i + 1], dz = s.mz - samples[3 * i + 2];
    float yx = s.i00 * dx + s.i01 * dy + s.i02 * dz;
    float yy = s.i01 * dx + s.i11 * dy + s.i12 * dz;
    float yz = s.i02 * dx + s.i12 * dy + s.i22 * dz;
    out[i] = sqrtf(dx * yx + dy * yy + dz * yz);
}

// standalone sample_from_multivariate_normal (gauss_to_pc.py:140-155): out[k, g, :] = mean_g + chol(cov_g) eps
__global__ __launch_bounds__(SM_T) void k_sample_mvn(const float* __restrict__ means,
                                                    const float* __restrict__ cov9, long g_count, int n,
                                                    unsigned seed_lo, unsigned seed_hi, uint64_t gid_base,
                                                    unsigned attempt, float* __restrict__ out) {
    long g = (long)blockIdx.x * SM_T + threadIdx.x;
    if (g >= g_count) return;
    GaussSample s;
    load_gauss(means, cov9, (unsigned)g, s);
    const uint64_t gid = gid_base + (uint64_t)g;
    for (int k = 0; k < n; ++k) {
        float x, y, z;
        draw(s, seed_lo, seed_hi, (unsigned)gid, (unsigned)(gid >> 32), attempt, (unsigned)k, 0.f, x, y, z);
        float* dst = out + 3 * ((size_t)k * g_count + g);
        dst[0] = x; dst[1] = y; dst[2] = z;
    }
}

// wave_row_start[b] = rows of the wave-mode bins before b (a bin is wave-mode when quota - 1 >= wave_min_draws; quotas ascend
// with the bin index, so they are the last bins); [num_bins] = their total.  One thread: a few hundred bins at most.
__global__ void k_stage_plan(const uint32_t* __restrict__ bin_start, const int32_t* __restrict__ quota, int B, int wave_min_draws,
                             unsigned long long* __restrict__ wave_row_start) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long acc = 0ull;
    for (int b = 0; b < B; ++b) {
        wave_row_start[b] = acc;
        const int q = quota[b] - 1;
        if (q >= wave_min_draws) acc += (unsigned long long)(bin_start[b + 1] - bin_start[b]) * (unsigned long long)q;
    }
    wave_row_start[B] = acc;
}

static int bits_for(unsigned v) { int b = 0; while ((1u << b) <= v && b < 31) ++b; return b < 1 ? 1 : b; }

}  // namespace g2pc

extern "C" {
size_t g2pc_sampler_plan_workspace(int64_t g) {
    using namespace g2pc;
    return align_up((size_t)g * 4) * 4 + sort_workspace(g) + scan_workspace(1 << 16) + 4096;
}

int g2pc_sampler_plan(const int32_t* ppg, int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, int32_t num_bins,
                      uint32_t* perm, uint32_t* pbin, uint32_t* bin_start, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(g > 0 && ppg && bin_of_ppg && perm && pbin && bin_start && ws, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(num_bins >= 0 && num_bins < (1 << 16), G2PC_ERR_UNSUPPORTED, "more than 65535 bins");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    uint32_t* keys = ar.get<uint32_t>((size_t)g);
    uint32_t* vals = ar.get<uint32_t>((size_t)g);
    uint32_t* ktmp = ar.get<uint32_t>((size_t)g);
    uint32_t* vtmp = ar.get<uint32_t>((size_t)g);
    size_t sort_bytes = sort_workspace(g);
    char* sort_ws = ar.get<char>(sort_bytes);
    size_t scan_bytes = scan_workspace(num_bins + 1);
    char* scan_ws = ar.get<char>(scan_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_bin_keys, dim3(cdiv(g, SM_T)), dim3(SM_T), 0, s, ppg, (long)g, bin_of_ppg, (long)lut_len,
                       (int)num_bins, keys, vals);
    int rc = sort_pairs_u32(keys, vals, pbin, perm, ktmp, vtmp, g, 0, bits_for((unsigned)num_bins), sort_ws,
                            sort_bytes, s);
    if (rc) return rc;
    // bin_start = exclusive scan of the per-bin member counts (bin id num_bins = "no bin")
    hipMemsetAsync(bin_start, 0, (size_t)(num_bins + 2) * sizeof(uint32_t), s);
    rc = g2pc_bincount_i32((const int32_t*)pbin, g, bin_start, num_bins + 1, stream);
    if (rc) return rc;
    rc = scan_exclusive_u32(bin_start, bin_start, num_bins + 1, scan_ws, scan_bytes, s);
    if (rc) return rc;
    return check_launch("g2pc_sampler_plan");
}

size_t g2pc_sampler_bin_table_workspace(int64_t hist_len) {
    using namespace g2pc;
    const size_t n = (size_t)(hist_len > 0 ? hist_len : 0) + 2;
    return align_up(n * 4) * 4 + align_up(n * 8) * 2 + 1024;
}

/* The bin table on the device (see k_bin_table): hist u32[hist_len] = bincount of the points per Gaussian, stats = the
 * device-side result of g2pc_distribute_points ([3] = max points per Gaussian; NULL: hist_len - 1).  Out: lut i32[hist_len]
 * (bin of every point count, -1: none), quota i32[hist_len], bin_start u32[hist_len + 2], bin_lo i32[hist_len] (first
 * point count of every bin), plan_host (PINNED host memory, i64[12]: bins, Gaussians in bins, first wave-mode position,
 * any sampling, mean rows, row bound, error, start_bin, bin_size, distinct counts) written through its device mapping.
 * error 1: some Gaussian has hist_len or more points (build a longer histogram); 2: fewer than two distinct point counts in
 * binned mode (the reference's np.gradient raises). */
int g2pc_sampler_bin_table(const uint32_t* hist, int64_t hist_len, const int64_t* stats, int32_t exact, int32_t emit_means,
                           int32_t wave_min_draws, int32_t* lut, int32_t* quota, uint32_t* bin_start, int32_t* bin_lo,
                           int64_t* plan_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(hist && hist_len > 0 && lut && quota && bin_start && bin_lo && plan_host && ws, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(hist_len <= BT_MAX, G2PC_ERR_UNSUPPORTED, "histograms of more than 8192 entries go through the host");
    static_assert(sizeof(BinPlan) == 12 * sizeof(int64_t), "BinPlan is twelve int64");
    Arena ar(ws, ws_bytes);
    const size_t n = (size_t)hist_len + 2;
    int32_t* val = ar.get<int32_t>(n);
    uint32_t* cnt = ar.get<uint32_t>(n);
    int32_t* pd = ar.get<int32_t>(n);
    uint32_t* members = ar.get<uint32_t>(n);
    double* g1 = ar.get<double>(n);
    double* g2 = ar.get<double>(n);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_bin_table, dim3(1), dim3(BT_T), 0, (hipStream_t)stream, hist, (int)hist_len, stats, (int)exact,
                       (int)emit_means, (int)wave_min_draws, lut, quota, bin_start, bin_lo, val, cnt, g1, g2, pd, members,
                       (BinPlan*)plan_host);
    return check_launch("g2pc_sampler_bin_table");
}

/* The partition alone (bin keys + stable radix sort): perm / pbin as g2pc_sampler_plan, for callers that know the bin
 * sizes already (the host derives them from the same histogram it builds the bin table from) and upload bin_start. */
int g2pc_sampler_partition(const int32_t* ppg, int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, int32_t num_bins,
                           uint32_t* perm, uint32_t* pbin, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(g > 0 && ppg && bin_of_ppg && perm && pbin && ws, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(num_bins >= 0 && num_bins < (1 << 16), G2PC_ERR_UNSUPPORTED, "more than 65535 bins");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    uint32_t* keys = ar.get<uint32_t>((size_t)g);
    uint32_t* vals = ar.get<uint32_t>((size_t)g);
    uint32_t* ktmp = ar.get<uint32_t>((size_t)g);
    uint32_t* vtmp = ar.get<uint32_t>((size_t)g);
    size_t sort_bytes = sort_workspace(g);
    char* sort_ws = ar.get<char>(sort_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_bin_keys, dim3(cdiv(g, SM_T)), dim3(SM_T), 0, s, ppg, (long)g, bin_of_ppg, (long)lut_len,
                       (int)num_bins, keys, vals);
    int rc = sort_pairs_u32(keys, vals, pbin, perm, ktmp, vtmp, g, 0, bits_for((unsigned)num_bins), sort_ws, sort_bytes, s);
    if (rc) return rc;
    return check_launch("g2pc_sampler_partition");
}

static int sampler_count_impl(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                              const int32_t* quota, int64_t gv, int64_t p_wave_begin, float std_limit, int32_t attempt0,
                              int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added, uint32_t* dcount,
                              uint32_t* remaining, const G2pcSampleStage* st, uint32_t* hb, const uint32_t* bin_start,
                              void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(gv >= 0 && means && cov9 && perm && pbin && quota && added && dcount && remaining, G2PC_ERR_ARG,
                 "bad arguments");
    if (gv == 0) return G2PC_OK;
    if (p_wave_begin < 0 || p_wave_begin > gv) p_wave_begin = gv;
    hipStream_t s = (hipStream_t)stream;
    unsigned slo = (unsigned)seed, shi = (unsigned)(seed >> 32);
    if (p_wave_begin > 0)
        hipLaunchKernelGGL(k_count_thread, dim3(cdiv(p_wave_begin, SM_T)), dim3(SM_T), 0, s, means, cov9, perm, pbin,
                           quota, (long)p_wave_begin, (long)gv, std_limit, (int)attempt0, (int)num_attempts, slo, shi,
                           gid_base, added, dcount, remaining, st ? st->thread_rows : nullptr, hb);
    if (p_wave_begin < gv)
        hipLaunchKernelGGL(k_count_wave, dim3(cdiv(gv - p_wave_begin, SM_T / kWave)), dim3(SM_T), 0, s, means, cov9,
                           perm, pbin, quota, (long)p_wave_begin, (long)gv, std_limit, (int)attempt0,
                           (int)num_attempts, slo, shi, gid_base, added, dcount, remaining, st ? st->wave_rows : nullptr, hb,
                           bin_start, st ? (const unsigned long long*)st->wave_row_start : nullptr);
    return check_launch("g2pc_sampler_count");
}
int g2pc_sampler_count(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                       const int32_t* quota, int64_t gv, int64_t p_wave_begin, float std_limit, int32_t attempt0,
                       int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added, uint32_t* dcount,
                       uint32_t* remaining, void* stream) {
    return sampler_count_impl(means, cov9, perm, pbin, quota, gv, p_wave_begin, std_limit, attempt0, num_attempts, seed, gid_base,
                              added, dcount, remaining, nullptr, nullptr, nullptr, stream);
}
int g2pc_sampler_stage_plan(const uint32_t* bin_start, const int32_t* quota, int32_t num_bins, int32_t wave_min_draws,
                            uint64_t* wave_row_start, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(bin_start && quota && wave_row_start && num_bins >= 0, G2PC_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(k_stage_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, bin_start, quota, (int)num_bins,
                       (int)wave_min_draws, (unsigned long long*)wave_row_start);
    return check_launch("g2pc_sampler_stage_plan");
}
int g2pc_sampler_count_staged(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                              const int32_t* quota, const uint32_t* bin_start, int64_t gv, int64_t p_wave_begin, float std_limit,
                              int32_t attempt0, int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added,
                              uint32_t* dcount, uint32_t* have_before, uint32_t* remaining, const G2pcSampleStage* stage,
                              void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(stage && have_before && bin_start, G2PC_ERR_ARG, "bad arguments");
    if (p_wave_begin < 0 || p_wave_begin > gv) p_wave_begin = gv;
    G2PC_REQUIRE((p_wave_begin == 0 || stage->thread_rows) && (p_wave_begin == gv || (stage->wave_rows && stage->wave_row_start)),
                 G2PC_ERR_ARG, "missing staging arrays");
    return sampler_count_impl(means, cov9, perm, pbin, quota, gv, p_wave_begin, std_limit, attempt0, num_attempts, seed, gid_base,
                              added, dcount, remaining, stage, have_before, bin_start, stream);
}
}

extern "C" {
size_t g2pc_sampler_scan_workspace(int64_t gv, int32_t attempts) { return g2pc::scan_rows_workspace((long)gv, (int)attempts) + g2pc::scan_workspace((long)gv); }

/* dscan[a] = exclusive scan of dcount[a] for a = 0 .. attempts-1 (rows of gv resp. gv+1 entries): two launches for all
 * attempts when gv <= 2M, one scan per attempt beyond that */
int g2pc_sampler_scan_counts(const uint32_t* dcount, uint32_t* dscan, int64_t gv, int32_t attempts, void* ws, size_t ws_bytes,
                             void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(gv >= 0 && attempts >= 0, G2PC_ERR_ARG, "negative size");
    if (gv == 0 || attempts == 0) return G2PC_OK;
    G2PC_REQUIRE(dcount && dscan && ws, G2PC_ERR_ARG, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    int rc = scan_exclusive_rows_u32(dcount, dscan, (long)gv, (int)attempts, ws, ws_bytes, s);
    if (rc != G2PC_ERR_UNSUPPORTED) return rc;
    for (int a = 0; a < attempts; ++a) {
        rc = scan_exclusive_u32(dcount + (size_t)a * gv, dscan + (size_t)a * (gv + 1), (long)gv, ws, ws_bytes, s);
        if (rc) return rc;
    }
    return G2PC_OK;
}

/* Section table on the device (see k_sections): sec_base i64[num_bins * (1 + attempts) + 1]; info_host (optional,
 * PINNED host memory, i64[2]) receives {total rows M, *remaining} when the kernel runs -- the one value the host needs
 * before it can hand the cloud out, read after a single stream synchronisation at the very end of the job. */
int g2pc_sampler_sections(const uint32_t* bin_start, const int32_t* quota, int32_t num_bins, int32_t attempts,
                          const uint32_t* dscan, int64_t gv, int emit_means, int64_t* sec_base, int64_t* info_host,
                          const uint32_t* remaining, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(bin_start && quota && sec_base && num_bins >= 0 && attempts >= 0, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(attempts == 0 || dscan, G2PC_ERR_ARG, "missing scans");
    hipLaunchKernelGGL(k_sections, dim3(1), dim3(SEC_T), 0, (hipStream_t)stream, bin_start, quota, (int)num_bins, (int)attempts,
                       dscan, (long)gv, emit_means, sec_base, info_host, remaining);
    return check_launch("g2pc_sampler_sections");
}

/* Row-balanced emission of the whole cloud (means and every attempt's rows) in one launch: `rows_capacity` >= M is the
 * size the output arrays were allocated for (the launch covers it; blocks beyond the real M, read from sec_base, exit). */
static int sampler_emit_impl(const float* means, const float* cov9, const float* colours, const float* normals,
                             const uint32_t* perm, const uint32_t* bin_start, int32_t num_bins, int32_t attempt0,
                             int32_t attempts, int64_t gv, uint64_t seed, uint64_t gid_base, const uint32_t* dscan,
                             const int64_t* sec_base, int64_t rows_capacity, float* out_points, float* out_colours,
                             float* out_normals, int32_t* out_gauss, const G2pcSampleStage* st, int64_t p_wave,
                             const uint32_t* hb, const int32_t* quota, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(means && cov9 && colours && perm && bin_start && sec_base && out_points && out_colours, G2PC_ERR_ARG,
                 "bad arguments");
    G2PC_REQUIRE(!out_normals || normals, G2PC_ERR_ARG, "normals requested but not given");
    G2PC_REQUIRE(attempts == 0 || dscan, G2PC_ERR_ARG, "missing scans");
    if (rows_capacity <= 0 || num_bins <= 0) return G2PC_OK;
    hipLaunchKernelGGL(k_emit_rows, dim3(cdiv(rows_capacity, ER_ROWS)), dim3(ER_T), 0, (hipStream_t)stream, means, cov9, colours,
                       normals, perm, bin_start, (int)num_bins, (int)attempts, (long)gv, (int)attempt0, (unsigned)seed,
                       (unsigned)(seed >> 32), gid_base, dscan, sec_base, out_points, out_colours, out_normals, out_gauss,
                       (long)rows_capacity, st ? (const float*)st->thread_rows : nullptr, (long)p_wave,
                       st ? (const float*)st->wave_rows : nullptr, st ? (const unsigned long long*)st->wave_row_start : nullptr,
                       st ? hb : nullptr, quota);
    return check_launch("g2pc_sampler_emit_rows");
}
int g2pc_sampler_emit_rows(const float* means, const float* cov9, const float* colours, const float* normals,
                           const uint32_t* perm, const uint32_t* bin_start, int32_t num_bins, int32_t attempt0,
                           int32_t attempts, int64_t gv, uint64_t seed, uint64_t gid_base, const uint32_t* dscan,
                           const int64_t* sec_base, int64_t rows_capacity, float* out_points, float* out_colours,
                           float* out_normals, int32_t* out_gauss, void* stream) {
    return sampler_emit_impl(means, cov9, colours, normals, perm, bin_start, num_bins, attempt0, attempts, gv, seed, gid_base, dscan,
                             sec_base, rows_capacity, out_points, out_colours, out_normals, out_gauss, nullptr, 0, nullptr, nullptr,
                             stream);
}
int g2pc_sampler_emit_rows_staged(const float* means, const float* cov9, const float* colours, const float* normals,
                                  const uint32_t* perm, const uint32_t* bin_start, const int32_t* quota, int32_t num_bins,
                                  int32_t attempts, int64_t gv, int64_t p_wave_begin, const uint32_t* dscan,
                                  const uint32_t* have_before, const int64_t* sec_base, int64_t rows_capacity,
                                  const G2pcSampleStage* stage, float* out_points, float* out_colours, float* out_normals,
                                  int32_t* out_gauss, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(stage && quota && (attempts == 0 || have_before), G2PC_ERR_ARG, "bad arguments");
    if (p_wave_begin < 0 || p_wave_begin > gv) p_wave_begin = gv;
    return sampler_emit_impl(means, cov9, colours, normals, perm, bin_start, num_bins, 0, attempts, gv, 0, 0, dscan, sec_base,
                             rows_capacity, out_points, out_colours, out_normals, out_gauss, stage, p_wave_begin, have_before, quota,
                             stream);
}

int g2pc_mahalanobis(const float* means, const float* samples, const float* cov9, int64_t n, float* out,
                     void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(means && samples && cov9 && out, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_mahalanobis, dim3(cdiv(n, SM_T)), dim3(SM_T), 0, (hipStream_t)stream, means, samples, cov9,
                       (long)n, out);
    return check_launch("g2pc_mahalanobis");
}

int g2pc_sample_mvn(const float* means, const float* cov9, int64_t g, int32_t n, uint64_t seed, uint64_t gid_base,
                    int32_t attempt, float* out, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(g >= 0 && n >= 0, G2PC_ERR_ARG, "negative size");
    if (g == 0 || n == 0) return G2PC_OK;
    G2PC_REQUIRE(means && cov9 && out, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_sample_mvn, dim3(cdiv(g, SM_T)), dim3(SM_T), 0, (hipStream_t)stream, means, cov9, (long)g,
                       (int)n, (unsigned)seed, (unsigned)(seed >> 32), gid_base, (unsigned)attempt, out);
    return check_launch("g2pc_sample_mvn");
}
}

extern "C" {
/* The sampler's whole tail in ONE call (ABI 7): partition by bin -> staged count pass -> scans -> section table -> copying emission,
 * everything it needs between them in ONE caller-provided workspace.  The launches are those of g2pc_sampler_partition,
 * _stage_plan, _count_staged, _scan_counts, _sections and _emit_rows_staged in that order (bit-identical results); what the call
 * removes is the host work between them -- a dozen interpreter-level calls and allocations per job, which pace a job whose kernels
 * take 0.6 ms.  attempts <= 8 (the binned default runs 5; exact_num_points' 100 keep the chunked loop of the caller). */
struct SamplerRunLayout { size_t perm, pbin, part_ws, part_bytes, added, dcount, hb, dscan, scan_ws, scan_bytes, sec_base, wrs, stage_t, stage_w, total; };
static SamplerRunLayout sampler_run_layout(int64_t G, int64_t gv, int64_t p_wave, int32_t B, int32_t A, int64_t lane_planes, int64_t wave_rows) {
    using namespace g2pc;
    SamplerRunLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = align_up(off); off = o + bytes; return o; };
    const size_t gvs = (size_t)(gv > 0 ? gv : 1);
    l.perm = take((size_t)G * 4); l.pbin = take((size_t)G * 4);
    l.part_bytes = g2pc_sampler_plan_workspace(G); l.part_ws = take(l.part_bytes);
    l.added = take((gvs + 1) * 4);
    l.dcount = take((size_t)A * gvs * 4); l.hb = take((size_t)A * gvs * 4); l.dscan = take((size_t)A * (gvs + 1) * 4);
    l.scan_bytes = g2pc_sampler_scan_workspace(gv, A); l.scan_ws = take(l.scan_bytes);
    l.sec_base = take(((size_t)(B > 0 ? B : 1) * (1 + A) + 1) * 8);
    l.wrs = take(((size_t)B + 1) * 8);
    l.stage_t = take((size_t)(lane_planes > 0 ? lane_planes : 0) * (size_t)(p_wave > 0 ? p_wave : 0) * 12 + 16);
    l.stage_w = take((size_t)(wave_rows > 0 ? wave_rows : 0) * 12 + 16);
    l.total = align_up(off) + 256;
    return l;
}
size_t g2pc_sampler_run_workspace(int64_t g, int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempts,
                                  int64_t lane_planes, int64_t wave_rows) {
    return sampler_run_layout(g, gv, p_wave_begin, num_bins, attempts, lane_planes, wave_rows).total;
}
/* byte offset of the section table (i64[num_bins * (1 + attempts) + 1]) inside the workspace -- diagnostics read it */
size_t g2pc_sampler_run_sections_offset(int64_t g, int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempts,
                                        int64_t lane_planes, int64_t wave_rows) {
    return sampler_run_layout(g, gv, p_wave_begin, num_bins, attempts, lane_planes, wave_rows).sec_base;
}
int g2pc_sampler_run(const float* means, const float* cov9, const float* colours, const float* normals, const int32_t* ppg,
                     int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, const int32_t* quota, const uint32_t* bin_start,
                     int32_t num_bins, int64_t gv, int64_t p_wave_begin, int32_t wave_min_draws, int64_t lane_planes,
                     int64_t wave_rows, float std_limit, int32_t attempts, uint64_t seed, uint64_t gid_base, int emit_means,
                     int64_t rows_capacity, float* out_points, float* out_colours, float* out_normals, int32_t* out_gauss,
                     int64_t* info_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(means && cov9 && colours && ppg && bin_of_ppg && quota && bin_start && ws && g > 0 &&
                     (rows_capacity <= 0 || (out_points && out_colours)), G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(attempts >= 0 && attempts <= 8 && num_bins >= 0 && gv >= 0, G2PC_ERR_ARG, "bad sizes (attempts <= 8)");
    if (p_wave_begin < 0 || p_wave_begin > gv) p_wave_begin = gv;
    const SamplerRunLayout l = sampler_run_layout(g, gv, p_wave_begin, num_bins, attempts, lane_planes, wave_rows);
    G2PC_REQUIRE(l.total <= ws_bytes, G2PC_ERR_WORKSPACE, "workspace too small");
    char* b = (char*)ws;
    uint32_t* perm = (uint32_t*)(b + l.perm); uint32_t* pbin = (uint32_t*)(b + l.pbin);
    uint32_t* added = (uint32_t*)(b + l.added); uint32_t* dcount = (uint32_t*)(b + l.dcount); uint32_t* hb = (uint32_t*)(b + l.hb);
    uint32_t* dscan = (uint32_t*)(b + l.dscan); int64_t* sec_base = (int64_t*)(b + l.sec_base);
    uint64_t* wrs = (uint64_t*)(b + l.wrs);
    const size_t gvs = (size_t)(gv > 0 ? gv : 1);
    uint32_t* remaining = added + gvs;
    hipStream_t s = (hipStream_t)stream;
    int rc = g2pc_sampler_partition(ppg, g, bin_of_ppg, lut_len, num_bins, perm, pbin, b + l.part_ws, l.part_bytes, stream);
    if (rc) return rc;
    hipMemsetAsync(added, 0, (gvs + 1) * 4, s);
    G2pcSampleStage st{(float*)(b + l.stage_t), (float*)(b + l.stage_w), wrs};
    const bool sampling = attempts > 0 && gv > 0;
    if (sampling) {
        if (p_wave_begin < gv) {
            rc = g2pc_sampler_stage_plan(bin_start, quota, num_bins, wave_min_draws, wrs, stream);
            if (rc) return rc;
        }
        rc = g2pc_sampler_count_staged(means, cov9, perm, pbin, quota, bin_start, gv, p_wave_begin, std_limit, 0, attempts, seed,
                                       gid_base, added, dcount, hb, remaining, &st, stream);
        if (rc) return rc;
        rc = g2pc_sampler_scan_counts(dcount, dscan, gv, attempts, b + l.scan_ws, l.scan_bytes, stream);
        if (rc) return rc;
    }
    const int A = sampling ? attempts : 0;
    rc = g2pc_sampler_sections(bin_start, quota, num_bins, A, A ? dscan : nullptr, gv, emit_means, sec_base, info_host, remaining, stream);
    if (rc) return rc;
    if (rows_capacity > 0 && gv > 0 && num_bins > 0) {
        rc = A ? g2pc_sampler_emit_rows_staged(means, cov9, colours, normals, perm, bin_start, quota, num_bins, A, gv, p_wave_begin, dscan,
                                               hb, sec_base, rows_capacity, &st, out_points, out_colours, out_normals, out_gauss, stream)
               : g2pc_sampler_emit_rows(means, cov9, colours, normals, perm, bin_start, num_bins, 0, 0, gv, seed, gid_base, nullptr,
                                        sec_base, rows_capacity, out_points, out_colours, out_normals, out_gauss, stream);
        if (rc) return rc;
    }
    return check_launch("g2pc_sampler_run");
}
}
