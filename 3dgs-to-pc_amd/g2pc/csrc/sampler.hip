// Point sampler: multivariate-normal draws with Mahalanobis rejection and first-k emission.
//
// Reference: gauss_to_pc.py:140-155 (MultivariateNormal: Cholesky + loc + L eps), :92-103 (Mahalanobis via
// the explicit inverse), :157-275 (attempt loop; emits the FIRST min(n - added, accepted) draws of every
// Gaussian, not the accepted ones), :277-371 (bin loop and output order).
//
// The reference walks bins on the host and re-concatenates the whole cloud per bin (torch.cat).  Here every
// bin is processed at once: Gaussians are stably partitioned by bin (radix sort on the bin id), a counting
// pass produces d[attempt][position], per-attempt scans + a tiny host-side section table give every
// (bin, attempt, Gaussian) its output offset, and an emission pass regenerates the keyed Philox draws and
// writes them straight to their final place.  Small quotas run one Gaussian per lane (quota is uniform
// inside a bin, so lanes of a wave run in lock-step); large quotas run one Gaussian per wave with the 64
// lanes striding over the draws (ballot/popcount for the accept count, coalesced emission).
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
                                                      uint32_t* __restrict__ remaining) {
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
        if (have < (unsigned)n) {
            unsigned acc = 0;
            for (int k = 0; k < n; ++k) {
                float x, y, z;
                acc += draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), (unsigned)k, std_limit, x, y, z) ? 1u : 0u;
            }
            unsigned room = (unsigned)n - have;
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
                                                    uint32_t* __restrict__ remaining) {
    const unsigned lane = threadIdx.x & 63;
    long p = p_begin + (long)blockIdx.x * (SM_T / kWave) + (threadIdx.x >> 6);
    if (p >= gv) return;                                   // whole wave leaves together
    const unsigned g = perm[p];
    const int n = quota[pbin[p]] - 1;
    unsigned have = added[p];
    if (n <= 0 || have >= (unsigned)n) {
        if (lane == 0) for (int a = 0; a < num_attempts; ++a) dcount[(size_t)a * gv + p] = 0;
        return;
    }
    GaussSample s;
    load_gauss(means, cov9, g, s);
    const uint64_t gid = gid_base + g;
    const unsigned gid_lo = (unsigned)gid, gid_hi = (unsigned)(gid >> 32);
    for (int a = 0; a < num_attempts; ++a) {
        unsigned d = 0;
        if (have < (unsigned)n) {
            unsigned acc = 0;
            for (int k0 = 0; k0 < n; k0 += 64) {
                int k = k0 + (int)lane;
                float x, y, z;
                bool ok = false;
                if (k < n) ok = draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), (unsigned)k, std_limit, x, y, z);
                acc += (unsigned)__popcll(__ballot(ok));
            }
            unsigned room = (unsigned)n - have;
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

// ---- emission pass --------------------------------------------------------------------------------------
__device__ __forceinline__ void put3(float* __restrict__ dst, size_t idx, float a, float b, float c) {
    dst[3 * idx + 0] = a; dst[3 * idx + 1] = b; dst[3 * idx + 2] = c;
}

__global__ __launch_bounds__(SM_T) void k_emit_means(const float* __restrict__ means,
                                                    const float* __restrict__ colours,
                                                    const float* __restrict__ normals,
                                                    const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ pbin,
                                                    const uint32_t* __restrict__ bin_start,
                                                    const int32_t* __restrict__ quota, long gv, int sec_stride,
                                                    const int64_t* __restrict__ sec_base,
                                                    float* __restrict__ out_points, float* __restrict__ out_colours,
                                                    float* __restrict__ out_normals, int32_t* __restrict__ out_gauss) {
    long p = (long)blockIdx.x * SM_T + threadIdx.x;
    if (p >= gv) return;
    const unsigned b = pbin[p];
    if (quota[b] <= 0) return;
    const unsigned g = perm[p];
    const size_t o = (size_t)sec_base[(size_t)b * sec_stride] + (size_t)(p - bin_start[b]);
    put3(out_points, o, means[3 * (size_t)g], means[3 * (size_t)g + 1], means[3 * (size_t)g + 2]);
    put3(out_colours, o, colours[3 * (size_t)g], colours[3 * (size_t)g + 1], colours[3 * (size_t)g + 2]);
    if (out_normals) put3(out_normals, o, normals[3 * (size_t)g], normals[3 * (size_t)g + 1], normals[3 * (size_t)g + 2]);
    if (out_gauss) out_gauss[o] = (int32_t)g;
}

// Lane-per-Gaussian emission (quotas below a wave's worth of draws).  The draws of a lane go to a run of consecutive output
// rows and the runs of consecutive lanes follow one another (prefix sums of d), so a wave's output for one attempt is
// one contiguous range -- but written lane by lane it would be 64 scattered 12-byte rows per store.  The wave therefore
// generates its points into a wave-private LDS window in output order and then writes the window row-per-lane:
// fully coalesced 12-byte stores for the points, and for the colours / normals / ids of each row's owner (found by a
// 6-step search of the lanes' run offsets).
constexpr int EM_WIN = 512;                       // output rows staged per wave and pass
__global__ __launch_bounds__(SM_T) void k_emit_thread(
    const float* __restrict__ means, const float* __restrict__ cov9, const float* __restrict__ colours,
    const float* __restrict__ normals, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pbin,
    const uint32_t* __restrict__ bin_start, long p_end, long gv, int attempt0, int num_attempts, int sec_stride,
    unsigned seed_lo, unsigned seed_hi, uint64_t gid_base, const uint32_t* __restrict__ dcount,
    const uint32_t* __restrict__ dscan, const int64_t* __restrict__ sec_base, float* __restrict__ out_points,
    float* __restrict__ out_colours, float* __restrict__ out_normals, int32_t* __restrict__ out_gauss) {
    constexpr int NW = SM_T / kWave;
    __shared__ float s_xyz[NW][EM_WIN][3];
    __shared__ uint32_t s_run[NW][kWave + 1];      // exclusive prefix of d over the lanes (+ total): where each run starts
    __shared__ unsigned long long s_o0[NW][kWave]; // first output row of each lane's run
    __shared__ float s_cn[NW][kWave][6];           // colour, normal of each lane's Gaussian
    __shared__ int32_t s_id[NW][kWave];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * SM_T + threadIdx.x;
    const bool valid = p < p_end;
    unsigned any_draws = 0;
    if (valid)
        for (int a = 0; a < num_attempts; ++a) any_draws += dcount[(size_t)a * gv + p];
    if (!__any(any_draws != 0)) return;            // whole wave has nothing to emit
    unsigned g = 0, b = 0, bs = 0;
    GaussSample s = {};
    unsigned gid_lo = 0, gid_hi = 0;
    if (any_draws) {
        g = perm[p];
        b = pbin[p];
        bs = bin_start[b];
        load_gauss(means, cov9, g, s);
        const uint64_t gid = gid_base + g;
        gid_lo = (unsigned)gid; gid_hi = (unsigned)(gid >> 32);
        s_cn[w][lane][0] = colours[3 * (size_t)g]; s_cn[w][lane][1] = colours[3 * (size_t)g + 1]; s_cn[w][lane][2] = colours[3 * (size_t)g + 2];
        if (out_normals) { s_cn[w][lane][3] = normals[3 * (size_t)g]; s_cn[w][lane][4] = normals[3 * (size_t)g + 1]; s_cn[w][lane][5] = normals[3 * (size_t)g + 2]; }
        s_id[w][lane] = (int32_t)g;
    }
    for (int a = 0; a < num_attempts; ++a) {
        const unsigned d = any_draws ? dcount[(size_t)a * gv + p] : 0u;
        const unsigned incl = wave_incl_scan_u32(d);
        const unsigned total = __shfl(incl, 63);
        if (total == 0) continue;                  // uniform
        const unsigned run = incl - d;
        wave_sync();                               // the previous attempt's readers are done
        s_run[w][lane] = run;
        if (lane == 63) s_run[w][kWave] = total;
        if (d) {
            const uint32_t* sc = dscan + (size_t)a * (gv + 1);
            s_o0[w][lane] = (unsigned long long)sec_base[(size_t)b * sec_stride + 1 + attempt0 + a] + (unsigned long long)(sc[p] - sc[bs]);
        }
        for (unsigned w0 = 0; w0 < total; w0 += EM_WIN) {
            // this lane's draws whose rows fall into [w0, w0 + EM_WIN)
            const unsigned klo = w0 > run ? w0 - run : 0u;
            const unsigned khi = (run + d > w0 + EM_WIN) ? (w0 + EM_WIN - run) : d;      // run <= w0 + EM_WIN whenever klo < d matters
            if (run < w0 + EM_WIN)
                for (unsigned k = klo; k < khi; ++k) {
                    float x, y, z;
                    draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), k, 0.f, x, y, z);
                    float* dst = s_xyz[w][run + k - w0];
                    dst[0] = x; dst[1] = y; dst[2] = z;
                }
            wave_sync();
            const unsigned wend = total < w0 + EM_WIN ? total : w0 + EM_WIN;
            for (unsigned t = w0 + lane; t < wend; t += 64) {
                // owner = the last lane whose run starts at or before row t (runs of d = 0 share their successor's start)
                unsigned lo = 0, hi = kWave;                                  // invariant: s_run[lo] <= t < s_run[hi]
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (s_run[w][mid] <= t) lo = mid; else hi = mid;
                }
                const size_t o = (size_t)s_o0[w][lo] + (size_t)(t - s_run[w][lo]);
                const float* src = s_xyz[w][t - w0];
                put3(out_points, o, src[0], src[1], src[2]);
                put3(out_colours, o, s_cn[w][lo][0], s_cn[w][lo][1], s_cn[w][lo][2]);
                if (out_normals) put3(out_normals, o, s_cn[w][lo][3], s_cn[w][lo][4], s_cn[w][lo][5]);
                if (out_gauss) out_gauss[o] = s_id[w][lo];
            }
            wave_sync();                           // window drained before it is refilled
        }
    }
}

__global__ __launch_bounds__(SM_T) void k_emit_wave(
    const float* __restrict__ means, const float* __restrict__ cov9, const float* __restrict__ colours,
    const float* __restrict__ normals, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pbin,
    const uint32_t* __restrict__ bin_start, long p_begin, long gv, int attempt0, int num_attempts, int sec_stride,
    unsigned seed_lo, unsigned seed_hi, uint64_t gid_base, const uint32_t* __restrict__ dcount,
    const uint32_t* __restrict__ dscan, const int64_t* __restrict__ sec_base, float* __restrict__ out_points,
    float* __restrict__ out_colours, float* __restrict__ out_normals, int32_t* __restrict__ out_gauss) {
    const unsigned lane = threadIdx.x & 63;
    long p = p_begin + (long)blockIdx.x * (SM_T / kWave) + (threadIdx.x >> 6);
    if (p >= gv) return;
    const unsigned g = perm[p];
    const unsigned b = pbin[p];
    const unsigned bs = bin_start[b];
    GaussSample s;
    load_gauss(means, cov9, g, s);
    const float cr = colours[3 * (size_t)g], cg = colours[3 * (size_t)g + 1], cb = colours[3 * (size_t)g + 2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (out_normals) { nx = normals[3 * (size_t)g]; ny = normals[3 * (size_t)g + 1]; nz = normals[3 * (size_t)g + 2]; }
    const uint64_t gid = gid_base + g;
    const unsigned gid_lo = (unsigned)gid, gid_hi = (unsigned)(gid >> 32);
    for (int a = 0; a < num_attempts; ++a) {
        const unsigned d = dcount[(size_t)a * gv + p];
        if (d == 0) continue;
        const uint32_t* sc = dscan + (size_t)a * (gv + 1);
        const size_t o0 = (size_t)sec_base[(size_t)b * sec_stride + 1 + attempt0 + a] + (size_t)(sc[p] - sc[bs]);
        for (unsigned k = lane; k < d; k += 64) {
            float x, y, z;
            draw(s, seed_lo, seed_hi, gid_lo, gid_hi, (unsigned)(attempt0 + a), k, 0.f, x, y, z);
            const size_t o = o0 + k;
            put3(out_points, o, x, y, z);
            put3(out_colours, o, cr, cg, cb);
            if (out_normals) put3(out_normals, o, nx, ny, nz);
            if (out_gauss) out_gauss[o] = (int32_t)g;
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
        put3(out, (size_t)k * g_count + g, x, y, z);
    }
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

int g2pc_sampler_count(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                       const int32_t* quota, int64_t gv, int64_t p_wave_begin, float std_limit, int32_t attempt0,
                       int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added, uint32_t* dcount,
                       uint32_t* remaining, void* stream) {
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
                           gid_base, added, dcount, remaining);
    if (p_wave_begin < gv)
        hipLaunchKernelGGL(k_count_wave, dim3(cdiv(gv - p_wave_begin, SM_T / kWave)), dim3(SM_T), 0, s, means, cov9,
                           perm, pbin, quota, (long)p_wave_begin, (long)gv, std_limit, (int)attempt0,
                           (int)num_attempts, slo, shi, gid_base, added, dcount, remaining);
    return check_launch("g2pc_sampler_count");
}

int g2pc_sampler_emit(const float* means, const float* cov9, const float* colours, const float* normals,
                      const uint32_t* perm, const uint32_t* pbin, const uint32_t* bin_start, const int32_t* quota,
                      int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempt0, int32_t num_attempts,
                      int32_t sec_stride, uint64_t seed, uint64_t gid_base, const uint32_t* dcount,
                      const uint32_t* dscan, const int64_t* sec_base, int emit_means, float* out_points,
                      float* out_colours, float* out_normals, int32_t* out_gauss, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(gv >= 0 && means && cov9 && colours && perm && pbin && bin_start && quota && sec_base &&
                     out_points && out_colours,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(!out_normals || normals, G2PC_ERR_ARG, "normals requested but not given");
    (void)num_bins;
    if (gv == 0) return G2PC_OK;
    if (p_wave_begin < 0 || p_wave_begin > gv) p_wave_begin = gv;
    hipStream_t s = (hipStream_t)stream;
    unsigned slo = (unsigned)seed, shi = (unsigned)(seed >> 32);
    if (emit_means)
        hipLaunchKernelGGL(k_emit_means, dim3(cdiv(gv, SM_T)), dim3(SM_T), 0, s, means, colours, normals, perm, pbin,
                           bin_start, quota, (long)gv, (int)sec_stride, sec_base, out_points, out_colours,
                           out_normals, out_gauss);
    if (num_attempts > 0) {
        G2PC_REQUIRE(dcount && dscan, G2PC_ERR_ARG, "missing counts");
        if (p_wave_begin > 0)
            hipLaunchKernelGGL(k_emit_thread, dim3(cdiv(p_wave_begin, SM_T)), dim3(SM_T), 0, s, means, cov9, colours,
                               normals, perm, pbin, bin_start, (long)p_wave_begin, (long)gv, (int)attempt0,
                               (int)num_attempts, (int)sec_stride, slo, shi, gid_base, dcount, dscan, sec_base,
                               out_points, out_colours, out_normals, out_gauss);
        if (p_wave_begin < gv)
            hipLaunchKernelGGL(k_emit_wave, dim3(cdiv(gv - p_wave_begin, SM_T / kWave)), dim3(SM_T), 0, s, means, cov9,
                               colours, normals, perm, pbin, bin_start, (long)p_wave_begin, (long)gv, (int)attempt0,
                               (int)num_attempts, (int)sec_stride, slo, shi, gid_base, dcount, dscan, sec_base,
                               out_points, out_colours, out_normals, out_gauss);
    }
    return check_launch("g2pc_sampler_emit");
}
}

extern "C" {
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
