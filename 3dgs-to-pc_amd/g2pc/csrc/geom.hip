// Per-Gaussian geometry kernels (one thread per Gaussian; HBM-bound streaming kernels).
// Reference behaviour: gauss_handler.py (file:line cited per kernel).
#include "g2pc_internal.h"

namespace g2pc {

constexpr int GEO_T = 256;

// gauss_handler.py:26-47 (build_rotation, quaternion r,x,y,z, no normalisation),
// :49-58 (L = R diag(exp(mod*s))), :60-63 (Sigma = L L^T), :12-24 (strip_symmetric),
// :89-106 (normal = R[:, argmin(log-scale)]).
__global__ __launch_bounds__(GEO_T) void k_build_cov(const float* __restrict__ ls, const float4* __restrict__ rot,
                                                    float mod, long n, float* __restrict__ cov9,
                                                    float* __restrict__ cov6, float* __restrict__ normals,
                                                    float* __restrict__ rotmat) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= n) return;
    float s0 = ls[3 * i + 0], s1 = ls[3 * i + 1], s2 = ls[3 * i + 2];
    float4 q = rot[i];
    float r = q.x, x = q.y, y = q.z, z = q.w;
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - r * z);
    R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y);
    R[2][1] = 2.0f * (y * z + r * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    if (rotmat) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) rotmat[9 * i + 3 * a + b] = R[a][b];
    }
    // exp in double from IEEE operations only, rounded once (exp_cr, g2pc_device.inl): the correctly rounded f32 exponential,
    // bit for bit the same on the device and in the CPU build.  torch.exp on the CPU (MKL VML, high-accuracy mode) returns it
    // for 98.9 % of its arguments (measured, tools/torch_order_probe.py); a 1-ulp f32 expf would halve that.
    float e[3] = {exp_cr(mod * s0), exp_cr(mod * s1), exp_cr(mod * s2)};
    float L[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) L[a][b] = R[a][b] * e[b];
    float C[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) C[a][b] = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
    float* o = cov9 + 9 * i;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) o[3 * a + b] = C[a][b];
    if (cov6) {
        float* p = cov6 + 6 * i;
        p[0] = C[0][0]; p[1] = C[0][1]; p[2] = C[0][2]; p[3] = C[1][1]; p[4] = C[1][2]; p[5] = C[2][2];
    }
    if (normals) {
        int ax = 0;
        float m = s0;
        if (s1 < m) { m = s1; ax = 1; }
        if (s2 < m) { m = s2; ax = 2; }
        normals[3 * i + 0] = R[0][ax];
        normals[3 * i + 1] = R[1][ax];
        normals[3 * i + 2] = R[2][ax];
    }
}

// unit eigenvector of symmetric A for eigenvalue lam: largest cross product of rows of (A - lam I)
__device__ __forceinline__ void sym3_eigvec(const double A[6], double lam, double v[3]) {
    double r0[3] = {A[0] - lam, A[1], A[2]};
    double r1[3] = {A[1], A[3] - lam, A[4]};
    double r2[3] = {A[2], A[4], A[5] - lam};
    double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    double c1[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    double c2[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
    double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
    double n2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
    const double* c = c0;
    double nn = n0;
    if (n1 > nn) { c = c1; nn = n1; }
    if (n2 > nn) { c = c2; nn = n2; }
    if (nn > 0.0) {
        double inv = 1.0 / sqrt(nn);
        v[0] = c[0] * inv; v[1] = c[1] * inv; v[2] = c[2] * inv;
    } else {
        v[0] = 1.0; v[1] = 0.0; v[2] = 0.0;
    }
}

// Eigen-decomposition of a symmetric 3x3 by deflation: the eigenvector of the best separated eigenvalue of the
// closed-form estimate e[] first, then the exact 2x2 problem in its orthogonal complement.  The trigonometric closed
// form loses half its digits on the two eigenvalues that are close RELATIVE TO THE SPREAD (acos near +-1): with a
// spectrum (1e6, 1e-3, 1e-7) it is off by ~1e-2 on the small pair, while the isolated eigenvalue stays accurate to
// fp64 rounding; the deflated pair then carries an absolute error of ~1e-16 * lambda_max.  lam[0] = the isolated
// eigenvalue (v0), lam[1] >= lam[2] the pair (v1, v2).  Robust to repeated eigenvalues.
__device__ __forceinline__ void sym3_decompose(const double A[6], const double e[3], double lam[3], double v0[3],
                                               double v1[3], double v2[3]) {
    double gap_lo = e[1] - e[0], gap_hi = e[2] - e[1];
    int first = gap_hi >= gap_lo ? 2 : 0;
    sym3_eigvec(A, e[first], v0);
    // orthonormal basis (u, w) of the complement
    double u[3];
    if (fabs(v0[0]) > fabs(v0[1])) {
        double inv = 1.0 / sqrt(v0[0] * v0[0] + v0[2] * v0[2]);
        u[0] = -v0[2] * inv; u[1] = 0.0; u[2] = v0[0] * inv;
    } else {
        double inv = 1.0 / sqrt(v0[1] * v0[1] + v0[2] * v0[2]);
        u[0] = 0.0; u[1] = v0[2] * inv; u[2] = -v0[1] * inv;
    }
    double w[3] = {v0[1] * u[2] - v0[2] * u[1], v0[2] * u[0] - v0[0] * u[2], v0[0] * u[1] - v0[1] * u[0]};
    // 2x2 projected matrix
    double Au[3] = {A[0] * u[0] + A[1] * u[1] + A[2] * u[2], A[1] * u[0] + A[3] * u[1] + A[4] * u[2],
                    A[2] * u[0] + A[4] * u[1] + A[5] * u[2]};
    double Aw[3] = {A[0] * w[0] + A[1] * w[1] + A[2] * w[2], A[1] * w[0] + A[3] * w[1] + A[4] * w[2],
                    A[2] * w[0] + A[4] * w[1] + A[5] * w[2]};
    double m00 = u[0] * Au[0] + u[1] * Au[1] + u[2] * Au[2];
    double m01 = u[0] * Aw[0] + u[1] * Aw[1] + u[2] * Aw[2];
    double m11 = w[0] * Aw[0] + w[1] * Aw[1] + w[2] * Aw[2];
    double tr = 0.5 * (m00 + m11), df = 0.5 * (m00 - m11);
    double rad = sqrt(df * df + m01 * m01);
    double l1 = tr + rad, l2 = tr - rad;
    double cs = 1.0, sn = 0.0;   // eigenvector of l1 in (u,w) coordinates
    if (rad > 0.0) {
        double a = df + rad, b = m01;
        if (fabs(a) < fabs(df - rad)) { a = m01; b = -(df - rad); }   // other column, better conditioned
        double nn = sqrt(a * a + b * b);
        if (nn > 0.0) { cs = a / nn; sn = b / nn; }
    }
    v1[0] = cs * u[0] + sn * w[0]; v1[1] = cs * u[1] + sn * w[1]; v1[2] = cs * u[2] + sn * w[2];
    v2[0] = -sn * u[0] + cs * w[0]; v2[1] = -sn * u[1] + cs * w[1]; v2[2] = -sn * u[2] + cs * w[2];
    // the isolated eigenvalue as the Rayleigh quotient of its eigenvector (second-order accurate in the vector's error)
    double Av[3] = {A[0] * v0[0] + A[1] * v0[1] + A[2] * v0[2], A[1] * v0[0] + A[3] * v0[1] + A[4] * v0[2],
                    A[2] * v0[0] + A[4] * v0[1] + A[5] * v0[2]};
    lam[0] = v0[0] * Av[0] + v0[1] * Av[1] + v0[2] * Av[2];
    lam[1] = l1;
    lam[2] = l2;
}

// smallest eigenvalue of symmetric A: the closed form where it is trustworthy (its error on a close pair is
// ~1.5e-8 * spread: decided with a 100x margin), the deflated decomposition otherwise.  Well-conditioned matrices --
// every Gaussian of a sane scene -- take the first branch.
__device__ __forceinline__ double sym3_min_eig(const double A[6], const double e[3], double thr) {
    const double spread = fmax(fabs(e[0]), fabs(e[2]));
    if (fabs(e[0] - thr) > 1.5e-6 * spread || !(spread == spread)) return e[0];
    double lam[3], v0[3], v1[3], v2[3];
    sym3_decompose(A, e, lam, v0, v1, v2);
    return fmin(lam[0], lam[2]);
}

// clamp_covariances (gauss_handler.py:114-127): A <- V max(w, eps) V^T.
__device__ __forceinline__ void sym3_clamp(double A[6], const double e[3], double eps) {
    double lam[3], v0[3], v1[3], v2[3];
    sym3_decompose(A, e, lam, v0, v1, v2);
    double w0 = lam[0] < eps ? eps : lam[0];
    double w1 = lam[1] < eps ? eps : lam[1];
    double w2 = lam[2] < eps ? eps : lam[2];
    A[0] = w0 * v0[0] * v0[0] + w1 * v1[0] * v1[0] + w2 * v2[0] * v2[0];
    A[1] = w0 * v0[0] * v0[1] + w1 * v1[0] * v1[1] + w2 * v2[0] * v2[1];
    A[2] = w0 * v0[0] * v0[2] + w1 * v1[0] * v1[2] + w2 * v2[0] * v2[2];
    A[3] = w0 * v0[1] * v0[1] + w1 * v1[1] * v1[1] + w2 * v2[1] * v2[1];
    A[4] = w0 * v0[1] * v0[2] + w1 * v1[1] * v1[2] + w2 * v2[1] * v2[2];
    A[5] = w0 * v0[2] * v0[2] + w1 * v1[2] * v1[2] + w2 * v2[2] * v2[2];
}

// validate_covariances (gauss_handler.py:142-166) incl. regularise (:129-140), the eigen test (:108-112)
// and clamp (:114-127).  The reference tests eigvals(cov).real <= eps with a general (non-symmetric)
// LAPACK solver on the float32 matrix; here: closed-form fp64 eigenvalues of the symmetrised matrix.
// sqrt of the Knud-Thomsen ellipsoid area (p = 1.6075) from the eigenvalues, in fp32 as the reference evaluates it
__device__ __forceinline__ float sqrt_ellipsoid_area(const double e[3]) {
    const float p = 1.6075f;
    float a = sqrtf((float)e[0]), b = sqrtf((float)e[1]), cc = sqrtf((float)e[2]);
    float radicand = (powf(a * b, p) + powf(a * cc, p) + powf(b * cc, p)) / 3.0f;
    float area = 12.566370614359172f * powf(radicand, (float)(1.0 / 1.6075));
    return sqrtf(area);
}

__global__ __launch_bounds__(GEO_T) void k_validate_cov(float* __restrict__ cov9, long n, int regularise,
                                                       float reg_eps, float eps, float min_eps, int iters,
                                                       uint8_t* __restrict__ keep, uint32_t* __restrict__ culled_count,
                                                       float* __restrict__ sqrt_area) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= n) return;
    float* c = cov9 + 9 * i;
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = c[k];
    bool dirty = false;
    if (regularise) { m[0] += reg_eps; m[4] += reg_eps; m[8] += reg_eps; dirty = true; }
    double e[3];
    for (int it = 0; it < iters; ++it) {
        double A[6] = {(double)m[0], 0.5 * ((double)m[1] + (double)m[3]), 0.5 * ((double)m[2] + (double)m[6]),
                       (double)m[4], 0.5 * ((double)m[5] + (double)m[7]), (double)m[8]};
        sym3_eigvals(A[0], A[1], A[2], A[3], A[4], A[5], e);
        if (!(sym3_min_eig(A, e, (double)eps) <= (double)eps)) break;
        sym3_clamp(A, e, (double)eps);
        m[0] = (float)A[0]; m[1] = (float)A[1]; m[2] = (float)A[2];
        m[3] = (float)A[1]; m[4] = (float)A[3]; m[5] = (float)A[4];
        m[6] = (float)A[2]; m[7] = (float)A[4]; m[8] = (float)A[5];
        dirty = true;
    }
    const double Af[6] = {(double)m[0], 0.5 * ((double)m[1] + (double)m[3]), 0.5 * ((double)m[2] + (double)m[6]),
                          (double)m[4], 0.5 * ((double)m[5] + (double)m[7]), (double)m[8]};
    sym3_eigvals(Af[0], Af[1], Af[2], Af[3], Af[4], Af[5], e);
    const bool cull = sym3_min_eig(Af, e, (double)min_eps) <= (double)min_eps;    // NaN compares false -> kept, as in the reference
    keep[i] = cull ? 0 : 1;
    if (cull && culled_count) atomicAdd(culled_count, 1u);                        // rare: no contention in practice
    // the eigenvalues of the matrix as it is stored below are exactly what get_gaussian_magnitudes derives from it: keep
    // sqrt(ellipsoid area) so that the magnitudes are one multiply per Gaussian instead of a second eigen-decomposition
    if (sqrt_area) sqrt_area[i] = sqrt_ellipsoid_area(e);
    if (dirty) {
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = m[k];
    }
}

// get_gaussian_magnitudes (gauss_handler.py:252-279): fp32 formula on the (fp64-accurate, rounded to
// fp32) eigenvalues; the result is symmetric in (a,b,c) so the eigenvalue order does not matter.
__global__ __launch_bounds__(GEO_T) void k_magnitudes(const float* __restrict__ cov9,
                                                     const float* __restrict__ weights, long n,
                                                     double* __restrict__ sizes) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= n) return;
    const float* c = cov9 + 9 * i;
    double e[3];
    sym3_eigvals((double)c[0], 0.5 * ((double)c[1] + (double)c[3]), 0.5 * ((double)c[2] + (double)c[6]),
                 (double)c[4], 0.5 * ((double)c[5] + (double)c[7]), (double)c[8], e);
    float mag = sqrt_ellipsoid_area(e) * weights[i];
    sizes[i] = (double)mag;
}
// the same from the sqrt(area) the validation pass kept (k_validate_cov): one multiply per Gaussian
__global__ __launch_bounds__(GEO_T) void k_magnitudes_area(const float* __restrict__ sqrt_area, const float* __restrict__ weights,
                                                          long n, double* __restrict__ sizes) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i < n) sizes[i] = (double)(sqrt_area[i] * weights[i]);
}


// apply_min_opacity / apply_bounding_box (gauss_handler.py:195-224): mask &= tests (strict inequalities)
__global__ __launch_bounds__(GEO_T) void k_cull_mask(const float* __restrict__ xyz, const float* __restrict__ opac,
                                                    long n, int use_opacity, float min_opacity, int use_min,
                                                    float3 bmin, int use_max, float3 bmax,
                                                    uint8_t* __restrict__ mask) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= n) return;
    bool ok = mask[i] != 0;
    if (use_opacity) ok = ok && (opac[i] > min_opacity);
    if (use_min | use_max) {
        float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (use_min) ok = ok && (x > bmin.x) && (y > bmin.y) && (z > bmin.z);
        if (use_max) ok = ok && (x < bmax.x) && (y < bmax.y) && (z < bmax.z);
    }
    mask[i] = ok ? 1 : 0;
}

__global__ __launch_bounds__(GEO_T) void k_mask_to_u32(const uint8_t* __restrict__ mask, long n,
                                                      uint32_t* __restrict__ flag) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i < n) flag[i] = mask[i] ? 1u : 0u;
}

__global__ __launch_bounds__(GEO_T) void k_compact_index(const uint8_t* __restrict__ mask,
                                                        const uint32_t* __restrict__ rank, long n,
                                                        uint32_t* __restrict__ index) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i < n && mask[i]) index[rank[i]] = (uint32_t)i;
}

// boolean-index gather of rows (filter_gaussians, gauss_handler.py:171-193): dst[j,:] = src[index[j],:]
__global__ __launch_bounds__(GEO_T) void k_gather_rows(const uint32_t* __restrict__ src,
                                                      const uint32_t* __restrict__ index, long m, int row_words,
                                                      uint32_t* __restrict__ dst) {
    long t = (long)blockIdx.x * GEO_T + threadIdx.x;
    long total = m * row_words;
    if (t >= total) return;
    long j = t / row_words;
    int w = (int)(t - j * row_words);
    dst[t] = src[(size_t)index[j] * row_words + w];
}

// the same for up to eight arrays that share the index (filter_gaussians compacts xyz, scales, rotations, colours,
// opacities, covariances, normals with ONE index): one launch instead of one per array -- the job's tail is launch-bound
struct GatherSet { const uint32_t* src[8]; uint32_t* dst[8]; int words[8]; int first[9]; int count; };
__global__ __launch_bounds__(GEO_T) void k_gather_rows_multi(GatherSet g, const uint32_t* __restrict__ index, long m) {
    const int total_words = g.first[g.count];
    long t = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (t >= m * total_words) return;
    const long j = t / total_words;
    const int w = (int)(t - j * total_words);
    int a = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < g.count && w >= g.first[k]) a = k;
    const int ww = w - g.first[a];
    g.dst[a][(size_t)j * g.words[a] + ww] = g.src[a][(size_t)index[j] * g.words[a] + ww];
}

// save_xyz_to_ply (gauss_dataloader.py:172-202): pack one binary-little-endian PLY vertex record per point,
// x y z [nx ny nz] r g b = 3 (or 6) float32 + 3 uchar (colours truncated like numpy's astype(np.uint8)).
// Records are 15 / 27 bytes (unaligned): a block stages 256 records in LDS and streams them out as whole dwords.
__global__ __launch_bounds__(GEO_T) void k_pack_ply(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                   const float* __restrict__ col, long m, int rec,
                                                   uint32_t* __restrict__ out) {
    __shared__ uint32_t stage[GEO_T * 27 / 4 + 1];
    uint8_t* sb = (uint8_t*)stage;
    const long base = (long)blockIdx.x * GEO_T;
    const long i = base + threadIdx.x;
    if (i < m) {
        uint8_t* r = sb + (size_t)threadIdx.x * rec;
        float v[6];
        int nf = 3;
        v[0] = pts[3 * i]; v[1] = pts[3 * i + 1]; v[2] = pts[3 * i + 2];
        if (nrm) { v[3] = nrm[3 * i]; v[4] = nrm[3 * i + 1]; v[5] = nrm[3 * i + 2]; nf = 6; }
        for (int k = 0; k < nf; ++k) {
            uint32_t bits = __float_as_uint(v[k]);
            r[4 * k + 0] = (uint8_t)bits; r[4 * k + 1] = (uint8_t)(bits >> 8);
            r[4 * k + 2] = (uint8_t)(bits >> 16); r[4 * k + 3] = (uint8_t)(bits >> 24);
        }
        for (int k = 0; k < 3; ++k) r[4 * nf + k] = (uint8_t)(int)col[3 * i + k];
    }
    __syncthreads();
    const long cnt = (m - base) < GEO_T ? (m - base) : GEO_T;
    const long bytes = cnt * rec;
    const long words = (bytes + 3) / 4;                 // the output buffer is padded to a multiple of 4 bytes
    uint32_t* dst = out + (size_t)blockIdx.x * (GEO_T * rec / 4);
    for (long w = threadIdx.x; w < words; w += GEO_T) dst[w] = stage[w];
}

}  // namespace g2pc

extern "C" {
int g2pc_build_covariances(const float* log_scales, const float* rots, float scaling_modifier, int64_t n,
                           float* cov9, float* cov6, float* normals, float* rotmat, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && (n == 0 || (log_scales && rots && cov9)), G2PC_ERR_ARG, "null input");
    if (n == 0) return G2PC_OK;
    hipLaunchKernelGGL(k_build_cov, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, log_scales,
                       (const float4*)rots, scaling_modifier, (long)n, cov9, cov6, normals, rotmat);
    return check_launch("g2pc_build_covariances");
}

int g2pc_validate_covariances_area(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                                   int iters, uint8_t* keep, uint32_t* culled_count, float* sqrt_area, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && (n == 0 || (cov9 && keep)), G2PC_ERR_ARG, "null input");
    if (n == 0) return G2PC_OK;
    hipLaunchKernelGGL(k_validate_cov, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, cov9, (long)n,
                       regularise, reg_eps, eps, min_eps, iters, keep, culled_count, sqrt_area);
    return check_launch("g2pc_validate_covariances");
}
int g2pc_validate_covariances_counted(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                                      int iters, uint8_t* keep, uint32_t* culled_count, void* stream) {
    return g2pc_validate_covariances_area(cov9, n, regularise, reg_eps, eps, min_eps, iters, keep, culled_count, nullptr, stream);
}
int g2pc_gaussian_magnitudes_from_area(const float* sqrt_area, const float* weights, int64_t n, double* sizes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && (n == 0 || (sqrt_area && weights && sizes)), G2PC_ERR_ARG, "null input");
    if (n == 0) return G2PC_OK;
    hipLaunchKernelGGL(k_magnitudes_area, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, sqrt_area, weights, (long)n, sizes);
    return check_launch("g2pc_gaussian_magnitudes_from_area");
}

int g2pc_validate_covariances(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                              int iters, uint8_t* keep, void* stream) {
    return g2pc_validate_covariances_counted(cov9, n, regularise, reg_eps, eps, min_eps, iters, keep, nullptr, stream);
}

int g2pc_gaussian_magnitudes(const float* cov9, const float* weights, int64_t n, double* sizes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && (n == 0 || (cov9 && weights && sizes)), G2PC_ERR_ARG, "null input");
    if (n == 0) return G2PC_OK;
    hipLaunchKernelGGL(k_magnitudes, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, cov9, weights,
                       (long)n, sizes);
    return check_launch("g2pc_gaussian_magnitudes");
}
}

extern "C" {
int g2pc_cull_mask(const float* xyz, const float* opacities, int64_t n, int use_min_opacity, float min_opacity,
                   const float* bbox_min, const float* bbox_max, uint8_t* mask, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && mask, G2PC_ERR_ARG, "null mask");
    G2PC_REQUIRE(!use_min_opacity || opacities, G2PC_ERR_ARG, "opacities missing");
    G2PC_REQUIRE((!bbox_min && !bbox_max) || xyz, G2PC_ERR_ARG, "xyz missing");
    if (n == 0) return G2PC_OK;
    float3 a = bbox_min ? make_float3(bbox_min[0], bbox_min[1], bbox_min[2]) : make_float3(0, 0, 0);
    float3 b = bbox_max ? make_float3(bbox_max[0], bbox_max[1], bbox_max[2]) : make_float3(0, 0, 0);
    hipLaunchKernelGGL(k_cull_mask, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, xyz, opacities,
                       (long)n, use_min_opacity, min_opacity, bbox_min ? 1 : 0, a, bbox_max ? 1 : 0, b, mask);
    return check_launch("g2pc_cull_mask");
}

size_t g2pc_compact_workspace(int64_t n) {
    return g2pc::align_up((size_t)(n + 1) * 4) * 2 + g2pc::scan_workspace(n) + 1024;
}

int g2pc_compact_index(const uint8_t* mask, int64_t n, uint32_t* index, uint32_t* count, void* ws, size_t ws_bytes,
                       void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0 && mask && index && count && ws, G2PC_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { hipMemsetAsync(count, 0, 4, s); return G2PC_OK; }
    Arena ar(ws, ws_bytes);
    uint32_t* flag = ar.get<uint32_t>((size_t)n + 1);
    uint32_t* rank = ar.get<uint32_t>((size_t)n + 1);
    size_t sb = scan_workspace(n);
    char* sw = ar.get<char>(sb);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_mask_to_u32, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, s, mask, (long)n, flag);
    int rc = scan_exclusive_u32(flag, rank, n, sw, sb, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_index, dim3(cdiv(n, GEO_T)), dim3(GEO_T), 0, s, mask, rank, (long)n, index);
    hipMemcpyAsync(count, rank + n, 4, hipMemcpyDeviceToDevice, s);
    return check_launch("g2pc_compact_index");
}

int g2pc_gather_rows(const void* src, const uint32_t* index, int64_t m, int32_t row_bytes, void* dst, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(m >= 0 && row_bytes > 0 && row_bytes % 4 == 0, G2PC_ERR_ARG, "row_bytes must be a multiple of 4");
    if (m == 0) return G2PC_OK;
    G2PC_REQUIRE(src && index && dst, G2PC_ERR_ARG, "null pointer");
    long total = (long)m * (row_bytes / 4);
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(total, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream,
                       (const uint32_t*)src, index, (long)m, (int)(row_bytes / 4), (uint32_t*)dst);
    return check_launch("g2pc_gather_rows");
}
}

extern "C" {
int g2pc_gather_rows_multi(const void* const* srcs, void* const* dsts, const int32_t* row_bytes, int32_t count,
                           const uint32_t* index, int64_t m, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(count >= 1 && count <= 8 && m >= 0 && srcs && dsts && row_bytes, G2PC_ERR_ARG, "1 .. 8 arrays");
    if (m == 0) return G2PC_OK;
    G2PC_REQUIRE(index, G2PC_ERR_ARG, "null index");
    GatherSet g{};
    g.count = count;
    g.first[0] = 0;
    for (int k = 0; k < count; ++k) {
        G2PC_REQUIRE(srcs[k] && dsts[k] && row_bytes[k] > 0 && row_bytes[k] % 4 == 0, G2PC_ERR_ARG, "rows must be whole 4-byte words");
        g.src[k] = (const uint32_t*)srcs[k]; g.dst[k] = (uint32_t*)dsts[k]; g.words[k] = row_bytes[k] / 4;
        g.first[k + 1] = g.first[k] + g.words[k];
    }
    const long total = (long)m * g.first[count];
    hipLaunchKernelGGL(k_gather_rows_multi, dim3(cdiv(total, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, g, index, (long)m);
    return check_launch("g2pc_gather_rows_multi");
}
}

namespace g2pc {
__global__ __launch_bounds__(GEO_T) void k_scatter_ones_u8(const uint32_t* __restrict__ index, long m, uint8_t* __restrict__ dst,
                                                          long n) {
    long i = (long)blockIdx.x * GEO_T + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = index[i];
    if ((long)j < n) dst[j] = 1;
}
}  // namespace g2pc

extern "C" {
/* dst[index[i]] = 1 for i < m (uint8 mask of n entries; out-of-range positions are ignored) */
int g2pc_scatter_ones_u8(const uint32_t* index, int64_t m, uint8_t* dst, int64_t n, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(m >= 0 && n >= 0, G2PC_ERR_ARG, "negative size");
    if (m == 0) return G2PC_OK;
    G2PC_REQUIRE(index && dst, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_scatter_ones_u8, dim3(cdiv(m, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, index, (long)m, dst, (long)n);
    return check_launch("g2pc_scatter_ones_u8");
}
}

extern "C" {
int g2pc_pack_ply_vertices(const float* points, const float* normals, const float* colours, int64_t m, void* out,
                           void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(m >= 0, G2PC_ERR_ARG, "negative count");
    if (m == 0) return G2PC_OK;
    G2PC_REQUIRE(points && colours && out, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_pack_ply, dim3(cdiv(m, GEO_T)), dim3(GEO_T), 0, (hipStream_t)stream, points, normals, colours,
                       (long)m, normals ? 27 : 15, (uint32_t*)out);
    return check_launch("g2pc_pack_ply_vertices");
}
}
