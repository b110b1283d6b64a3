/*
 * g2pc -- C ABI of the MI355X-native 3DGS -> point-cloud hot path (libg2pc.so, gfx950).
 *
 * Drop-in boundary for Lewis-Stuart-11/3DGS-to-PC.  Every entry point takes raw DEVICE pointers
 * (HBM resident, caller owned), sizes, a caller-owned workspace and a HIP stream handle
 * (`void* stream` == hipStream_t; NULL = default stream).  All calls are stream-ordered and return
 * immediately; none allocates, none synchronises.  Return value: 0 (G2PC_OK) or a negative
 * G2PC_ERR_* code; g2pc_last_error() gives a thread-local message.
 *
 * Each declaration cites the reference code (file:line under the reference repository) whose
 * behaviour it replaces.  The reference-side bindings (ctypes stubs a maintainer would add to
 * gauss_handler.py / gauss_to_pc.py / gauss_render.py) are shown in INTEGRATION.md; the
 * in-tree mirror of those modules lives in 3dgs-to-pc_amd/.
 */
#ifndef G2PC_H
#define G2PC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2PC_ABI_VERSION 1

#define G2PC_OK 0
#define G2PC_ERR_ARG (-1)
#define G2PC_ERR_WORKSPACE (-2)
#define G2PC_ERR_LAUNCH (-3)
#define G2PC_ERR_UNSUPPORTED (-4)

const char* g2pc_last_error(void);
int g2pc_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Device primitives (replace cub::DeviceScan::InclusiveSum rasterizer_impl.cu:285,
 * cub::DeviceRadixSort::SortPairs rasterizer_impl.cu:311-316, and the torch.unique / boolean
 * index compactions of gauss_to_pc.py:225-238,334).
 * ------------------------------------------------------------------------------------------- */
size_t g2pc_scan_workspace(int64_t n);
/* out[0..n] = exclusive prefix sums of in[0..n-1]; out[n] = total.  in may alias out. */
int g2pc_scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, void* ws, size_t ws_bytes, void* stream);
size_t g2pc_sort_workspace(int64_t n);
/* stable LSD radix sort of (key,value) pairs on key bits [bit_lo, bit_hi) */
int g2pc_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                        uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                        size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gaussian model (gauss_handler.py)
 * ------------------------------------------------------------------------------------------- */
/* build_covariance_from_scaling_rotation (gauss_handler.py:26-63) fused with calculate_normals
 * (gauss_handler.py:89-106).  log_scales f32[n,3], rots f32[n,4] (r,x,y,z; NOT normalised, as in the
 * reference).  cov9 f32[n,3,3] out; cov6 f32[n,6] (xx,xy,xz,yy,yz,zz; strip_symmetric
 * gauss_handler.py:12-24), normals f32[n,3] and rotmat f32[n,3,3] (build_rotation) are optional (NULL to skip). */
int g2pc_build_covariances(const float* log_scales, const float* rots, float scaling_modifier, int64_t n,
                           float* cov9, float* cov6, float* normals, float* rotmat, void* stream);

/* validate_covariances (gauss_handler.py:108-166): in-place regularise (+reg_eps*I when regularise),
 * `iters` rounds of {min eig <= eps -> clamp eigenvalues to eps and recompose}, then
 * keep[i] = (min eig > min_eps).  cov9 f32[n,3,3] in/out, keep u8[n] out. */
int g2pc_validate_covariances(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                              int iters, uint8_t* keep, void* stream);

/* get_gaussian_magnitudes (gauss_handler.py:252-279): Knud-Thomsen ellipsoid area from the eigenvalues,
 * sqrt, times weights (contributions or opacities) -> f64[n]. */
int g2pc_gaussian_magnitudes(const float* cov9, const float* weights, int64_t n, double* sizes, void* stream);

/* apply_min_opacity + apply_bounding_box (gauss_handler.py:195-224): mask[i] &= opacity > min (when
 * use_min_opacity) & strict box tests (bbox_min / bbox_max are HOST float[3] or NULL).  mask u8[n] in/out. */
int g2pc_cull_mask(const float* xyz, const float* opacities, int64_t n, int use_min_opacity, float min_opacity,
                   const float* bbox_min, const float* bbox_max, uint8_t* mask, void* stream);

/* filter_gaussians (gauss_handler.py:171-193) = stream compaction: index u32[<=n] of the set mask entries in
 * ascending order, count u32[1] (device); then gather whole rows (row_bytes multiple of 4) per tensor. */
size_t g2pc_compact_workspace(int64_t n);
int g2pc_compact_index(const uint8_t* mask, int64_t n, uint32_t* index, uint32_t* count, void* ws, size_t ws_bytes,
                       void* stream);
int g2pc_gather_rows(const void* src, const uint32_t* index, int64_t m, int32_t row_bytes, void* dst, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Point allocation (gauss_to_pc.py:73-138)
 * ------------------------------------------------------------------------------------------- */
size_t g2pc_distribute_points_workspace(int64_t n);
/* distribute_points (gauss_to_pc.py:73-90) including the negative-slice quirk.
 * sizes f64[n] -> ppg_f64[n] (optional) and ppg_i32[n]; stats i64[4] (device) =
 * {sum of ppg before zero fill, number of zeros, k as used by the reference slice, max ppg after fill}. */
int g2pc_distribute_points(const double* sizes, int64_t n, int64_t num_points, double* ppg_f64, int32_t* ppg_i32,
                           int64_t* stats, void* ws, size_t ws_bytes, void* stream);
/* torch.bincount(points_per_gaussian) (gauss_to_pc.py:110): hist u32[hist_len] must be zeroed by the caller;
 * values >= hist_len are ignored. */
int g2pc_bincount_i32(const int32_t* values, int64_t n, uint32_t* hist, int64_t hist_len, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Point sampler (gauss_to_pc.py:140-371).  See DESIGN.md "Sampler" for the three stages.
 * Noise is keyed Philox4x32-10: eps(seed, gid_base + g, attempt, k) (oracle/np_philox.py).
 * ------------------------------------------------------------------------------------------- */
size_t g2pc_sampler_plan_workspace(int64_t g);
/* Stage 1: stable partition of the Gaussians by bin.  bin_of_ppg i32[lut_len] maps a ppg value to its bin
 * (or -1: not a member of any bin / quota <= 0).  Outputs: perm u32[g] (positions sorted by (bin, index),
 * Gaussians without a bin last), pbin u32[g] (bin id per sorted position, num_bins for "none"),
 * bin_start u32[num_bins+2] (exclusive offsets per bin; [num_bins] = number of Gaussians that have a bin). */
int g2pc_sampler_plan(const int32_t* ppg, int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, int32_t num_bins,
                      uint32_t* perm, uint32_t* pbin, uint32_t* bin_start, void* ws, size_t ws_bytes, void* stream);

/* Stage 2: attempts [attempt0, attempt0 + num_attempts) of create_new_gaussian_points
 * (gauss_to_pc.py:157-275) for every bin at once, counting only.  means f32[*,3], cov9 f32[*,3,3]
 * indexed by Gaussian; quota i32[num_bins] = points per Gaussian of the bin (the mean counts as the
 * first one, gauss_to_pc.py:352-361); added u32[gv] in/out (zero before attempt 0);
 * dcount u32[num_attempts, gv] out = points emitted per (attempt, sorted position);
 * remaining u32[1] out (+= number of Gaussians still short after the last attempt; caller zeroes).
 * Sorted positions < p_wave_begin run one Gaussian per lane, the rest one Gaussian per wave64 (pick the
 * first position whose quota-1 >= 32; quotas ascend with the bin index). */
int g2pc_sampler_count(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                       const int32_t* quota, int64_t gv, int64_t p_wave_begin, float std_limit, int32_t attempt0,
                       int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added, uint32_t* dcount,
                       uint32_t* remaining, void* stream);

/* Stage 3: emission in the reference's order (bins ascending; per bin: all means, then attempt 0, 1, ...;
 * inside a section Gaussians in index order, each with its FIRST d draws -- gauss_to_pc.py:247-258).
 * dscan u32[num_attempts, gv+1] = per-attempt exclusive scans of dcount; sec_base i64[num_bins, 1+num_attempts_total]
 * = output offset of every section (column 0 = means); colours f32[*,3], normals f32[*,3] or NULL by Gaussian.
 * Writes out_points/out_colours/out_normals f32[m,3] and (optional) out_gauss i32[m].
 * emit_means != 0 also writes the means sections (do it on the first attempt chunk only). */
int g2pc_sampler_emit(const float* means, const float* cov9, const float* colours, const float* normals,
                      const uint32_t* perm, const uint32_t* pbin, const uint32_t* bin_start, const int32_t* quota,
                      int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempt0, int32_t num_attempts,
                      int32_t sec_stride, uint64_t seed, uint64_t gid_base, const uint32_t* dcount,
                      const uint32_t* dscan, const int64_t* sec_base, int emit_means, float* out_points,
                      float* out_colours, float* out_normals, int32_t* out_gauss, void* stream);

/* mahalanobis() (gauss_to_pc.py:92-103): out[i] = sqrt(d^T inv(cov_i) d), d = means_i - samples_i; NaN when the
 * quadratic form is negative (the caller's `<=` then rejects, as in the reference). */
int g2pc_mahalanobis(const float* means, const float* samples, const float* cov9, int64_t n, float* out,
                     void* stream);
/* sample_from_multivariate_normal (gauss_to_pc.py:140-155): out f32[n, g, 3] (sample-major like
 * MultivariateNormal.sample((n,))) = mean + chol(cov) eps(seed, gid_base + g, attempt, k). */
int g2pc_sample_mvn(const float* means, const float* cov9, int64_t g, int32_t n, uint64_t seed, uint64_t gid_base,
                    int32_t attempt, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* G2PC_H */
