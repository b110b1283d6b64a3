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
/* the library is built with -fvisibility=hidden: the entry points below are ALL it exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define G2PC_ABI_VERSION 7
#define G2PC_TILE_PARENTS 4       /* entries per tile of G2pcTileLayout.tile_parent */

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
/* diagnostic: out[4w..4w+3] = {dpp max, dpp min, shuffle max, shuffle min} of the 64 values of wave w */
int g2pc_selftest_wave_reduce(const uint32_t* in, uint32_t* out, int64_t waves, void* stream);
size_t g2pc_scan_workspace(int64_t n);
/* out[0..n] = exclusive prefix sums of in[0..n-1]; out[n] = total.  in may alias out. */
int g2pc_scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, void* ws, size_t ws_bytes, void* stream);
size_t g2pc_sort_workspace(int64_t n);
/* --- hipGraph capture of a sequence of g2pc_* calls --------------------------------------------------------------------
 * Every g2pc_* entry point only queues work on `stream` (no allocation, no synchronisation), so whatever is called
 * between g2pc_graph_capture_begin(stream) and g2pc_graph_capture_end(stream, &graph) -- on that stream, with fixed
 * sizes -- is recorded instead of executed and can then be replayed with ONE launch per replay.  `stream` must be a
 * non-default stream. */
int g2pc_graph_capture_begin(void* stream);
int g2pc_graph_capture_end(void* stream, void** graph_exec);
int g2pc_graph_launch(void* graph_exec, void* stream);
int g2pc_graph_destroy(void* graph_exec);

/* stable LSD radix sort of (key,value) pairs on key bits [bit_lo, bit_hi); vals_in = vals_out = vals_tmp = NULL: keys only */
int g2pc_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                        uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                        size_t ws_bytes, void* stream);
/* Stable ascending sort of (key, value) pairs whose keys are bit patterns of positive floats spread over their range (a
 * camera's depths): range-normalised bucket pass (per-chunk histogram table, no global atomics) + one wave per bucket
 * sorting (key, position) composites in LDS -- six launches, every key moved once.  Keys 0xFFFFFFFF go last, in
 * unspecified order among themselves (everything else is THE stable ascending order).  vals == NULL: the values are the
 * input positions.
 * *overflow (device u32) != 0 afterwards: a bucket held more keys than its room (1024; 4096 when the mean bucket exceeds
 * 256 keys) and the result is NOT sorted -- repeat with g2pc_sort_pairs_u32.  Pays up to ~2 M keys (MI355X: 90 vs 101 us
 * at 1 M, 533 vs 278 us at 5 M). */
size_t g2pc_bucket_sort_workspace(int64_t n);
int g2pc_bucket_sort_u32(const uint32_t* keys, const uint32_t* vals, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
                         uint32_t* overflow, void* ws, size_t ws_bytes, void* stream);

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
/* the same; culled_count (optional, device u32, zeroed by the caller) += number of rows culled, so that the caller's
 * `if any culled: filter` needs one 4-byte read-back instead of a reduction over the mask */
int g2pc_validate_covariances_counted(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                                      int iters, uint8_t* keep, uint32_t* culled_count, void* stream);

/* ABI 7: validate_covariances and get_gaussian_magnitudes share ONE eigen-decomposition: sqrt_area (optional, f32[n] out) =
 * sqrt of the Knud-Thomsen ellipsoid area of the matrix as the pass leaves it (gauss_handler.py:259-277) -- bit for bit the
 * factor g2pc_gaussian_magnitudes derives from the stored matrix --, and g2pc_gaussian_magnitudes_from_area multiplies it
 * with the weights (contributions or opacities, :278-279). */
int g2pc_validate_covariances_area(float* cov9, int64_t n, int regularise, float reg_eps, float eps, float min_eps,
                                   int iters, uint8_t* keep, uint32_t* culled_count, float* sqrt_area, void* stream);
int g2pc_gaussian_magnitudes_from_area(const float* sqrt_area, const float* weights, int64_t n, double* sizes, void* stream);

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
/* The same for `count` (1 .. 8) arrays that share the index, in ONE launch: dsts[k][j, :] = srcs[k][index[j], :], rows of
 * row_bytes[k] bytes (multiples of 4).  srcs / dsts / row_bytes are HOST arrays of device pointers / sizes.
 * filter_gaussians (gauss_handler.py:171-193) compacts every per-Gaussian array with one index. */
int g2pc_gather_rows_multi(const void* const* srcs, void* const* dsts, const int32_t* row_bytes, int32_t count,
                           const uint32_t* index, int64_t m, void* stream);
/* dst[index[i]] = 1 for i < m (uint8 mask of n entries): the keep mask of cull_large_gaussians (gauss_handler.py:235-250) */
int g2pc_scatter_ones_u8(const uint32_t* index, int64_t m, uint8_t* dst, int64_t n, void* stream);

/* --- clean_point_cloud (mesh_handler.py:89-94), "next" row f4 -----------------------------------------------------
 * The reference hands the cloud to Open3D (not vendored, version unpinned): PointCloud::RemoveStatisticalOutliers(
 * nb_neighbors = 20, std_ratio): for every point the mean Euclidean distance to its nb_neighbors nearest points (the
 * point itself included at distance 0; float64; distances summed in ascending order); kept when
 * 0 < mean < mean(means) + std_ratio * std(means) (sample std, over the means > 0).
 * Here the means come from an exact kNN on a uniform grid: _grid_build bins the points into cubic cells of edge `cell`
 * (grid origin[3], dims[3], dims[0]*dims[1]*dims[2] < 2^31; points outside the grid are clamped into its border cells,
 * which stays exact -- the shell bounds only rely on the cell index growing with the coordinate -- so the grid may cover
 * a robust quantile box instead of the bounding box when a few far-away points would otherwise dictate the cell size)
 * -> sorted_pos f32[m,4] (x, y, z, original index bits) in cell order, cell_start u32[cells+2] (exclusive offsets; [cells] = [cells+1] = m), and
 * (optional) the number of non-empty cells in *occupied (device) so the caller can refine the resolution;
 * _knn_mean_distance searches growing shells of cells until the k-th distance is provably final and writes the
 * mean distance of point i (original order) to avg[i] (f64).  k <= 32.  slack: absolute safety margin subtracted from
 * the shell bound (covers the float rounding of the cell assignment; 1e-5 * cell + 1e-6 * max|coordinate| is ample).
 * Queries: all m points (query_idx NULL) or the num_queries original indices in query_idx (then `points`, the cloud in
 * original order, supplies their coordinates).  A query still open after max_rings shells is not answered but appended
 * to unresolved[*unresolved_count] (device u32[queries], device u32 zeroed by the call): the caller rebuilds the grid
 * with a larger cell and asks again for those -- a cascade that ends when the shells cover the whole grid. */
size_t g2pc_outlier_grid_workspace(int64_t m);
int g2pc_outlier_grid_build(const float* points, int64_t m, const float* origin, float cell, const int32_t* dims,
                            float* sorted_pos, uint32_t* cell_start, uint32_t* occupied, void* ws, size_t ws_bytes,
                            void* stream);
int g2pc_outlier_knn_mean_distance(const float* sorted_pos, const uint32_t* cell_start, int64_t m, const float* origin,
                                   float cell, const int32_t* dims, int32_t k, double slack, const float* points,
                                   const uint32_t* query_idx, int64_t num_queries, int32_t max_rings, uint32_t* unresolved,
                                   uint32_t* unresolved_count, double* avg, void* stream);

/* save_xyz_to_ply (gauss_dataloader.py:118-202), "next" row f1: pack m binary-little-endian PLY vertex records
 * (x y z [nx ny nz] red green blue; 27 bytes with normals, 15 without; colours f32 -> uchar by truncation) into `out`
 * (device, 4-byte aligned, at least ceil(m*rec/4)*4 bytes). */
int g2pc_pack_ply_vertices(const float* points, const float* normals, const float* colours, int64_t m, void* out,
                           void* stream);

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

/* Stage 3 (emission in the reference's order) is g2pc_sampler_scan_counts / _sections / _emit_rows below. */

/* The bin table of generate_pointcloud / calculate_bin_sizes (gauss_to_pc.py:105-138, :308-337) built ON THE DEVICE from the
 * histogram of points per Gaussian (g2pc_bincount_i32), numpy's arithmetic value for value (float64 second differences,
 * group sums, cut = max // 50, start_bin counted from the peak as the reference does, tail rounded up to multiples of
 * bin_size, quota = floor(s + (e - s) / 2)).  Replaces the mid-pipeline round trip histogram -> host numpy -> look-up table.
 * hist u32[hist_len <= 8192]; stats = g2pc_distribute_points' device result ([3] = max points per Gaussian; NULL: hist_len-1).
 * Out (device): lut i32[hist_len] (bin of a point count, -1 none), quota i32[hist_len], bin_start u32[hist_len + 2]
 * (exclusive scan of the bins' member counts), bin_lo i32[hist_len] (first point count of each bin).  plan_host: PINNED
 * host memory, i64[12] (ABI 7; [10] = the largest quota - 1 of a lane-mode bin, [11] reserved), written through its device mapping: {bins, Gaussians in bins, first wave-mode position, any
 * sampling, mean rows, upper bound of output rows, error, start_bin, bin_size, distinct point counts}; error 1 = some
 * Gaussian has >= hist_len points (use a longer histogram on the host path), 2 = fewer than two distinct point counts in
 * binned mode (the reference's np.gradient raises there). */
size_t g2pc_sampler_bin_table_workspace(int64_t hist_len);
int g2pc_sampler_bin_table(const uint32_t* hist, int64_t hist_len, const int64_t* stats, int32_t exact, int32_t emit_means,
                           int32_t wave_min_draws, int32_t* lut, int32_t* quota, uint32_t* bin_start, int32_t* bin_lo,
                           int64_t* plan_host, void* ws, size_t ws_bytes, void* stream);
/* --- sampler, device-resident bookkeeping (no host round trip between the count and the emission) ---------------
 * g2pc_sampler_scan_counts: dscan[a] (u32[gv+1] per attempt) = exclusive scan of dcount[a] (u32[gv] per attempt).
 * g2pc_sampler_sections:    sec_base i64[num_bins * (1 + attempts) + 1] = first output row of every (bin, means |
 *                           attempt a) section in the reference's order (gauss_to_pc.py:341-371), last entry = M;
 *                           info_host (optional, pinned, i64[2]) <- {M, *remaining}.
 * g2pc_sampler_emit_rows:   the whole cloud in one row-per-lane launch (create_new_gaussian_points' emission,
 *                           gauss_to_pc.py:247-258, for all bins and attempts); outputs sized rows_capacity >= M. */
int g2pc_sampler_partition(const int32_t* ppg, int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, int32_t num_bins,
                           uint32_t* perm, uint32_t* pbin, void* ws, size_t ws_bytes, void* stream);   /* g2pc_sampler_plan without bin_start */
size_t g2pc_sampler_scan_workspace(int64_t gv, int32_t attempts);
int g2pc_sampler_scan_counts(const uint32_t* dcount, uint32_t* dscan, int64_t gv, int32_t attempts, void* ws, size_t ws_bytes,
                             void* stream);
int g2pc_sampler_sections(const uint32_t* bin_start, const int32_t* quota, int32_t num_bins, int32_t attempts,
                          const uint32_t* dscan, int64_t gv, int emit_means, int64_t* sec_base, int64_t* info_host,
                          const uint32_t* remaining, void* stream);
int g2pc_sampler_emit_rows(const float* means, const float* cov9, const float* colours, const float* normals,
                           const uint32_t* perm, const uint32_t* bin_start, int32_t num_bins, int32_t attempt0,
                           int32_t attempts, int64_t gv, uint64_t seed, uint64_t gid_base, const uint32_t* dscan,
                           const int64_t* sec_base, int64_t rows_capacity, float* out_points, float* out_colours,
                           float* out_normals, int32_t* out_gauss, void* stream);

/* --- ABI 7: draw-once sampling ------------------------------------------------------------------------------------------------
 * create_new_gaussian_points draws every sample once (gauss_to_pc.py:204-212) and emits the first d of them (:247-258).  The
 * two-pass form above evaluates a draw twice (count, then emission: Philox + Box-Muller + Cholesky, ~250 instructions each time);
 * here the count pass KEEPS every point it may have to emit -- draw k of an attempt with k < room, room = quota - 1 - rows so
 * far -- and the emission copies.  Lane-mode positions (p < p_wave_begin: one Gaussian per lane) stage plane-major,
 * thread_rows f32[planes][p_wave_begin][3] with planes >= the largest quota - 1 among them (< wave_min_draws): the lanes of
 * a wave write one contiguous piece per row.  Wave-mode positions stage Gaussian-major, wave_rows f32[rows][3] with the
 * rows of position p of bin b at wave_row_start[b] + (p - bin_start[b]) * (quota[b] - 1) (g2pc_sampler_stage_plan fills
 * wave_row_start u64[num_bins + 1], [num_bins] = total): the lanes of a wave write consecutive rows.  have_before
 * u32[num_attempts][gv] (same chunking as dcount) = rows a position had before the attempt.  Results are bit for bit those of
 * g2pc_sampler_count + g2pc_sampler_emit_rows. */
typedef struct G2pcSampleStage {
    float* thread_rows;
    float* wave_rows;
    const uint64_t* wave_row_start;
} G2pcSampleStage;
int g2pc_sampler_stage_plan(const uint32_t* bin_start, const int32_t* quota, int32_t num_bins, int32_t wave_min_draws,
                            uint64_t* wave_row_start, void* stream);
int g2pc_sampler_count_staged(const float* means, const float* cov9, const uint32_t* perm, const uint32_t* pbin,
                              const int32_t* quota, const uint32_t* bin_start, int64_t gv, int64_t p_wave_begin, float std_limit,
                              int32_t attempt0, int32_t num_attempts, uint64_t seed, uint64_t gid_base, uint32_t* added,
                              uint32_t* dcount, uint32_t* have_before, uint32_t* remaining, const G2pcSampleStage* stage,
                              void* stream);
int g2pc_sampler_emit_rows_staged(const float* means, const float* cov9, const float* colours, const float* normals,
                                  const uint32_t* perm, const uint32_t* bin_start, const int32_t* quota, int32_t num_bins,
                                  int32_t attempts, int64_t gv, int64_t p_wave_begin, const uint32_t* dscan,
                                  const uint32_t* have_before, const int64_t* sec_base, int64_t rows_capacity,
                                  const G2pcSampleStage* stage, float* out_points, float* out_colours, float* out_normals,
                                  int32_t* out_gauss, void* stream);

/* The sampler's tail in ONE call: g2pc_sampler_partition -> _stage_plan -> _count_staged (attempts 0 .. attempts-1, <= 8) ->
 * _scan_counts -> _sections -> _emit_rows_staged, with perm / pbin / counts / scans / section table / staging arrays all inside ONE
 * workspace of g2pc_sampler_run_workspace() bytes (the section table at g2pc_sampler_run_sections_offset() for diagnostics).
 * Same launches, same results; what it removes is the host work between them (create_new_gaussian_points' loop body is a
 * dozen torch calls in the reference, gauss_to_pc.py:195-263).  bin_of_ppg / quota / bin_start as g2pc_sampler_bin_table
 * leaves them; lane_planes / wave_rows size the staging (plan word 10; rows_ub - means_rows).  info_host: PINNED i64[2] <- {M,
 * Gaussians still short}. */
size_t g2pc_sampler_run_workspace(int64_t g, int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempts,
                                  int64_t lane_planes, int64_t wave_rows);
size_t g2pc_sampler_run_sections_offset(int64_t g, int64_t gv, int64_t p_wave_begin, int32_t num_bins, int32_t attempts,
                                        int64_t lane_planes, int64_t wave_rows);
int g2pc_sampler_run(const float* means, const float* cov9, const float* colours, const float* normals, const int32_t* ppg,
                     int64_t g, const int32_t* bin_of_ppg, int64_t lut_len, const int32_t* quota, const uint32_t* bin_start,
                     int32_t num_bins, int64_t gv, int64_t p_wave_begin, int32_t wave_min_draws, int64_t lane_planes,
                     int64_t wave_rows, float std_limit, int32_t attempts, uint64_t seed, uint64_t gid_base, int emit_means,
                     int64_t rows_capacity, float* out_points, float* out_colours, float* out_normals, int32_t* out_gauss,
                     int64_t* info_host, void* ws, size_t ws_bytes, void* stream);

/* --- stand-alone helpers of the python renderer (the reference's public gauss_render functions) ------------------
 * eval_sh (gauss_render.py:43-99): sh f32[n, channels, coeffs] (coefficient index on the LAST axis, as the
 * reference indexes it), dirs f32[n,3] (may be NULL for degree 0), out f32[n, channels]; degree 0..4. */
int g2pc_eval_sh(int32_t deg, const float* sh, const float* dirs, int64_t n, int32_t channels, int32_t coeffs, float* out,
                 void* stream);
/* build_covariance_2d (gauss_render.py:101-148): cov2d f32[n,2,2] = (J W S W^T J^T)[:2,:2] + 0.3 I; viewmatrix is
 * the HOST 4x4 world_view_transform (row-vector convention, 16 floats); lim = 1.3 * tan(fov / 2), the frustum clamp of
 * gauss_render.py:128-129, formed by the caller in DOUBLE from the field of view (as the reference forms it from python
 * floats) and rounded to f32.  Evaluated in the order torch's CPU build evaluates the reference (csrc/py_project.inl). */
int g2pc_build_covariance_2d(const float* means3D, const float* cov9, int64_t n, const float* viewmatrix, float lim_x,
                             float lim_y, float focal_x, float focal_y, float* cov2d, void* stream);
/* projection_ndc (gauss_render.py:151-168): p_proj f32[n,4], p_view f32[n,4], in_mask u8[n] (p_view.z <= -1e-6);
 * viewmatrix / projmatrix are HOST 4x4 matrices (16 floats each). */
int g2pc_projection_ndc(const float* points, int64_t n, const float* viewmatrix, const float* projmatrix, float* p_proj,
                        float* p_view, uint8_t* in_mask, void* stream);
/* get_radius (gauss_render.py:171-180): radius f32[n] = 3 ceil(sqrt(max eigenvalue)), discriminant clipped at 0.1 */
int g2pc_get_radius(const float* cov2d, int64_t n, float* radius, void* stream);
/* get_rect (gauss_render.py:183-193): rect_min / rect_max f32[n,2] = pix -+ radius clipped to [0, width-1] x [0, height-1] */
int g2pc_get_rect(const float* pix_coord, const float* radii, int64_t n, float width, float height, float* rect_min,
                  float* rect_max, void* stream);

/* mahalanobis() (gauss_to_pc.py:92-103): out[i] = sqrt(d^T inv(cov_i) d), d = means_i - samples_i; NaN when the
 * quadratic form is negative (the caller's `<=` then rejects, as in the reference). */
int g2pc_mahalanobis(const float* means, const float* samples, const float* cov9, int64_t n, float* out,
                     void* stream);
/* sample_from_multivariate_normal (gauss_to_pc.py:140-155): out f32[n, g, 3] (sample-major like
 * MultivariateNormal.sample((n,))) = mean + chol(cov) eps(seed, gid_base + g, attempt, k). */
int g2pc_sample_mvn(const float* means, const float* cov9, int64_t g, int32_t n, uint64_t seed, uint64_t gid_base,
                    int32_t attempt, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rasteriser (replaces _C.rasterize_gaussians, rasterize_points.h:18-41 / rasterizer_impl.cu:197-352, and
 * the pure-torch GaussPythonRenderer, gauss_render.py:266-465).  One camera = front half, ONE 4-byte read-back
 * by the caller (offsets[n] = number of (tile, Gaussian) instances; the reference has the same hidden
 * read-back at rasterizer_impl.cu:289), back half.
 * ------------------------------------------------------------------------------------------- */
typedef struct G2pcCamera {          /* HOST struct, passed by value to the kernels */
    float view[16];                  /* world_view_transform / viewmatrix, row-major (row-vector convention) */
    float proj[16];                  /* PY: projection_matrix (P^T);  CUDA semantics: full projmatrix = view @ P^T */
    float tan_fovx, tan_fovy;        /* tan(fov/2) */
    float focal_x, focal_y;
    int32_t width, height;
    float bg[3];
    float lim_x, lim_y;              /* PY semantics: 1.3 * tan(fov / 2) formed in double (gauss_render.py:128-129); unused by
                                      * the CUDA semantics (forward.cu:83-84 multiplies in f32) */
} G2pcCamera;

typedef struct G2pcTileLayout {      /* HOST struct of DEVICE pointers: tiles = x-intervals times y-intervals */
    int32_t nx, ny;
    const int32_t* xs;               /* [nx] first pixel column of interval i */
    const int32_t* ws;               /* [nx] width */
    const int32_t* ys;               /* [ny] */
    const int32_t* hs;               /* [ny] */
    const int32_t* tile_seq;         /* [ny*nx] processing order of tile iy*nx+ix (python renderer: FIFO quad-tree order) */
    const int32_t* seq_tile;         /* [ny*nx] inverse of tile_seq */
    const int32_t* tile_pix_off;     /* [ny*nx+1] pixel offset of every tile inside the per-tile colour buffer */
    int32_t num_chunks;              /* blend work list: one wave64 per chunk = chunk_subblocks consecutive 8x8 pixel */
    const int32_t* chunk_tile;       /* [num_chunks]    sub-blocks of a tile (sub-blocks row-major inside the tile)   */
    const int32_t* chunk_pix0;       /* [num_chunks] first sub-block of the chunk; chunk_subblocks = 2: a | b << 16, two adjacent sub-blocks (b = 0xFFFF: none) */
    int32_t chunk_subblocks;         /* 2 (sub-blocks per wave; 1 and 4 exist in -DG2PC_EXPERIMENTS builds only) */
    int32_t seq_bits;                /* ABI 3: width of the tile-sequence field of the packed visibility keys, 12 .. 14 (0 = 12):
                                      * key = contribution bits << 32 | ~(camera_slot << (12 + seq_bits) | tile_seq << 12 | pixel).
                                      * ny*nx <= 1 << seq_bits tiles, camera slots 1 .. (1 << (20 - seq_bits)) - 1 (255 / 127 / 63).
                                      * Every camera whose keys meet in one best_key array must use the same width (a renderer
                                      * that needs a wider field later rebases its keys first, g2pc_raster_rebase_keys, or widens
                                      * them in place, g2pc_raster_repack_keys). */
    /* --- the reference's DATA-DEPENDENT quad-tree (gauss_render.py:311-335); all optional, zero = a plain tile grid ---
     * A camera whose tree departs from the fixed leaf grid is rendered in several passes over different layouts (the leaf
     * grid, then the children of overloaded leaves level by level) that share the camera's slot: */
    int32_t seq_base, seq_count;     /* seq_count != 0: tile_seq holds the sequence numbers as they go into the keys, all in
                                      * [seq_base, seq_base + seq_count); seq_tile is indexed by (number - seq_base); the colour
                                      * update / resolve of this layout leaves keys outside the range alone */
    const uint8_t* tile_mask;        /* [ny*nx], optional: image assembly paints only the pixels of tiles with a non-zero entry and
                                      * leaves every other pixel of `image` as it is (NULL: every pixel is written) */
    int32_t depth;                   /* splits above the leaves (nx == ny == 1 << depth), with the three tables below: */
    const int32_t* inner_x;          /* [(1 << depth) - 1][2]: first and last pixel column of the interior nodes' column intervals,
                                      * level k (0 = root) interval i at row (1 << k) - 1 + i */
    const int32_t* inner_y;          /* the same for rows */
    const int32_t* tile_stick;       /* [ny*nx]: bit k set = the leaf reaches beyond its level-k ancestor (odd splits) */
    const uint8_t* tile_force;       /* ABI 5, optional [ny*nx]: non-zero = a node of the size-driven tree that is still larger than
                                      * max_tile_size (image sizes whose border nodes fit the limit a level before the interior
                                      * ones, gauss_render.py:319): the gate treats it like a leaf over max_per_tile whenever it
                                      * holds a Gaussian -- empty range, state 1, its children rendered in further passes.
                                      * Value 2: the same, but the children follow as a STATIC pass of the same launch plan
                                      * (tile_parent below): state 3, not reported through count_host */
    const int32_t* tile_parent;      /* ABI 5, optional [ny*nx][G2PC_TILE_PARENTS] (16-byte aligned): this layout is the child level of
                                      * another one; entries = the tiles of that layout this tile is a child of, -1 = none (a child
                                      * reaches one pixel beyond an odd-sized parent, so neighbouring parents can have the SAME
                                      * rectangle among their children: up to two per axis).  With G2pcCameraJob.alive (see there):
                                      * the child pass of a camera -- static (the children of tile_force nodes, every camera) or on
                                      * demand (the children of the leaves a camera overloaded); a tile exists if ANY parent is split */
} G2pcTileLayout;

size_t g2pc_raster_front_workspace(int64_t n);
/* PY semantics front half (gauss_render.py:101-193,404-437): projection, EWA covariance, conic, radius, rect ->
 * tile ranges; depth sort (nearest first, ties in descending index = torch.sort + flip); tiles-touched scan.
 * means3D f32[n,3], cov9 f32[n,3,3], opacity f32[n], colours f32[n,3] (what the python renderer blends).
 * Out: rec f32[n,16] -- ONE 64-byte record per Gaussian with what the blend stages from it ((mx, my, A, B), (C, opacity,
 * z, radius), (r, g, b, -), pad): a blend lane gathers one cache line per list entry --, rect u32[n], sorted_idx u32[n] (Gaussian indices in depth order), offsets u32[n+1].  count_host (optional, PINNED host
 * memory): receives offsets[n] by an asynchronous copy queued on `stream` behind the kernels. */
int g2pc_raster_front_py(const G2pcCamera* cam, const G2pcTileLayout* layout, const float* means3D, const float* cov9,
                         const float* opacity, const float* colours, int64_t n, float* rec, uint32_t* rect,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream);
size_t g2pc_raster_back_workspace(int64_t num_instances, int32_t num_tiles);
/* PY semantics back half (gauss_render.py:290-402): duplicate, stable tile sort, ranges, blend with per-Gaussian
 * max-contribution / arg-max pixel, colour update, optional image (f32[H,W,3], already flipped as the reference
 * returns it).  rec / rect / sorted_idx / offsets from the front half; best_key u64[n] and colours_out f32[n,3] are the renderer's
 * running state (zero-initialised by the caller); tilebuf f32[tile_pix_off[T],3] scratch.  camera_slot in
 * [1, 255] (seq_bits 12; see G2pcTileLayout) must increase from camera to camera (call g2pc_raster_rebase_keys before wrapping around).
 * t_floor: 0 = exact python semantics; > 0 stops a pixel chunk once every pixel's transmittance is below it
 * (all later contributions and colour terms are then < t_floor).
 * phases: bit 0 = binning (duplicate, tile sort, ranges), bit 1 = blend, bit 2 = colour update + image; 7 = all
 * (the phases share `ws`).  The packed-key atomicMax is commutative, so the blends of different cameras may run
 * concurrently on different streams; only the colour updates must be issued in camera order.
 * max_per_tile (0 = no limit) / overflow_flag (optional, device u32, zeroed by the caller): a tile holding more than
 * max_per_tile Gaussians is NOT blended -- the reference subdivides such a leaf (gauss_render.py:319): the caller renders
 * its children in further passes (layouts with seq_base / seq_count / tile_mask) -- and overflow_flag receives max(tile load).
 * means3D / cov9 (optional, the front half's inputs): with them and the layout's quad-tree tables, a leaf whose members all
 * live in the strip it reaches beyond an EMPTY ancestor is not blended either (the reference paints that ancestor with the
 * background and never visits its children, gauss_render.py:311-314).  g2pc_raster_tile_states reports both decisions. */
int g2pc_raster_back_py(const G2pcCamera* cam, const G2pcTileLayout* layout, int64_t n, int64_t num_instances,
                        const float* rec, const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                        const float* means3D, const float* cov9, uint32_t camera_slot, float t_floor,
                        unsigned long long* best_key, float* colours_out, float* tilebuf, float* image, int phases,
                        uint32_t max_per_tile, uint32_t* overflow_flag, void* ws, size_t ws_bytes, void* stream);
/* --- the same camera without a host round trip (capture-safe) -----------------------------------------------------
 * The two-call API above makes the host read the instance count between the halves (as the reference does at
 * rasterizer_impl.cu:289) and costs ~35 kernel launches per camera; with ~0.5 ms of GPU work per camera the host's
 * launch rate becomes the bound.  g2pc_raster_camera_py runs preprocess .. blend with every per-camera value in
 * DEVICE memory: the camera, its slot and the transmittance floor in a G2pcCameraJob, the instance count L in the
 * workspace.  All launch geometry derives from (n, capacity, layout), so the call can be captured once
 * (g2pc_graph_capture_begin / _end) and replayed for every camera with one g2pc_graph_launch.
 *   job_host (optional, PINNED = device-mapped host memory, e.g. hipHostMalloc / torch pin_memory): the first queued
 *            kernel fetches it into job_dev through the device mapping -- inside a graph it is re-read at every replay,
 *            so the host only rewrites the pinned struct between replays.
 *   capacity: instances the buffers hold.  A camera with L > capacity is skipped as a whole (nothing is blended);
 *            count_host (optional, PINNED, written by a kernel through the device mapping) always receives the true L,
 *            so the caller can detect this and render the camera again with more room.
 *   phases:  bit 0 = preprocess .. tile ranges, bit 1 = blend; 3 = the whole camera; bit 3 (8) = the job's t_floor is 0
 *            ("to the letter"): blend with the kernel that keeps the reference's operation order.  (A caller that wants HIP events
 *            around the blend alone captures phase 1 and issues phase 2 directly: this runtime refuses event-record
 *            nodes inside a captured graph.)
 * The colour update is not part of the call: updates must be issued in camera order across streams
 * (g2pc_raster_camera_update_py, after the caller's cross-stream wait).  ws: g2pc_raster_camera_workspace bytes, private
 * to the stream (it holds p0/p1/rect/sorted_idx/offsets too). */
typedef struct G2pcCameraJob {       /* DEVICE (and pinned host staging) struct */
    G2pcCamera cam;
    uint32_t camera_slot;            /* [1,255], see g2pc_raster_back_py */
    float t_floor;
    uint32_t tilebuf_lo, tilebuf_hi; /* != 0: device address of THIS camera's per-tile colour buffer (overrides the tilebuf
                                      * argument baked into a captured graph) -- lets a caller keep one buffer per camera and
                                      * resolve the winners' colours once, g2pc_raster_resolve_colours_py */
    uint32_t reserved;
    uint32_t alive_lo, alive_hi;     /* ABI 5, optional: device address of u8[num_tiles of the PARENT layout].  A pass over a layout
                                      * WITHOUT tile_parent writes "tile t is split for this camera" there (the gate, one byte per
                                      * tile: a tile_force node holding a Gaussian, or a leaf holding more than max_per_tile); a
                                      * later pass of the same camera over the child level (a layout WITH tile_parent) exists for
                                      * the children of those tiles only -- the preprocess counts, the duplication emits and the
                                      * gate admits no other tile: the reference never visits the children of a node it did not
                                      * split (gauss_render.py:311-335) */
} G2pcCameraJob;
size_t g2pc_raster_camera_workspace(int64_t n, int64_t capacity, int32_t num_tiles);
/* After phase 1 of g2pc_raster_back_py (same ws, num_instances, num_tiles): Gaussians per tile (counts u32[T], optional) and
 * what the gate decided (states u32[T], optional): 0 = blended, 1 = more than max_per_tile members (not blended: to be split,
 * gauss_render.py:319), 2 | level << 8 = lies under the empty level-`level` node (not blended, gauss_render.py:311-314). */
int g2pc_raster_tile_states(const void* ws, size_t ws_bytes, int64_t num_instances, int32_t num_tiles, uint32_t* counts,
                            uint32_t* states, void* stream);
/* Gaussians per node for arbitrary pixel rectangles nodes i32[num_nodes][4] = (x0, y0, w, h) (device): the reference's
 * `tile_mask.sum()` (gauss_render.py:306-309) for the interior nodes of its quad-tree.  counts u32[num_nodes] (device). */
int g2pc_raster_node_counts(const G2pcCamera* cam, const float* means3D, const float* cov9, int64_t n, const int32_t* nodes,
                            int32_t num_nodes, uint32_t* counts, void* stream);
/* Widen the tile field of every key in place (old_seq_bits -> new_seq_bits, 12 <= old <= new <= 14), keeping camera slot, tile
 * sequence and pixel: legal while no key carries a camera slot above (1 << (20 - new_seq_bits)) - 1. */
int g2pc_raster_repack_keys(unsigned long long* best_key, int64_t n, int32_t old_seq_bits, int32_t new_seq_bits, void* stream);
/* BATCHED form: `batch` cameras (1 .. G2PC_MAX_CAMERA_BATCH) through ONE launch sequence -- every kernel of the sequence runs
 * with grid.y = batch, camera c reading the c-th job of the arrays jobs_dev / jobs_host and working in the c-th of `batch`
 * consecutive workspaces of g2pc_raster_camera_workspace() bytes each (ws_bytes >= batch times that).  A 50-camera job is
 * then 13 dependent chains of ~20 launches instead of 50, each kernel four times as wide, and the blends of a batch drain
 * together.  count_host: u32[batch][4] = (instances, depth sort gave up, load of an overloaded leaf or 0, reserved): a camera
 * with [2] != 0 had leaves over max_per_tile that were left out (g2pc_raster_back_py) and needs their children rendered.  The loop over cameras this replaces: gauss_to_pc.py:437-454 (the reference renders
 * them one at a time).  g2pc_raster_camera_py is batch = 1. */
#define G2PC_MAX_CAMERA_BATCH 8
int g2pc_raster_cameras_py(const G2pcCameraJob* jobs_dev, const G2pcCameraJob* jobs_host, int32_t batch,
                           const G2pcTileLayout* layout, const float* means3D, const float* cov9, const float* opacity,
                           const float* colours, int64_t n, int64_t capacity, unsigned long long* best_key, float* tilebuf,
                           uint32_t* count_host, uint32_t max_per_tile, uint32_t* overflow_flag, int phases, void* ws,
                           size_t ws_bytes, void* stream);
int g2pc_raster_camera_py(const G2pcCameraJob* job_dev, const G2pcCameraJob* job_host, const G2pcTileLayout* layout,
                          const float* means3D, const float* cov9, const float* opacity, const float* colours, int64_t n,
                          int64_t capacity, unsigned long long* best_key, float* tilebuf, uint32_t* count_host,
                          uint32_t max_per_tile, uint32_t* overflow_flag, int phases, void* ws, size_t ws_bytes,
                          void* stream);
int g2pc_raster_camera_update_py(const G2pcTileLayout* layout, int64_t n, uint32_t camera_slot,
                                 const unsigned long long* best_key, const float* tilebuf, float* colours_out, void* stream);
/* Deferred form of the colour update (gauss_render.py:387-395): ONE pass after all blends of an epoch.  tilebufs: DEVICE
 * array of 256 addresses, [slot] = the per-tile colour buffer camera `slot` rendered into (0: leave the Gaussians won
 * by that slot untouched -- e.g. cameras that went through g2pc_raster_back_py, which updates at once).  All listed
 * cameras share `layout`.  Replaces the per-camera update and its camera-order chaining across streams. */
int g2pc_raster_resolve_colours_py(const G2pcTileLayout* layout, int64_t n, const unsigned long long* best_key,
                                   const unsigned long long* tilebufs, float* colours_out, void* stream);
/* Depth order inside g2pc_raster_cameras_py: a range-normalised bucket sort whose last kernel also emits the (tile, Gaussian)
 * instances (up to ~2 M Gaussians; a four-pass radix sort beyond).  count_host holds FOUR words per camera: [0] = instance
 * count, [1] != 0 = the bucket sort met a pile-up of equal depths (more than 1 024 keys in 1 / 4 096 of the depth range): the
 * camera was skipped as a whole and has to be rendered again through the two-call path (which always sorts by radix).
 * ABI 6: the library has no process-global state.  The tuning and diagnostic entry points of ABI <= 5 (g2pc_set_sort_tuning,
 * g2pc_set_depth_sort, g2pc_set_blend_variant, g2pc_debug_set_walk_cap / _head_threads / _extra_launches,
 * g2pc_raster_debug_chunk_work) and the blend kernels nothing selects exist only in -DG2PC_EXPERIMENTS builds
 * (csrc/experiments/, tools/experiments/build_variant.sh); G2pcTileLayout.chunk_subblocks must be 2. */
int g2pc_raster_rebase_keys(unsigned long long* best_key, int64_t n, void* stream);
/* --- native-rasteriser ("cuda") semantics: _C.rasterize_gaussians (rasterize_points.h:18-41) ----------------------
 * Deterministic spec of SURVEY.md §8(a.5): 16x16 tiles, near cull z_view <= 0.2, radius ceil(3 sqrt(lambda_max)),
 * stable (tile, depth) order, alpha rules (power > 0 skip, min(0.99, .), alpha < 1/255 skip, T(1-alpha) < 1e-4 stop),
 * per-Gaussian max contribution + arg-max pixel (ties -> lowest pixel id), surface distance against the expected
 * depth at the end of every 256-instance batch, optional mask (i32[H*W], 0 = skip the pixel) and SH colours
 * (sh f32[n, sh_coeffs, 3], degree <= 3, forward.cu:22-73).  cam->view = viewmatrix, cam->proj = FULL projmatrix
 * (view @ proj), cam->tan_fovx/y; focal = W / (2 tan_fovx) (rasterizer_impl.cu:229-230).
 * front: rec f32[n,16] (one 64-byte record per Gaussian: (x, y, conic a, b), (conic c, opacity, depth, radius), (r, g, b, -),
 * pad), rect u32[n] (u32[2n] for images beyond 4096 pixels a side: tile rectangles then take 16-bit coordinates, two
 * words per Gaussian), radii i32[n], sorted_idx u32[n], offsets u32[n+1] out. */
/* ABI 7: sh_coeffs < 0 in g2pc_raster_front_cu / g2pc_raster_camera_cu = |sh_coeffs| coefficients per Gaussian in PLANE-MAJOR
 * layout, f32[3K/4][n][4] (K a multiple of 4), as g2pc_sh_planes produces it from the reference's f32[n][K][3]: a wave reads
 * 1 KB contiguous per 16-byte vector instead of 64 pieces 192 bytes apart.  The scene is static over a job's cameras
 * (gauss_to_pc.py:437-454), so the binding transposes once per job; results are bit for bit the same. */
int g2pc_sh_planes(const float* shs, int64_t n, int32_t sh_coeffs, float* planes, void* stream);
int g2pc_raster_front_cu(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                         const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                         const float* campos, int64_t n, float* rec, uint32_t* rect, int32_t* radii,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream);
/* back: out_color f32[3,H,W], out_depth / out_invdepth f32[H,W] (zero-filled here, masked pixels stay 0);
 * running state max_contrib / total_contrib / min_surf f32[n], colours f32[n,3] updated as the reference's binding does
 * (gaussian_pointcloud_rasterization/__init__.py:128-158); winner_cam i32[n] (optional) records cam_index of the camera
 * that set the running maximum (multi-GPU tie-break); cur_* (optional) = this camera's gauss_contributions,
 * gauss_pixels, gauss_surface_distances.  phases: bit 0 = binning + zero fills, bit 1 = blend, bit 2 = running-state
 * update (must be issued in camera order; bits 0-1 of different cameras may overlap on different streams with
 * per-stream scratch).  Workspace size: g2pc_raster_back_workspace(num_instances, tiles). */
int g2pc_raster_back_cu(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t num_instances, const float* rec,
                        const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                        int calculate_surface_distance, unsigned long long* cam_key,
                        uint32_t* cam_surf, float* out_color, float* out_depth, float* out_invdepth,
                        float* max_contrib, float* total_contrib, float* colours, float* min_surf,
                        int32_t* winner_cam, int32_t cam_index, float* cur_contrib, int32_t* cur_pixels, float* cur_surf,
                        int phases, void* ws, size_t ws_bytes, void* stream);
/* The same with a tile shard: only tiles tile_first, tile_first + tile_step, ... are blended (multi-GPU jobs with fewer
 * cameras than ranks split every camera's 16x16 tiles over the ranks); the images hold zeros elsewhere, cam_key / cam_surf
 * cover this rank's tiles only and are merged by the caller (MAX / MIN over ranks, SUM of the images) before phase 4. */
int g2pc_raster_back_cu_tiles(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t num_instances, const float* rec,
                              const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                              int calculate_surface_distance, unsigned long long* cam_key, uint32_t* cam_surf,
                              float* out_color, float* out_depth, float* out_invdepth, float* max_contrib,
                              float* total_contrib, float* colours, float* min_surf, int32_t* winner_cam, int32_t cam_index,
                              float* cur_contrib, int32_t* cur_pixels, float* cur_surf, int phases, int32_t tile_first,
                              int32_t tile_step, void* ws, size_t ws_bytes, void* stream);
/* ABI 6: ONE pipelined camera of the native-rasteriser semantics in one call, without the host in the loop and without a separate
 * duplication: preprocess (forward.cu:153-271) -> depth bucket sort whose last kernel emits the (tile, Gaussian) instances (what
 * rasterizer_impl.cu:285-316 does with an InclusiveSum, duplicateWithKeys and a 64-bit SortPairs) -> stable tile sort -> ranges
 * (rasterizer_impl.cu:322-331) -> blend (forward.cu:303-497).  Results bit for bit those of g2pc_raster_front_cu +
 * g2pc_raster_back_cu_dev.  count_host (PINNED, optional): [0] instances, [1] != 0: the depths piled up beyond a bucket's room;
 * a camera with [0] > capacity or [1] != 0 was skipped as a whole: render it again with g2pc_raster_front_cu / _back_cu.
 * Returns G2PC_ERR_UNSUPPORTED for grids beyond 256 x 256 tiles or more than ~2 M Gaussians (use the two calls).
 * Follow with g2pc_raster_back_cu_tiles(phases = 4, num_instances = capacity, ws = this ws) for the running-state update -- it
 * reads none of rect / sorted_idx / offsets. */
size_t g2pc_raster_camera_cu_workspace(int64_t n, int64_t capacity, int32_t num_tiles);
int g2pc_raster_camera_cu(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                          const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                          const float* campos, const int32_t* mask, int64_t n, int64_t capacity, float* rec, uint32_t* rect,
                          int32_t* radii, int calculate_surface_distance, unsigned long long* cam_key, uint32_t* cam_surf,
                          float* out_color, float* out_depth, float* out_invdepth, uint32_t* count_host, int32_t tile_first,
                          int32_t tile_step, void* ws, size_t ws_bytes, void* stream);
/* Bin + blend of one camera with the instance count kept on the device: launches sized for `capacity`, the count goes to
 * the pinned count_host[0] (count_host[1] = 0) asynchronously; a camera with more instances is skipped as a whole and
 * must be rendered again through g2pc_raster_back_cu[_tiles].  Removes the host read-back between the halves of a camera
 * (rasterizer_impl.cu:289 has it).  Follow with g2pc_raster_back_cu_tiles(phases = 4, num_instances = capacity). */
int g2pc_raster_back_cu_dev(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t capacity, const float* rec,
                            const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                            int calculate_surface_distance, unsigned long long* cam_key, uint32_t* cam_surf, float* out_color,
                            float* out_depth, float* out_invdepth, uint32_t* count_host, int32_t tile_first, int32_t tile_step,
                            void* ws, size_t ws_bytes, void* stream);

/* --- ABI 7: the reference's native entry point itself ---------------------------------------------------------------------
 * _C.rasterize_gaussians (gaussian-pointcloud-rasterization/rasterize_points.h:18-41, rasterize_points.cu:36-145, bound at
 * ext.cpp:15-18): the same 22 arguments in the same order, the same 11 results.  The reference's UNMODIFIED binding
 * gaussian_pointcloud_rasterization/__init__.py:90-158 runs on it through the ctypes module
 * 3dgs-to-pc_amd/gaussian_pointcloud_rasterization/_C.py.
 *   - tensors the reference signals as absent by an EMPTY tensor (colors, scales, rotations, cov3D_precomp, sh; forward.cu:204,251
 *     test their data pointers) are NULL here; exactly one of (colors, sh) and of (cov3D_precomp, scales + rotations);
 *   - background f32[3], viewmatrix f32[16], projmatrix f32[16] (the FULL projection, view @ proj), campos f32[3] are HOST
 *     arrays (16 + 16 + 6 floats that become kernel arguments); everything else is a device pointer;
 *   - scales are ACTIVATED (not log-space) and the quaternion is used as given: computeCov3D, forward.cu:115-150, evaluated
 *     operation by operation (no contraction) -- bit for bit what oracle/_ref computes from the same inputs;
 *   - antialiasing multiplies the opacity by sqrt(max(0.000025, det(cov) / det(cov + 0.3 I))) (forward.cu:217-225,264);
 *   - prefiltered is accepted and ignored: the reference traps the DEVICE when a Gaussian fails the frustum test although
 *     `prefiltered` was set (auxiliary.h:168-172); this library culls such a Gaussian as it does without the flag;
 *   - debug != 0: the stream is synchronised before returning and a failed launch is reported (the reference's CHECK_CUDA,
 *     auxiliary.h:178-185);
 *   - scratch: the reference grows three byte tensors through std::function<char*(size_t)> callbacks (rasterizer.h:32-34,
 *     rasterize_points.cu:25-34) and returns them; here the three G2pcResizeFn play that role -- each is called exactly once per
 *     rasterisation with the bytes needed (geometry: per-Gaussian state incl. the per-camera visibility keys; binning: instances
 *     and sort space, sized from num_rendered; image: tile ranges) and must return device memory of at least that size that
 *     stays valid until the work queued on `stream` has run;
 *   - like the reference (rasterizer_impl.cu:289) the call blocks ONCE on the host, to read the instance count between its
 *     halves -- the only entry point of this header that synchronises (the pipelined production path is g2pc_raster_camera_cu).
 * Outputs (device, caller-allocated; rasterize_points.cu:73-93 gives their shapes and initial values, which are written here):
 * out_color f32[3,H,W], out_depth f32[1,H,W], radii i32[P], out_invdepth f32[1,H,W], gauss_contributions f32[P] (0 where never
 * blended), gauss_surface_distances f32[P] (FLT_MAX where not measured), gauss_pixels i32[P] (0 where never blended);
 * *num_rendered = number of (tile, Gaussian) instances.  P == 0: the images are zero-filled, nothing else happens. */
typedef void* (*G2pcResizeFn)(void* user, size_t bytes);
typedef struct G2pcRasterizeArgs {      /* rasterize_points.h:18-41, argument for argument */
    const float* background;            /* HOST f32[3] */
    const float* means3D;               /* f32[P,3] */
    const float* colors;                /* f32[P,3] or NULL */
    const float* opacity;               /* f32[P] (the reference passes [P,1]) */
    const float* scales;                /* f32[P,3] or NULL */
    const float* rotations;             /* f32[P,4] or NULL */
    float scale_modifier;
    const float* cov3D_precomp;         /* f32[P,6] or NULL */
    const float* viewmatrix;            /* HOST f32[16], as torch stores the [4,4] tensor (row-major = glm column-major) */
    const float* projmatrix;            /* HOST f32[16] */
    float tan_fovx, tan_fovy;
    int32_t image_height, image_width;
    const float* sh;                    /* f32[P,M,3] or NULL */
    int32_t degree;
    const float* campos;                /* HOST f32[3] */
    const int32_t* mask;                /* i32[H*W], 0 = skip the pixel; NULL = all ones (the binding's default, __init__.py:95-98) */
    int32_t prefiltered, antialiasing, calculate_surface_distance, debug;
    /* sizes torch tensors carry themselves */
    int64_t P;                          /* means3D.size(0) */
    int32_t M;                          /* sh.size(1), 0 without SHs */
} G2pcRasterizeArgs;
typedef struct G2pcRasterizeOut {       /* results 2-4 and 8-11 of the reference's tuple (1 = *num_rendered, 5-7 = the buffers) */
    float* out_color; float* out_depth; int32_t* radii; float* out_invdepth;
    float* gauss_contributions; float* gauss_surface_distances; int32_t* gauss_pixels;
} G2pcRasterizeOut;
int g2pc_rasterize_gaussians(const G2pcRasterizeArgs* args, const G2pcRasterizeOut* out, int32_t* num_rendered,
                             G2pcResizeFn geometry_buffer, void* geometry_user, G2pcResizeFn binning_buffer, void* binning_user,
                             G2pcResizeFn image_buffer, void* image_user, void* stream);

/* Multi-GPU exchange of the python-semantics state (cameras sharded over ranks): all-reduce MAX of best_key, then
 * g2pc_raster_key_owner (owner[i] = rank where this rank holds the winning key, INT32_MAX elsewhere), all-reduce MIN of
 * owner, g2pc_raster_keep_winner_colours (zero the colours of every Gaussian another rank was elected for), all-reduce
 * SUM of the colours: exactly one non-zero term per Gaussian, also for keys several ranks share after an earlier exchange. */
int g2pc_raster_key_owner(const unsigned long long* local_key, const unsigned long long* global_key, int64_t n,
                          int32_t rank, int32_t* owner, void* stream);
int g2pc_raster_keep_winner_colours(const int32_t* owner, int64_t n, int32_t rank, float* colours, void* stream);
/* _C.mark_visible (rasterize_points.h:43-46, unused by the reference's own pipeline): present[i] = the Gaussian centre is
 * in front of the near plane of the native rasteriser (z_view > 0.2, auxiliary.h:166).  viewmatrix: HOST float[16] as in
 * G2pcCamera.view. */
int g2pc_mark_visible(const float* means3D, int64_t n, const float* viewmatrix, uint8_t* present, void* stream);
/* gaussian_max_contribution f32[n] out of the packed keys (gauss_render.py:243-264 getters read this) */
int g2pc_raster_contributions(const unsigned long long* best_key, int64_t n, float* out, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* G2PC_H */
