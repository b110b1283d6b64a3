// Statistical outlier removal of the sampled cloud ("next" row f4 of SURVEY.md §8).
//
// Reference: mesh_handler.py:89-94 clean_point_cloud -> Open3D (not vendored, version unpinned)
// geometry::PointCloud::RemoveStatisticalOutliers(nb_neighbors = 20, std_ratio = 10): for every point the mean of the
// Euclidean distances to its nb_neighbors nearest points (KD-tree kNN in float64, the point itself included at
// distance 0, distances summed in ascending order); a point is kept when 0 < mean < cloud_mean + std_ratio * std.
//
// Here: exact kNN on a uniform grid.  Points are binned into cubic cells (radix sort by cell id), every thread owns one
// point (in cell order, so a wave reads neighbouring cells) and searches growing cubic shells of cells until the k-th
// distance found is provably final -- no unsearched cell can hold a closer point.  All distance arithmetic is float64
// on the float32 coordinates, in the KD-tree's order ((dx^2 + dy^2) + dz^2), so the means equal Open3D's bit for bit
// up to the order of equal distances.
#include <hip/hip_runtime.h>

#include "g2pc_internal.h"

namespace g2pc {

constexpr int CL_T = 256;
__global__ void k_tile_ranges(const uint32_t* __restrict__ tile_sorted, long L, int T, uint32_t* __restrict__ tile_start,
                              const uint32_t* __restrict__ l_dev, int gshift, size_t cs);   // raster.hip: boundaries of sorted ids

struct Grid {
    float ox, oy, oz, h, inv_h;
    int nx, ny, nz;
};

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h, int n) {
    int c = (int)floorf((x - o) * inv_h);          // monotone in x: the search bounds below rely on that
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ __launch_bounds__(CL_T) void k_cell_keys(const float* __restrict__ pts, long m, Grid g,
                                                   uint32_t* __restrict__ key, uint32_t* __restrict__ idx) {
    long i = (long)blockIdx.x * CL_T + threadIdx.x;
    if (i >= m) return;
    int cx = cell_coord(pts[3 * i], g.ox, g.inv_h, g.nx);
    int cy = cell_coord(pts[3 * i + 1], g.oy, g.inv_h, g.ny);
    int cz = cell_coord(pts[3 * i + 2], g.oz, g.inv_h, g.nz);
    key[i] = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(CL_T) void k_gather_sorted(const float* __restrict__ pts, const uint32_t* __restrict__ idx,
                                                       long m, float4* __restrict__ spos) {
    long i = (long)blockIdx.x * CL_T + threadIdx.x;
    if (i >= m) return;
    uint32_t j = idx[i];
    spos[i] = make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], __uint_as_float(j));
}

__global__ __launch_bounds__(CL_T) void k_count_occupied(const uint32_t* __restrict__ cell_start, long cells,
                                                        uint32_t* __restrict__ occupied) {
    long c = (long)blockIdx.x * CL_T + threadIdx.x;
    int occ = (c < cells && cell_start[c + 1] > cell_start[c]) ? 1 : 0;
    int n = __syncthreads_count(occ);
    if (threadIdx.x == 0 && n) atomicAdd(occupied, (uint32_t)n);
}

// K nearest squared distances, ascending, in registers (fully unrolled compare-exchange insertion)
template <int K>
__device__ __forceinline__ void knn_insert(double (&best)[K], double d2) {
#pragma unroll
    for (int j = K - 1; j >= 1; --j) {
        const double lo = best[j - 1];
        best[j] = d2 < lo ? lo : (d2 < best[j] ? d2 : best[j]);
    }
    best[0] = d2 < best[0] ? d2 : best[0];
}

// Queries: every point in cell order (query_idx == nullptr), or the listed original indices (a later level of the cascade:
// points that a finer grid could not settle within its shell budget).  A query that is still open after max_rings
// shells is appended to `unresolved` instead of being answered.
template <int K>
__global__ __launch_bounds__(CL_T) void k_knn_mean(const float4* __restrict__ spos, const uint32_t* __restrict__ cell_start,
                                                  long m, Grid g, int k, double slack, const float* __restrict__ points,
                                                  const uint32_t* __restrict__ query_idx, long num_queries, int max_rings,
                                                  uint32_t* __restrict__ unresolved, uint32_t* __restrict__ unresolved_count,
                                                  double* __restrict__ avg) {
    long i = (long)blockIdx.x * CL_T + threadIdx.x;
    if (i >= num_queries) return;
    float4 p;
    if (query_idx) {
        const uint32_t q = query_idx[i];
        p = make_float4(points[3 * (size_t)q], points[3 * (size_t)q + 1], points[3 * (size_t)q + 2], __uint_as_float(q));
    } else {
        p = spos[i];
    }
    const int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx), cy = cell_coord(p.y, g.oy, g.inv_h, g.ny),
              cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
    const double px = p.x, py = p.y, pz = p.z;
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = 1.0e300;
    const int kk = (long)k < m ? k : (int)m;                   // a cloud smaller than k: all of it (Open3D: dist.size())
    const int rmax = min(max(max(g.nx, g.ny), g.nz), max_rings);
    bool settled = false;
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
        auto visit = [&](int x, int y, int z) {
            const long c = ((long)z * g.ny + y) * g.nx + x;
            const uint32_t b = cell_start[c], e = cell_start[c + 1];
            for (uint32_t q = b; q < e; ++q) {
                const float4 s = spos[q];
                const double dx = px - (double)s.x, dy = py - (double)s.y, dz = pz - (double)s.z;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < best[K - 1]) knn_insert<K>(best, d2);
            }
        };
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                // cells at Chebyshev distance exactly r (the rest of the block was searched at r - 1): on a y or z face
                // of the shell the whole x run is new, elsewhere only its two end cells
                if (z == cz - r || z == cz + r || y == cy - r || y == cy + r) {
                    for (int x = x0; x <= x1; ++x) visit(x, y, z);
                } else {
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (cx + r <= g.nx - 1) visit(cx + r, y, z);
                }
            }
        // every unsearched point lies outside the block of cells [c - r, c + r]^3: at least `bound` away
        double bound = 1.0e300;
        if (cx - r > 0) bound = fmin(bound, px - ((double)g.ox + (double)(cx - r) * (double)g.h));
        if (cx + r < g.nx - 1) bound = fmin(bound, ((double)g.ox + (double)(cx + r + 1) * (double)g.h) - px);
        if (cy - r > 0) bound = fmin(bound, py - ((double)g.oy + (double)(cy - r) * (double)g.h));
        if (cy + r < g.ny - 1) bound = fmin(bound, ((double)g.oy + (double)(cy + r + 1) * (double)g.h) - py);
        if (cz - r > 0) bound = fmin(bound, pz - ((double)g.oz + (double)(cz - r) * (double)g.h));
        if (cz + r < g.nz - 1) bound = fmin(bound, ((double)g.oz + (double)(cz + r + 1) * (double)g.h) - pz);
        if (bound > 1.0e299) { settled = true; break; }        // the block covers the whole grid
        bound -= slack;                                        // rounding of the float cell assignment
        double kth = 1.0e300;
#pragma unroll
        for (int j = 0; j < K; ++j) if (j == kk - 1) kth = best[j];
        if (bound > 0.0 && kth <= bound * bound) { settled = true; break; }
    }
    if (!settled) {                                            // shell budget spent: a coarser grid takes over
        unresolved[atomicAdd(unresolved_count, 1u)] = __float_as_uint(p.w);
        return;
    }
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) if (j < kk) sum += sqrt(best[j]);
    avg[__float_as_uint(p.w)] = sum / (double)kk;
}

// Later levels of the cascade: few queries, each with a large candidate set (a far-away floater ends up scanning the
// whole cloud) -> one 256-thread block per query.  A row of cells along x is one contiguous range of the cell-sorted
// array, so the threads stride through the (2R+1)^2 rows of the radius-R block with coalesced 16-byte loads, each
// keeping a private top-K; the K smallest of the 256 lists are then extracted by K rounds of a block-wide arg-min.
template <int K>
__global__ __launch_bounds__(CL_T) void k_knn_mean_block(const float4* __restrict__ spos, const uint32_t* __restrict__ cell_start,
                                                        long m, Grid g, int k, double slack, const float* __restrict__ points,
                                                        const uint32_t* __restrict__ query_idx, int rings,
                                                        uint32_t* __restrict__ unresolved, uint32_t* __restrict__ unresolved_count,
                                                        double* __restrict__ avg) {
    __shared__ double s_list[CL_T][K + 1];
    __shared__ double s_wmin[CL_T / kWave];
    __shared__ int s_wtid[CL_T / kWave];
    __shared__ double s_result[K];
    const uint32_t q = query_idx[blockIdx.x];
    const double px = points[3 * (size_t)q], py = points[3 * (size_t)q + 1], pz = points[3 * (size_t)q + 2];
    const int cx = cell_coord((float)px, g.ox, g.inv_h, g.nx), cy = cell_coord((float)py, g.oy, g.inv_h, g.ny),
              cz = cell_coord((float)pz, g.oz, g.inv_h, g.nz);
    const int r = rings;
    const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
    const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = 1.0e300;
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const long row = ((long)z * g.ny + y) * g.nx;
            const uint32_t b = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (uint32_t i0 = b; i0 < e; i0 += 2 * CL_T) {
                const uint32_t ia = i0 + threadIdx.x, ib = ia + CL_T;
                float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
                if (ia < e) sa = spos[ia];                         // two loads in flight per thread
                if (ib < e) sb = spos[ib];
                if (ia < e) {
                    const double dx = px - (double)sa.x, dy = py - (double)sa.y, dz = pz - (double)sa.z;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best[K - 1]) knn_insert<K>(best, d2);
                }
                if (ib < e) {
                    const double dx = px - (double)sb.x, dy = py - (double)sb.y, dz = pz - (double)sb.z;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best[K - 1]) knn_insert<K>(best, d2);
                }
            }
        }
#pragma unroll
    for (int j = 0; j < K; ++j) s_list[threadIdx.x][j] = best[j];
    s_list[threadIdx.x][K] = 1.0e300;
    __syncthreads();
    const int kk = (long)k < m ? k : (int)m;
    int head = 0;
    for (int round = 0; round < kk; ++round) {
        double v = s_list[threadIdx.x][head];
        int t = (int)threadIdx.x;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {                         // wave arg-min, ties to the lowest thread
            const double ov = __shfl_xor(v, o);
            const int ot = __shfl_xor(t, o);
            if (ov < v || (ov == v && ot < t)) { v = ov; t = ot; }
        }
        if ((threadIdx.x & 63) == 0) { s_wmin[threadIdx.x >> 6] = v; s_wtid[threadIdx.x >> 6] = t; }
        __syncthreads();
        double bv = s_wmin[0];
        int bt = s_wtid[0];
#pragma unroll
        for (int w = 1; w < CL_T / kWave; ++w)
            if (s_wmin[w] < bv || (s_wmin[w] == bv && s_wtid[w] < bt)) { bv = s_wmin[w]; bt = s_wtid[w]; }
        if ((int)threadIdx.x == bt) ++head;
        if (threadIdx.x == 0) s_result[round] = bv;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double bound = 1.0e300;
    if (cx - r > 0) bound = fmin(bound, px - ((double)g.ox + (double)(cx - r) * (double)g.h));
    if (cx + r < g.nx - 1) bound = fmin(bound, ((double)g.ox + (double)(cx + r + 1) * (double)g.h) - px);
    if (cy - r > 0) bound = fmin(bound, py - ((double)g.oy + (double)(cy - r) * (double)g.h));
    if (cy + r < g.ny - 1) bound = fmin(bound, ((double)g.oy + (double)(cy + r + 1) * (double)g.h) - py);
    if (cz - r > 0) bound = fmin(bound, pz - ((double)g.oz + (double)(cz - r) * (double)g.h));
    if (cz + r < g.nz - 1) bound = fmin(bound, ((double)g.oz + (double)(cz + r + 1) * (double)g.h) - pz);
    bool settled = bound > 1.0e299;                                // the block covers the whole grid
    if (!settled) {
        bound -= slack;
        settled = bound > 0.0 && s_result[kk - 1] <= bound * bound;
    }
    if (!settled) {
        unresolved[atomicAdd(unresolved_count, 1u)] = q;
        return;
    }
    double sum = 0.0;
    for (int j = 0; j < kk; ++j) sum += sqrt(s_result[j]);
    avg[q] = sum / (double)kk;
}

static Grid make_grid(const float* origin, float cell, const int32_t* dims) {
    Grid g;
    g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
    g.h = cell; g.inv_h = 1.0f / cell;
    g.nx = dims[0]; g.ny = dims[1]; g.nz = dims[2];
    return g;
}
static int bits_for(unsigned v) { int b = 1; while ((1ull << b) < v && b < 32) ++b; return b; }

}  // namespace g2pc

extern "C" {

size_t g2pc_outlier_grid_workspace(int64_t m) {
    using namespace g2pc;
    return align_up((size_t)m * 4) * 6 + sort_workspace((long)m) + 4096;
}

int g2pc_outlier_grid_build(const float* points, int64_t m, const float* origin, float cell, const int32_t* dims,
                            float* sorted_pos, uint32_t* cell_start, uint32_t* occupied, void* ws, size_t ws_bytes,
                            void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(points && origin && dims && sorted_pos && cell_start && ws && m > 0 && cell > 0.0f, G2PC_ERR_ARG,
                 "bad arguments");
    const long cells = (long)dims[0] * dims[1] * dims[2];
    G2PC_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && cells < (1l << 31), G2PC_ERR_ARG, "bad grid dimensions");
    hipStream_t s = (hipStream_t)stream;
    Grid g = make_grid(origin, cell, dims);
    Arena ar(ws, ws_bytes);
    uint32_t* key = ar.get<uint32_t>((size_t)m);
    uint32_t* idx = ar.get<uint32_t>((size_t)m);
    uint32_t* key_s = ar.get<uint32_t>((size_t)m);
    uint32_t* idx_s = ar.get<uint32_t>((size_t)m);
    uint32_t* ktmp = ar.get<uint32_t>((size_t)m);
    uint32_t* vtmp = ar.get<uint32_t>((size_t)m);
    size_t sort_bytes = sort_workspace((long)m);
    char* sort_ws = ar.get<char>(sort_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_cell_keys, dim3(cdiv(m, CL_T)), dim3(CL_T), 0, s, points, (long)m, g, key, idx);
    int rc = sort_pairs_u32(key, idx, key_s, idx_s, ktmp, vtmp, (long)m, 0, bits_for((unsigned)cells), sort_ws, sort_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(m + 1, CL_T)), dim3(CL_T), 0, s, key_s, (long)m, (int)cells, cell_start,
                       (const uint32_t*)nullptr, 0, (size_t)0);
    hipLaunchKernelGGL(k_gather_sorted, dim3(cdiv(m, CL_T)), dim3(CL_T), 0, s, points, idx_s, (long)m, (float4*)sorted_pos);
    if (occupied) {
        hipMemsetAsync(occupied, 0, sizeof(uint32_t), s);
        hipLaunchKernelGGL(k_count_occupied, dim3(cdiv(cells, CL_T)), dim3(CL_T), 0, s, cell_start, cells, occupied);
    }
    return check_launch("g2pc_outlier_grid_build");
}

int g2pc_outlier_knn_mean_distance(const float* sorted_pos, const uint32_t* cell_start, int64_t m, const float* origin,
                                   float cell, const int32_t* dims, int32_t k, double slack, const float* points,
                                   const uint32_t* query_idx, int64_t num_queries, int32_t max_rings, uint32_t* unresolved,
                                   uint32_t* unresolved_count, double* avg, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(sorted_pos && cell_start && origin && dims && avg && unresolved && unresolved_count && m > 0 && cell > 0.0f,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(k >= 1 && k <= 32, G2PC_ERR_UNSUPPORTED, "k must be in [1, 32]");
    G2PC_REQUIRE(query_idx == nullptr || points != nullptr, G2PC_ERR_ARG, "query_idx needs the points in original order");
    hipStream_t s = (hipStream_t)stream;
    Grid g = make_grid(origin, cell, dims);
    const long nq = query_idx ? (long)num_queries : (long)m;
    hipMemsetAsync(unresolved_count, 0, sizeof(uint32_t), s);
    if (nq <= 0) return G2PC_OK;
    if (query_idx) {                 // a later level of the cascade: block per query
        if (k <= 20)
            hipLaunchKernelGGL(k_knn_mean_block<20>, dim3((unsigned)nq), dim3(CL_T), 0, s, (const float4*)sorted_pos, cell_start,
                               (long)m, g, (int)k, slack, points, query_idx, (int)max_rings, unresolved, unresolved_count, avg);
        else
            hipLaunchKernelGGL(k_knn_mean_block<32>, dim3((unsigned)nq), dim3(CL_T), 0, s, (const float4*)sorted_pos, cell_start,
                               (long)m, g, (int)k, slack, points, query_idx, (int)max_rings, unresolved, unresolved_count, avg);
    } else if (k <= 20)
        hipLaunchKernelGGL(k_knn_mean<20>, dim3(cdiv(nq, CL_T)), dim3(CL_T), 0, s, (const float4*)sorted_pos, cell_start, (long)m,
                           g, (int)k, slack, points, query_idx, nq, (int)max_rings, unresolved, unresolved_count, avg);
    else
        hipLaunchKernelGGL(k_knn_mean<32>, dim3(cdiv(nq, CL_T)), dim3(CL_T), 0, s, (const float4*)sorted_pos, cell_start, (long)m,
                           g, (int)k, slack, points, query_idx, nq, (int)max_rings, unresolved, unresolved_count, avg);
    return check_launch("g2pc_outlier_knn_mean_distance");
}

}  // extern "C"
