// Tile-binned splat rasteriser with per-Gaussian visibility accumulation (gfx950, wave64).
//
// Semantics "PY" = the reference's pure-torch renderer (the parity target named by BASELINE north_star):
//   gauss_render.py:101-148 build_covariance_2d, :151-168 projection_ndc, :171-193 get_radius/get_rect,
//   :266-402 GaussPythonRenderer.render (quad-tree leaf tiles, strict rect overlap, depth order, alpha
//   clip 0.99 with NO cut-offs, T = cumprod, per-Gaussian max of T*alpha over a tile and arg-max pixel,
//   strict-> running update of the contribution and of the colour = that tile's pixel colour).
// Kernel structure = the reference's native rasteriser re-designed for CDNA4:
//   preprocess (forward.cu:153-271) -> DEPTH sort of the N Gaussians (4 radix passes over N) -> duplicate in
//   depth order (rasterizer_impl.cu:69-110) -> stable sort of the L instances by TILE id only (2 passes over L,
//   instead of the reference's 6 passes over 64-bit keys, rasterizer_impl.cu:311-316) -> per-tile ranges ->
//   blend (forward.cu:303-497) with LDS-staged batches, 4 pixels per lane and wave64 reductions feeding a
//   packed 64-bit (contribution bits << 32 | ~order) atomicMax, which makes the cross-tile / cross-camera
//   arg-max exact and deterministic (the reference's CUDA kernel races here, SURVEY.md §2.2 defect 3).
#include "g2pc_internal.h"
#include "py_project.inl"
#include <type_traits>

namespace g2pc {

constexpr int RA_T = 256;
// Block size of the python-semantics head kernels without block-level cooperation (k_preprocess_py, k_duplicate, k_tile_ranges).
// A 256-thread block needs a free wave slot and its registers on all four SIMDs of ONE CU at the same moment -- rare while the
// single-wave blocks of another camera's blend hold 5 x 92 of every SIMD's 512 VGPRs; a 64-thread block goes wherever one
// wave fits (the reason k_bk_sort is one wave per block).  g2pc_debug_set_head_threads, 64 / 128 / 256.
static int g_head_threads = RA_T;
// DIAGNOSTIC (g2pc_debug_set_extra_launches): this many empty one-wave kernels are launched after the preprocess of every camera
// batch -- what does one more kernel boundary in a head chain cost the JOB?
static int g_extra_launches = 0;
__global__ void k_nothing(uint32_t* __restrict__ p) { if (p && threadIdx.x == 1000) p[0] = 0; }
__global__ void k_tile_ranges(const uint32_t* __restrict__ tile_sorted, long L, int T, uint32_t* __restrict__ tile_start,
                              const uint32_t* __restrict__ l_dev, int gshift, size_t cs);
constexpr float LOG2E = 1.4426950408889634f;

struct Cam {            // device copy of G2pcCamera (passed by value as kernel argument)
    float V[16];
    float P[16];
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H;
    float bg[3];
    float lim_x, lim_y;
};

struct Layout {         // device pointers of G2pcTileLayout
    int nx, ny, num_chunks, seq_bits;
    const int32_t *xs, *ws, *ys, *hs;
    const int32_t *tile_seq, *seq_tile, *tile_pix_off;
    int seq_base, seq_count;          // the keys of this layout carry sequence numbers [seq_base, seq_base + seq_count)
    const uint8_t* tile_mask;         // image assembly: compose only these tiles (nullptr = all, whole image written)
    int depth;                        // quad-tree info (python semantics): nx == ny == 1 << depth, 0 = none
    const int32_t *inner_x, *inner_y; // [(1 << depth) - 1][2] inclusive pixel extents of the interior nodes per axis
    const int32_t* tile_stick;        // [ny*nx] bit k: the leaf reaches beyond its level-k ancestor (nullptr = none does)
    const uint8_t* tile_force;        // [ny*nx] non-zero: always split when it holds a Gaussian (nullptr = none); 2 = children follow statically
    const int32_t* tile_parent;       // [ny*nx][G2PC_TILE_PARENTS] child level of another layout: the parent tiles there (-1 none), with G2pcCameraJob.alive
    int walk_cap;                     // DIAGNOSTIC (g2pc_debug_set_walk_cap): the dual-list blend stops a walk after this many batches (0 = never; results are then WRONG)
};

// ---------------------------------------------------------------------------------------------------------
// K1 (PY): per Gaussian projection, EWA covariance, conic, radius, pixel rect -> tile index ranges.
// Writes the depth-sort input in REVERSED index order so that the stable ascending radix sort leaves equal
// depths in descending index order = torch.sort (stable on CPU) followed by flip (gauss_render.py:340-342).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void interval_range(const int32_t* __restrict__ start, const int32_t* __restrict__ size,
                                               int n, float rmin, float rmax, int& i0, int& i1) {
    // Interval i overlaps iff  min(rmax, hi_i) > max(rmin, lo_i)  (rect_max.clip(max = tile end) > rect_min.clip(min = tile
    // start), gauss_render.py:306-309), lo_i = start, hi_i = start + size - 1.  lo and hi grow with i, so the overlapped
    // intervals are those between the first with hi_i > rmin and the last with lo_i < rmax: two binary searches instead of
    // a pass over all n (the pass was two thirds of this kernel's instructions), the reference's predicate itself deciding
    // the two ends.  NaNs compare false everywhere: nothing overlaps, as before.
    auto overlaps = [&](int i) {
        const float lo = (float)start[i], hi = (float)(start[i] + size[i] - 1);
        const float tl = rmin > lo ? rmin : lo;        // rect_min.clip(min = tile start)
        const float br = rmax < hi ? rmax : hi;        // rect_max.clip(max = tile end)
        return br > tl;
    };
    int a = 0, b = n;                                  // first i with hi_i > rmin
    while (a < b) { const int m = (a + b) >> 1; if ((float)(start[m] + size[m] - 1) > rmin) b = m; else a = m + 1; }
    int first = a;
    a = -1; b = n - 1;                                 // last i with lo_i < rmax
    while (a < b) { const int m = (a + b + 1) >> 1; if ((float)start[m] < rmax) a = m; else b = m - 1; }
    int last = a;
    while (first <= last && !overlaps(first)) ++first;
    while (last >= first && !overlaps(last)) --last;
    if (first <= last) { i0 = first; i1 = last; } else { i0 = n; i1 = -1; }
}

// MULTI (round 5): ONE thread per Gaussian for ALL cameras of the batch (grid.y = 1, `ncam` cameras in a loop) instead of one
// per (Gaussian, camera): the Gaussian's 17 input words, the tile-interval stage and the block's barriers are paid once per
// batch -- beside the blends of the other streams a wave of this kernel is a chain of round trips (stage, inputs, stores),
// and half as many waves walk it.  The arithmetic per camera is the same instruction sequence: bit-identical outputs.
template <bool CAM_ON_DEVICE, bool MULTI>
__global__ __launch_bounds__(RA_T) void k_preprocess_py(Cam cam_val, const Cam* __restrict__ cam_dev, Layout lay,
                                                       const float* __restrict__ means3D,
                                                       const float* __restrict__ cov9,
                                                       const float* __restrict__ opacity, long n,
                                                       uint32_t* __restrict__ depth_key_rev0,
                                                       uint32_t* __restrict__ index_rev0,
                                                       uint32_t* __restrict__ tiles_touched0,
                                                       const float* __restrict__ colours,
                                                       float4* __restrict__ rec0, uint32_t* __restrict__ rect0, size_t cs,
                                                       BucketHdr* __restrict__ mm0, uint32_t mm_slots, int ncam) {
    // device-resident camera: lets ONE captured launch sequence serve every camera (scalar loads, see below).
    // Batched launch (grid.y cameras, or MULTI): camera c's job is the c-th G2pcCameraJob, its outputs live in the c-th arena.
    // mm != nullptr: the depth keys go to the bucket sort (prims.hip), whose first pass -- the range of the keys -- is folded
    // in here: every block leaves (max ~key, max key) in slot blockIdx.x % mm_slots of the (zeroed) header.
    __shared__ uint32_t s_mm[2 * G2PC_MAX_CAMERA_BATCH];
    // The tile intervals of both axes, staged in LDS (<= 256 per axis, rect packs 8-bit tile coordinates): interval_range runs two
    // binary searches per axis, ~20 DEPENDENT loads per Gaussian -- from global memory each was a round trip to the L2, and
    // beside the blends of the other cameras (one or two waves of this kernel resident per SIMD, nothing to hide behind) those
    // round trips were the kernel's duration: 148 us per camera under load against 44 alone (round 4).
    __shared__ int32_t s_xs[256], s_ws[256], s_ys[256], s_hs[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) {
        s_xs[k] = k < lay.nx ? lay.xs[k] : 0; s_ws[k] = k < lay.nx ? lay.ws[k] : 0;
        s_ys[k] = k < lay.ny ? lay.ys[k] : 0; s_hs[k] = k < lay.ny ? lay.hs[k] : 0;
    }
    if (threadIdx.x < 2 * G2PC_MAX_CAMERA_BATCH) s_mm[threadIdx.x] = 0u;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (block size: RA_T, or g_head_threads in the camera pipeline)
    // ALL of the Gaussian's inputs are requested at once (17 loads in one round): behind the in-front-of-the-camera test the
    // covariance, opacity and colour were a second round trip, which a wave with one or two neighbours on its SIMD (the rest
    // of the registers belong to another camera's blend) sits out in full
    float x = 0.f, y = 0.f, z = 0.f, S9[9], op_in = 0.f, col_r = 0.f, col_g = 0.f, col_b = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) S9[k] = 0.f;
    if (i < n) {
        x = means3D[3 * i]; y = means3D[3 * i + 1]; z = means3D[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 9; ++k) S9[k] = cov9[9 * i + k];
        op_in = opacity[i];
        col_r = colours[3 * i]; col_g = colours[3 * i + 1]; col_b = colours[3 * i + 2];
    }
    __syncthreads();
    const unsigned c_first = MULTI ? 0u : blockIdx.y, c_end = MULTI ? (unsigned)ncam : blockIdx.y + 1u;
    for (unsigned c = c_first; c < c_end; ++c) {
    uint32_t* depth_key_rev = seg_at(depth_key_rev0, cs, c); uint32_t* index_rev = seg_at(index_rev0, cs, c);
    uint32_t* tiles_touched = seg_at(tiles_touched0, cs, c);
    float4* rec = seg_at(rec0, cs, c); uint32_t* rect = seg_at(rect0, cs, c);
    // Device-resident camera (round 4): read through the constant address space -- the job was written before the launch
    // sequence started and no kernel modifies it -- so the 43 words arrive by SCALAR loads and live in SGPRs.  (Until round 4
    // they were staged through LDS: every matrix element then sat in a VGPR, 50 VGPRs against 36 for the by-value variant.)
    Cam cam_s = cam_val;
    if (CAM_ON_DEVICE) {
        const uint32_t G2PC_CONSTANT* cw =
            (const uint32_t G2PC_CONSTANT*)((const char*)cam_dev + (size_t)c * sizeof(G2pcCameraJob));
        uint32_t* dst = (uint32_t*)&cam_s;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(Cam) / 4); ++k) dst[k] = cw[k];
    }
    const uint8_t* alive = nullptr;          // child pass of a camera (G2pcCameraJob.alive + G2pcTileLayout.tile_parent)
    if (CAM_ON_DEVICE && lay.tile_parent) {
        const G2pcCameraJob* jb = (const G2pcCameraJob*)((const char*)cam_dev + (size_t)c * sizeof(G2pcCameraJob));
        alive = (const uint8_t*)(((unsigned long long)jb->alive_hi << 32) | jb->alive_lo);
    }
    const Cam& cam = cam_s;
    uint32_t key = 0xFFFFFFFFu, touched = 0, rc = 0;
    if (i < n) {
    const float* V = cam.V;
    float pv[4];
    py_view(V, x, y, z, pv);                                         // p_view = [x,1] @ V  (gauss_render.py:163)
    const bool in_mask = pv[2] <= -0.000001f;                       // :167
    if (in_mask) {
        float cv[4];                                                 // cov2d (:101-148), torch's evaluation order: py_project.inl
        py_cov2d(V, pv, cam.lim_x, cam.lim_y, cam.focal_x, cam.focal_y, S9, cv);
        const float c00 = cv[0], c01 = cv[1], c10 = cv[2], c11 = cv[3];
        float ph[4];
        py_hom(cam.P, pv, ph);                                       // projection (:160-163)
        float pw = 1.0f / (ph[3] + 0.000001f);
        float ndx = ph[0] * pw, ndy = ph[1] * pw;
        float mx = ((ndx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;        // gauss_render.py:435-436
        float my = ((ndy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
        float det;
        const float radius = py_radius(cv, det);                     // radius (:171-180) and rect (:182-193)
        float wmax = (float)cam.W - 1.0f, hmax = (float)cam.H - 1.0f;
        float rminx = fminf(fmaxf(mx - radius, 0.0f), wmax), rmaxx = fminf(fmaxf(mx + radius, 0.0f), wmax);
        float rminy = fminf(fmaxf(my - radius, 0.0f), hmax), rmaxy = fminf(fmaxf(my + radius, 0.0f), hmax);
        int ix0, ix1, iy0, iy1;
        interval_range(s_xs, s_ws, lay.nx, rminx, rmaxx, ix0, ix1);
        interval_range(s_ys, s_hs, lay.ny, rminy, rmaxy, iy0, iy1);
        // conic = inverse(cov2d) (:349); exponent pre-scaled for exp2:  w = exp(-0.5 q) = exp2(A dx^2 + C dy^2 + B dx dy)
        float idet = 1.0f / det;
        float k00 = c11 * idet, k11 = c00 * idet, k01 = -c01 * idet, k10 = -c10 * idet;
        const float sc = -0.5f * LOG2E;
        bool ok = (ix1 >= ix0) && (iy1 >= iy0) && (mx == mx) && (my == my) && (det == det);
        if (ok && alive) {
            // child pass (tile_parent + the camera's `alive` bytes): only the children of the nodes the first pass split exist
            // for this camera -- nothing is emitted, sorted or binned for the others (k_duplicate applies the same test)
            uint32_t cnt = 0;
            for (int iy = iy0; iy <= iy1; ++iy)
                for (int ix = ix0; ix <= ix1; ++ix) {
                    cnt += child_exists(lay.tile_parent, alive, iy * lay.nx + ix) ? 1u : 0u;
                }
            ok = cnt > 0;
            touched = cnt;
        } else if (ok) {
            touched = (uint32_t)((ix1 - ix0 + 1) * (iy1 - iy0 + 1));
        }
        if (ok) {
            rc = (uint32_t)ix0 | ((uint32_t)ix1 << 8) | ((uint32_t)iy0 << 16) | ((uint32_t)iy1 << 24);
            key = __float_as_uint(-pv[2]);                          // ascending = nearest first
        }
        // ONE 64-byte record per Gaussian holds what the blend stages from it: a lane gathers one cache line per list
        // entry (plus the live running maximum from best_key) instead of touching three arrays.  The running maximum is
        // deliberately NOT snapshotted here: with four cameras in flight a snapshot is several blends old, the "can
        // this beat the maximum" filter lets many more candidates through and the job takes 48 ms instead of 26.
        // r1.z / r1.w / r2.w serve the blend's chunk-level cull (k_blend_py_pk): on the edge dx = e of a pixel rectangle
        // the exponent A dx^2 + B dx dy + C dy^2 peaks at dy = e * (-B / 2C) (dx = e * (-B / 2A) on an edge dy = e), and a
        // Gaussian whose peak exponent over the rectangle is below cull = -25.5 - log2(opacity) has alpha < 2^-25 on
        // every pixel of it.
        const float qa = sc * k00, qb = sc * (k01 + k10), qc = sc * k11;
        rec[4 * i + 0] = make_float4(mx, my, qa, qb);
        rec[4 * i + 1] = make_float4(qc, op_in, -qb / (2.0f * qc), -qb / (2.0f * qa));
        rec[4 * i + 2] = make_float4(col_r, col_g, col_b, -25.5f - log2f(op_in));
    }
    const long r = n - 1 - i;
    depth_key_rev[r] = key;
    if (index_rev) index_rev[r] = (uint32_t)i;       // nullptr: the sort returns n - 1 - position itself (bucket sort, reversed)
    tiles_touched[i] = touched;
    rect[i] = rc;
    }
    if (mm0) {
        uint32_t a = key != 0xFFFFFFFFu ? ~key : 0u, b = key != 0xFFFFFFFFu ? key : 0u;
        a = wave_max_u32(a); b = wave_max_u32(b);
        const unsigned cl = MULTI ? c : 0u;
        if ((threadIdx.x & 63) == 0) { atomicMax(&s_mm[2 * cl], a); atomicMax(&s_mm[2 * cl + 1], b); }
    }
    }
    if (mm0) {
        __syncthreads();
        const unsigned nslots = 2u * (MULTI ? (unsigned)ncam : 1u);
        if (threadIdx.x < nslots) {
            const unsigned cl = threadIdx.x >> 1, c = MULTI ? cl : blockIdx.y;
            BucketHdr* mm = seg_at(mm0, cs, c);
            const uint32_t slot = blockIdx.x % mm_slots;
            atomicMax(&mm->partial[2 * slot + (threadIdx.x & 1u)], s_mm[threadIdx.x]);
        }
    }
}

__global__ __launch_bounds__(RA_T) void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                    long n, uint32_t* __restrict__ dst) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// K3: one (tile, gaussian) instance per overlapped tile, emitted in depth order (rasterizer_impl.cu:69-110)
template <bool WIDE>      // WIDE: 16-bit tile coordinates, two words per Gaussian (native-semantics images beyond 4 096 pixels)
__global__ __launch_bounds__(RA_T) void k_duplicate(const uint32_t* __restrict__ sorted_idx,
                                                   const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ rect, long n, int nx,
                                                   uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_g,
                                                   const uint32_t* __restrict__ l_eff, int gshift, size_t cs,
                                                   const int32_t* __restrict__ tile_parent,
                                                   const G2pcCameraJob* __restrict__ jobs) {
    // gshift > 0 (inst_g unused): ONE word per instance, tile << gshift | Gaussian -- the tile sort then moves keys only
    // tile_parent + jobs: a camera's child pass -- only the children of split nodes take instances (k_preprocess_py counted so)
    const uint8_t* alive = nullptr;
    if (tile_parent && jobs) {
        const G2pcCameraJob* jb = jobs + blockIdx.y;
        alive = (const uint8_t*)(((unsigned long long)jb->alive_hi << 32) | jb->alive_lo);
    }
    sorted_idx = seg(sorted_idx, cs); offsets = seg(offsets, cs); rect = seg(rect, cs); inst_tile = seg(inst_tile, cs);
    inst_g = seg(inst_g, cs); l_eff = seg(l_eff, cs);
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (l_eff && *l_eff == 0u) return;          // capacity-sized launch: nothing to emit (or more than fits)
    uint32_t off = offsets[p], end = offsets[p + 1];
    if (end == off) return;
    uint32_t g = sorted_idx[p];
    int ix0, ix1, iy0, iy1;
    if (WIDE) {
        const uint32_t rx = rect[2 * (size_t)g], ry = rect[2 * (size_t)g + 1];
        ix0 = rx & 0xFFFF; ix1 = rx >> 16; iy0 = ry & 0xFFFF; iy1 = ry >> 16;
    } else {
        const uint32_t rc = rect[g];
        ix0 = rc & 255; ix1 = (rc >> 8) & 255; iy0 = (rc >> 16) & 255; iy1 = rc >> 24;
    }
    for (int iy = iy0; iy <= iy1; ++iy)
        for (int ix = ix0; ix <= ix1; ++ix) {
            if (alive && !child_exists(tile_parent, alive, iy * nx + ix)) continue;
            if (gshift) {
                inst_tile[off] = ((uint32_t)(iy * nx + ix) << gshift) | g;
            } else {
                inst_tile[off] = (uint32_t)(iy * nx + ix);
                inst_g[off] = g;
            }
            ++off;
        }
}

// ---------------------------------------------------------------------------------------------------------
// K6 (PY): blend.  One single-wave block per (tile, 256-pixel chunk); 64 lanes x 4 pixels; Gaussians staged through
// wave-private LDS in batches of 64 (no s_barrier: 32 independent waves per CU, fine-grained early exit).  For every Gaussian the wave reduces (max T*alpha, lowest pixel among the maxima) and
// lane 0 publishes  key = contribution_bits << 32 | ~(slot << 24 | tile_seq << 12 | pixel)  with one 64-bit
// atomicMax -- but only when some lane can beat the value staged from the running maximum.
// ---------------------------------------------------------------------------------------------------------
constexpr int BL_T = 64, BL_BATCH = 64;

// Chunk-level cull of the PY blend: can this Gaussian's alpha reach 2^-25 anywhere on the pixel rectangle [rx0, rx1] x
// [ry0, ry1]?  The exponent A dx^2 + B dx dy + C dy^2 (r0.z, r0.w, r1.x) peaks at 0 if the centre (r0.x, r0.y) is inside
// the rectangle, else on the edge(s) facing the centre, where it is maximised in closed form (r1.z = -B / 2C, r1.w =
// -B / 2A, from k_preprocess_py); cth = -25.5 - log2(opacity).  Below 2^-25, T * (1 - alpha) == T bit for bit in fp32 (here
// and in the reference's cumprod) and the colour / contribution terms are < 3e-8: the visit is dropped and the survivors
// of a batch are compacted in depth order.  On the bench scene that is ~40 % of all visits.  NaNs compare false: kept.
__device__ __forceinline__ bool rect_may_touch(float mx, float my, float A, float B, float C, float slope_y, float slope_x,
                                               float cth, float rx0, float rx1, float ry0, float ry1) {
    const float ax = rx0 - mx, bx = rx1 - mx, ay = ry0 - my, by = ry1 - my;
    const bool xout = ax > 0.f || bx < 0.f, yout = ay > 0.f || by < 0.f;
    const float ex = ax > 0.f ? ax : bx, ey = ay > 0.f ? ay : by;
    const float dyc = fminf(fmaxf(ex * slope_y, ay), by), dxc = fminf(fmaxf(ey * slope_x, ax), bx);
    const float vx = ex * (A * ex + B * dyc) + (C * dyc) * dyc;
    const float vy = ey * (C * ey + B * dxc) + (A * dxc) * dxc;
    float peak = 0.0f;
    if (xout) peak = vx;
    if (yout) peak = xout ? fmaxf(vx, vy) : vy;
    return !(peak < cth);
}
__device__ __forceinline__ bool chunk_may_touch(const float4& r0, const float4& r1, float cth, float rx0, float rx1,
                                                float ry0, float ry1) {
    return rect_may_touch(r0.x, r0.y, r0.z, r0.w, r1.x, r1.z, r1.w, cth, rx0, rx1, ry0, ry1);
}

// Pixels of a tile are grouped in 8x8 sub-blocks (row-major inside the tile); a chunk = PPT consecutive sub-blocks,
// lane l owns pixel (l % 8, l / 8) of each of them.  Compact blocks saturate together (early exit) and the PPT
// template trades instruction count per (pixel, Gaussian) pair against the length of the serial chain a single
// wave has to walk through a tile's list (the launch's critical path).
template <int PPT, int U>
__global__ __launch_bounds__(BL_T) void k_blend_py(Layout lay, const int32_t* __restrict__ chunk_tile,
                                                  const int32_t* __restrict__ chunk_pix0,
                                                  const uint2* __restrict__ tile_range,
                                                  const uint32_t* __restrict__ inst_g, uint32_t gmask,
                                                  const float4* __restrict__ rec,
                                                  unsigned long long* __restrict__ best_key, uint32_t order_base,
                                                  float t_floor, float bg, float* __restrict__ tilebuf,
                                                  uint32_t* __restrict__ chunk_work,
                                                  const G2pcCameraJob* __restrict__ job, size_t cs) {
    // one wave64 per block: the LDS stage is wave-private, no s_barrier anywhere
    const unsigned chunk_i = blockIdx.z * gridDim.y + blockIdx.y;       // the blends: camera = blockIdx.x, chunk in (y, z)
    if ((int)chunk_i >= lay.num_chunks) return;
    tile_range = seg_at(tile_range, cs, blockIdx.x); inst_g = seg_at(inst_g, cs, blockIdx.x); rec = seg_at(rec, cs, blockIdx.x);
    if (job) {
        job += blockIdx.x;
        order_base = job->camera_slot << (12 + lay.seq_bits); t_floor = job->t_floor; bg = job->cam.bg[0];
        const unsigned long long tb = ((unsigned long long)job->tilebuf_hi << 32) | job->tilebuf_lo;
        if (tb) tilebuf = (float*)tb;              // one colour buffer per camera (deferred colour resolve)
    }   // see k_preprocess_py
    __shared__ float4 s_p0[BL_BATCH + 4];
    __shared__ float4 s_p1[BL_BATCH + 4];
    __shared__ float4 s_p2[BL_BATCH + 4];
    __shared__ uint32_t s_g[BL_BATCH];
    const int tile = chunk_tile[chunk_i];
    const int sb0 = chunk_pix0[chunk_i];                 // first 8x8 sub-block of this chunk
    const int ix = tile % lay.nx, iy = tile / lay.nx;
    const int x0 = lay.xs[ix], w = lay.ws[ix], y0 = lay.ys[iy], h = lay.hs[iy];
    const int nsbx = (w + 7) >> 3;
    const uint32_t order_tile = order_base | ((uint32_t)lay.tile_seq[tile] << 12);
    const unsigned lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;

    int pix[PPT];
    float px[PPT], py[PPT], T[PPT], cr[PPT], cg[PPT], cb[PPT];
    int bx0 = 1 << 30, bx1 = -1, by0 = 1 << 30, by1 = -1;          // pixel bounds of the chunk inside the tile (uniform)
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int sb = sb0 + j;
        const int sx = (sb % nsbx) * 8, sy = (sb / nsbx) * 8;
        int x = sx + lx, y = sy + ly;
        bool valid = (x < w) && (y < h);
        if (sy < h) {
            bx0 = sx < bx0 ? sx : bx0; by0 = sy < by0 ? sy : by0;
            bx1 = sx + 7 > bx1 ? sx + 7 : bx1; by1 = sy + 7 > by1 ? sy + 7 : by1;
        }
        pix[j] = valid ? y * w + x : -1;        // row-major pixel index inside the tile (the reference's arg-max order)
        px[j] = (float)(x0 + x);
        py[j] = (float)(y0 + y);
        T[j] = valid ? 1.0f : 0.0f;             // invalid slots never contribute (contribution = T * alpha = 0)
        cr[j] = cg[j] = cb[j] = 0.0f;
    }
    bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
    const float rx0 = (float)(x0 + bx0), rx1 = (float)(x0 + bx1), ry0 = (float)(y0 + by0), ry1 = (float)(y0 + by1);
    const bool cull = t_floor > 0.0f;        // t_floor = 0 is the to-the-letter mode: nothing is skipped
    const uint2 se = tile_range[tile];        // k_tile_gate: [first, end) of the tile's instances, empty for a gated tile
    const uint32_t start = se.x, end = se.y;
    // Software pipeline of the list staging (the gathers are two dependent HBM/L2 round trips and sit on the
    // critical path of the waves that never saturate): ids run two batches ahead, parameters one batch ahead.
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t g_cur = 0, g_nxt = 0;
    bool v_cur = (start + lane) < end, v_nxt = (start + BL_BATCH + lane) < end;
    if (v_cur) g_cur = inst_g[start + lane] & gmask;
    if (v_nxt) g_nxt = inst_g[start + BL_BATCH + lane] & gmask;
    // raw loads only (no arithmetic on them before the LDS write, or the compiler waits for the load right here)
    float4 r0 = zero4, r1 = zero4;                       // zero opacity = padding
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, cth = 0.f;
    uint32_t gmb = 0x7F000000u;                          // huge running maximum: padding is never a candidate
    if (v_cur) {
        r0 = rec[4 * (size_t)g_cur];
        r1 = rec[4 * (size_t)g_cur + 1];
        const float4 r2 = rec[4 * (size_t)g_cur + 2];
        c0 = r2.x; c1 = r2.y; c2 = r2.z; cth = r2.w; gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
    }
    uint32_t processed = 0;
    for (uint32_t b = start; b < end; b += BL_BATCH) {
        processed = b + BL_BATCH - start;
        // waves still walking after many batches are the launch's critical path (most chunks saturate within
        // ~700 entries): let them win issue arbitration over the short-lived waves sharing their SIMD
        if (processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(2);
        wave_sync();                            // everyone is done reading the previous batch
        // chunk-level cull (chunk_may_touch) + compaction of the survivors in depth order
        const bool keep = v_cur && (!cull || chunk_may_touch(r0, r1, cth, rx0, rx1, ry0, ry1));
        const unsigned long long kept = __ballot(keep ? 1 : 0);
        const int cnt = __popcll(kept);
        if (keep) {
            const int pos = __popcll(kept & ((1ull << lane) - 1ull));
            s_p0[pos] = r0;
            s_p1[pos] = r1;
            s_p2[pos] = make_float4(c0, c1, c2, fmaxf(__uint_as_float(gmb), 1.17549435e-38f));   // a 0 contribution never updates
            s_g[pos] = g_cur;
        }
        if (lane < (unsigned)U) {                          // the last trip reads up to U - 1 entries past cnt: neutral ones
            s_p0[cnt + lane] = zero4;
            s_p1[cnt + lane] = zero4;
            s_p2[cnt + lane] = make_float4(0.f, 0.f, 0.f, 1.17549435e-38f);
        }
        // issue the loads of batch b+1 (parameters) and b+2 (ids); they complete under the blend of batch b
        g_cur = g_nxt;
        v_cur = v_nxt;
        v_nxt = (b + 2 * BL_BATCH + lane) < end;
        g_nxt = 0;
        if (v_nxt) g_nxt = inst_g[b + 2 * BL_BATCH + lane] & gmask;
        r0 = zero4; r1 = zero4; c0 = c1 = c2 = 0.f; cth = 0.f; gmb = 0x7F000000u;
        if (v_cur) {
            r0 = rec[4 * (size_t)g_cur];
            r1 = rec[4 * (size_t)g_cur + 1];
            const float4 r2 = rec[4 * (size_t)g_cur + 2];
            c0 = r2.x; c1 = r2.y; c2 = r2.z; cth = r2.w; gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
        }
        wave_sync();
        // U Gaussians per trip: their weights (position only) are independent -> U exp chains in flight; the
        // transmittance recurrence and the visibility bookkeeping then run in depth order.
        for (int k0 = 0; k0 < cnt; k0 += U) {
            float alpha[U][PPT];
            float4 cc[U];                                               // colour + staged maximum, read with the rest so
#pragma unroll                                                          // the serial part below never waits on the LDS
            for (int u = 0; u < U; ++u) cc[u] = s_p2[k0 + u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 a = s_p0[k0 + u], q = s_p1[k0 + u];       // entries past cnt are zero-opacity padding
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    float dx = px[j] - a.x, dy = py[j] - a.y;
                    // A dx^2 + B dx dy + C dy^2 = dx (A dx + B dy) + (C dy) dy : 5 VALU
                    float power = fmaf(dx, fmaf(a.w, dy, a.z * dx), (q.x * dy) * dy);
                    float wgt = __builtin_amdgcn_exp2f(power);      // raw v_exp_f32 (results below 2^-126 flush to 0)
                    alpha[u][j] = fminf(wgt * q.y, 0.99f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < PPT; ++j) G2PC_PIN(alpha[u][j]);
                G2PC_PIN(cc[u].x); G2PC_PIN(cc[u].y); G2PC_PIN(cc[u].z); G2PC_PIN(cc[u].w);
            }
            // serial part: transmittance recurrence for the U Gaussians, branch-free; the (rare, after the first few
            // cameras) visibility updates are handled behind ONE wave-uniform test per trip
            float bestv[U];
            uint32_t bestp[U];
            bool any_cand = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 c = cc[u];
                float best = 0.0f;
                uint32_t bestpix = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    float contrib = T[j] * alpha[u][j];
                    cr[j] = fmaf(contrib, c.x, cr[j]);
                    cg[j] = fmaf(contrib, c.y, cg[j]);
                    cb[j] = fmaf(contrib, c.z, cb[j]);
                    T[j] -= contrib;
                    if (PPT == 1) { best = contrib; bestpix = (uint32_t)pix[j]; }
                    // sub-blocks of one lane are not ordered by pixel index: explicit tie-break to the lowest index
                    else if (contrib > best || (contrib == best && contrib > 0.0f && (uint32_t)pix[j] < bestpix)) { best = contrib; bestpix = (uint32_t)pix[j]; }
                }
                bestv[u] = best;
                bestp[u] = bestpix;
                any_cand = any_cand || (best >= c.w);                  // c.w = max(running maximum, FLT_MIN)
            }
            if (__any(any_cand ? 1 : 0)) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (__any(bestv[u] >= cc[u].w)) {
                        uint32_t bits = __float_as_uint(bestv[u]);
                        uint32_t m = wave_max_u32_dpp(bits);
                        uint32_t pm;
                        if (PPT == 1) {     // pixel index grows with the lane: the lowest lane among the maxima owns it
                            const unsigned long long at_max = __ballot(bits == m);
                            pm = (uint32_t)__builtin_amdgcn_readlane((int)bestp[u], __ffsll(at_max) - 1);
                        } else {
                            pm = wave_min_u32_dpp(bits == m ? bestp[u] : 0xFFFFFFFFu);
                        }
                        if (lane == 0) {
                            unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~(order_tile | pm));
                            atomicMax(&best_key[s_g[k0 + u]], key);
                        }
                    }
                }
            }
        }
        {   // chunk-level early exit.  t_floor = 0: only once every transmittance has underflowed to exactly 0.0f -- all
            // later contributions and colour terms are then exactly 0 in fp32 (as in the reference's cumprod), so this
            // is bit-exact; t_floor > 0: everything still to come is below t_floor.
            bool done = true;
#pragma unroll
            for (int j = 0; j < PPT; ++j) done = done && (T[j] <= t_floor);
            if (__all(done ? 1 : 0)) break;
        }
    }
    if (chunk_work && lane == 0) {                 // diagnostics: list length and how far this wave walked it
        chunk_work[8 * chunk_i] = end - start;
        chunk_work[8 * chunk_i + 1] = processed;
    }
    float* out = tilebuf + 3 * (size_t)lay.tile_pix_off[tile];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        if (pix[j] >= 0) {
            out[3 * (size_t)pix[j] + 0] = fmaf(T[j], bg, cr[j]);
            out[3 * (size_t)pix[j] + 1] = fmaf(T[j], bg, cg[j]);
            out[3 * (size_t)pix[j] + 2] = fmaf(T[j], bg, cb[j]);
        }
    }
}

// K6 (PY), packed variant: a chunk = 2 consecutive 8x8 sub-blocks, lane l owns pixel (l % 8, l / 8) of both, and the two
// pixels travel as one packed f32 pair through v_pk_{add,mul,fma}_f32 -- per Gaussian and lane 7 issue slots for the
// two quadratic forms instead of 14, 1+1 for the transmittance recurrence instead of 4, 3 for the colours instead
// of 6 (v_exp_f32 / v_min_f32 have no packed form).  Arithmetic per element is the scalar kernel's, bit for bit.
// Sub-blocks are numbered row-major inside the tile, so pixel 0 of a lane always has the lower in-tile index: ties
// between the two go to pixel 0, as the reference's arg-max does.
template <int U>
__global__ __launch_bounds__(BL_T) void k_blend_py_pk(Layout lay, const int32_t* __restrict__ chunk_tile,
                                                     const int32_t* __restrict__ chunk_pix0,
                                                     const uint2* __restrict__ tile_range,
                                                     const uint32_t* __restrict__ inst_g, uint32_t gmask,
                                                     const float4* __restrict__ rec,
                                                     unsigned long long* __restrict__ best_key, uint32_t order_base,
                                                     float t_floor, float bg, float* __restrict__ tilebuf,
                                                     uint32_t* __restrict__ chunk_work,
                                                     const G2pcCameraJob* __restrict__ job, size_t cs) {
    const unsigned chunk_i = blockIdx.z * gridDim.y + blockIdx.y;       // the blends: camera = blockIdx.x, chunk in (y, z)
    if ((int)chunk_i >= lay.num_chunks) return;
    tile_range = seg_at(tile_range, cs, blockIdx.x); inst_g = seg_at(inst_g, cs, blockIdx.x); rec = seg_at(rec, cs, blockIdx.x);
    if (job) {
        job += blockIdx.x;
        order_base = job->camera_slot << (12 + lay.seq_bits); t_floor = job->t_floor; bg = job->cam.bg[0];
        const unsigned long long tb = ((unsigned long long)job->tilebuf_hi << 32) | job->tilebuf_lo;
        if (tb) tilebuf = (float*)tb;              // one colour buffer per camera (deferred colour resolve)
    }
    __shared__ float4 s_p0[BL_BATCH + 4];
    __shared__ float4 s_p1[BL_BATCH + 4];
    __shared__ float4 s_p2[BL_BATCH + 4];
    __shared__ uint32_t s_g[BL_BATCH];
    const int tile = chunk_tile[chunk_i];
    const uint32_t sbpair = (uint32_t)chunk_pix0[chunk_i];    // two ADJACENT 8x8 sub-blocks: a | b << 16, b = 0xFFFF: none
    const int ix = tile % lay.nx, iy = tile / lay.nx;
    const int x0 = lay.xs[ix], w = lay.ws[ix], y0 = lay.ys[iy], h = lay.hs[iy];
    const int nsbx = (w + 7) >> 3;
    const uint32_t order_tile = order_base | ((uint32_t)lay.tile_seq[tile] << 12);
    const unsigned lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;

    int pix[2];
    float pxs[2], pys[2], Ts[2];
    int bx0 = 1 << 30, bx1 = -1, by0 = 1 << 30, by1 = -1;          // pixel bounds of the chunk inside the tile (uniform)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sb = (int)((sbpair >> (16 * j)) & 0xFFFFu);
        const bool present = sb != 0xFFFF;
        const int sx = present ? (sb % nsbx) * 8 : 0, sy = present ? (sb / nsbx) * 8 : 0;
        int x = sx + lx, y = sy + ly;
        bool valid = present && (x < w) && (y < h);
        pix[j] = valid ? y * w + x : -1;
        pxs[j] = (float)(x0 + x);
        pys[j] = (float)(y0 + y);
        Ts[j] = valid ? 1.0f : 0.0f;
        if (present) {
            bx0 = sx < bx0 ? sx : bx0; by0 = sy < by0 ? sy : by0;
            bx1 = sx + 7 > bx1 ? sx + 7 : bx1; by1 = sy + 7 > by1 ? sy + 7 : by1;
        }
    }
    bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
    const float rx0 = (float)(x0 + bx0), rx1 = (float)(x0 + bx1), ry0 = (float)(y0 + by0), ry1 = (float)(y0 + by1);
    const bool cull = t_floor > 0.0f;        // t_floor = 0 is the to-the-letter mode: nothing is skipped
    const pk2 px = pk_make(pxs[0], pxs[1]), py = pk_make(pys[0], pys[1]);
    pk2 T = pk_make(Ts[0], Ts[1]);
    pk2 cr = pk_splat(0.f), cg = pk_splat(0.f), cb = pk_splat(0.f);

    const uint2 se = tile_range[tile];        // k_tile_gate: [first, end) of the tile's instances, empty for a gated tile
    const uint32_t start = se.x, end = se.y;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t g_cur = 0, g_nxt = 0;
    bool v_cur = (start + lane) < end, v_nxt = (start + BL_BATCH + lane) < end;
    if (v_cur) g_cur = inst_g[start + lane] & gmask;
    if (v_nxt) g_nxt = inst_g[start + BL_BATCH + lane] & gmask;
    float4 r0 = zero4, r1 = zero4;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, cth = 0.f;
    uint32_t gmb = 0x7F000000u;
    if (v_cur) {
        r0 = rec[4 * (size_t)g_cur];
        r1 = rec[4 * (size_t)g_cur + 1];
        const float4 r2 = rec[4 * (size_t)g_cur + 2];
        c0 = r2.x; c1 = r2.y; c2 = r2.z; cth = r2.w; gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
    }
    uint32_t processed = 0;
    for (uint32_t b = start; b < end; b += BL_BATCH) {
        processed = b + BL_BATCH - start;
        if (processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(2);
        wave_sync();
        // chunk-level cull (chunk_may_touch) + compaction of the survivors in depth order
        const bool keep = v_cur && (!cull || chunk_may_touch(r0, r1, cth, rx0, rx1, ry0, ry1));
        const unsigned long long kept = __ballot(keep ? 1 : 0);
        const int cnt = __popcll(kept);
        if (keep) {
            const int pos = __popcll(kept & ((1ull << lane) - 1ull));
            s_p0[pos] = r0;
            s_p1[pos] = r1;
            s_p2[pos] = make_float4(c0, c1, c2, fmaxf(__uint_as_float(gmb), 1.17549435e-38f));
            s_g[pos] = g_cur;
        }
        if (lane < (unsigned)U) {                          // the last trip reads up to U - 1 entries past cnt: neutral ones
            s_p0[cnt + lane] = zero4;
            s_p1[cnt + lane] = zero4;
            s_p2[cnt + lane] = make_float4(0.f, 0.f, 0.f, 1.17549435e-38f);
        }
        g_cur = g_nxt;
        v_cur = v_nxt;
        v_nxt = (b + 2 * BL_BATCH + lane) < end;
        g_nxt = 0;
        if (v_nxt) g_nxt = inst_g[b + 2 * BL_BATCH + lane] & gmask;
        r0 = zero4; r1 = zero4; c0 = c1 = c2 = 0.f; cth = 0.f; gmb = 0x7F000000u;
        if (v_cur) {
            r0 = rec[4 * (size_t)g_cur];
            r1 = rec[4 * (size_t)g_cur + 1];
            const float4 r2 = rec[4 * (size_t)g_cur + 2];
            c0 = r2.x; c1 = r2.y; c2 = r2.z; cth = r2.w; gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
        }
        wave_sync();
        for (int k0 = 0; k0 < cnt; k0 += U) {
            pk2 alpha[U];
            float4 cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cc[u] = s_p2[k0 + u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 a = s_p0[k0 + u], q = s_p1[k0 + u];
                const pk2 dx = px - a.x, dy = py - a.y;
                const pk2 power = pk_fma(dx, pk_fma(pk_splat(a.w), dy, a.z * dx), (q.x * dy) * dy);
                const pk2 wgt = pk_make(__builtin_amdgcn_exp2f(power[0]), __builtin_amdgcn_exp2f(power[1]));
                const pk2 al = wgt * q.y;
                alpha[u] = pk_make(fminf(al[0], 0.99f), fminf(al[1], 0.99f));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                G2PC_PIN(alpha[u]);
                G2PC_PIN(cc[u].x); G2PC_PIN(cc[u].y); G2PC_PIN(cc[u].z); G2PC_PIN(cc[u].w);
            }
            pk2 contrib[U];
            bool any_cand = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 c = cc[u];
                contrib[u] = T * alpha[u];
                cr = pk_fma(contrib[u], pk_splat(c.x), cr);
                cg = pk_fma(contrib[u], pk_splat(c.y), cg);
                cb = pk_fma(contrib[u], pk_splat(c.z), cb);
                T = T - contrib[u];
                any_cand = any_cand || (fmaxf(contrib[u][0], contrib[u][1]) >= c.w);
            }
            if (__any(any_cand ? 1 : 0)) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float best = fmaxf(contrib[u][0], contrib[u][1]);
                    if (__any(best >= cc[u].w)) {
                        uint32_t bits = __float_as_uint(best);
                        uint32_t m = wave_max_u32_dpp(bits);
                        // lowest pixel among the maxima: inside each sub-block the pixel index grows with the lane, so
                        // per sub-block it is the lowest lane at the maximum (two ballots instead of a second reduction)
                        const unsigned long long at0 = __ballot(__float_as_uint(contrib[u][0]) == m);
                        const unsigned long long at1 = __ballot(__float_as_uint(contrib[u][1]) == m);
                        const uint32_t pa = at0 ? (uint32_t)__builtin_amdgcn_readlane(pix[0], __ffsll(at0) - 1) : 0xFFFFFFFFu;
                        const uint32_t pb = at1 ? (uint32_t)__builtin_amdgcn_readlane(pix[1], __ffsll(at1) - 1) : 0xFFFFFFFFu;
                        const uint32_t pm = pa < pb ? pa : pb;
                        if (lane == 0) {
                            unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~(order_tile | pm));
                            atomicMax(&best_key[s_g[k0 + u]], key);
                        }
                    }
                }
            }
        }
        if (__all((T[0] <= t_floor && T[1] <= t_floor) ? 1 : 0)) break;      // see k_blend_py
    }
    if (chunk_work && lane == 0) {
        chunk_work[8 * chunk_i] = end - start;
        chunk_work[8 * chunk_i + 1] = processed;
    }
    float* out = tilebuf + 3 * (size_t)lay.tile_pix_off[tile];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (pix[j] >= 0) {
            out[3 * (size_t)pix[j] + 0] = fmaf(T[j], bg, cr[j]);
            out[3 * (size_t)pix[j] + 1] = fmaf(T[j], bg, cg[j]);
            out[3 * (size_t)pix[j] + 2] = fmaf(T[j], bg, cb[j]);
        }
    }
}

// K6 (PY), dual-list variant: the wave owns the same two adjacent 8x8 sub-blocks as k_blend_py_pk (lane l = pixel
// (l % 8, l / 8) of each), loads and tests every list entry ONCE, but keeps one compacted list PER SUB-BLOCK: a Gaussian
// is blended into a sub-block only if its alpha can reach 2^-25 on that 8x8 square (chunk_may_touch), and a sub-block
// that has saturated stops taking entries while the other goes on.  Against the 16x8 granularity of the packed kernel
// that is ~25 % fewer (pixel, Gaussian) pairs on the bench scene for the same loads and tests.
// The exponent is evaluated in expanded form about the CENTRE of the sub-block: with u, v in {-3.5 .. 3.5} (lane
// constants, the same for both sub-blocks) and, per (Gaussian, sub-block), m = mean - centre,
//     A (u-mx)^2 + B (u-mx)(v-my) + C (v-my)^2 + log2(opacity)  =  u (A u + B v + Lu) + v (C v + Lv) + K,
//     Lu = -(2 A mx + B my),  Lv = -(2 C my + B mx),  K = (A mx + B my) mx + C my^2 + log2(opacity)
// -- five FMAs per pixel instead of seven operations, and the opacity multiply rides in K.  Rounding differs from the
// reference's order of operations by ~eps * (|exponent| + |A| 50): <= 2e-5 relative in alpha for the sharpest Gaussians
// the 0.3-pixel dilation admits, ~3e-6 typically (the reference's own dx = pixel - mean carries eps * |mean| already).
#ifndef G2PC_BLEND_VGPRS
#define G2PC_BLEND_VGPRS 0
#endif
#if G2PC_BLEND_VGPRS
#define G2PC_BLEND_ATTR __attribute__((amdgpu_num_vgpr(G2PC_BLEND_VGPRS)))
#else
#define G2PC_BLEND_ATTR
#endif
template <int U>
__global__ __launch_bounds__(BL_T) G2PC_BLEND_ATTR void k_blend_py_dl(Layout lay, const int32_t* __restrict__ chunk_tile,
                                                     const int32_t* __restrict__ chunk_pix0,
                                                     const uint2* __restrict__ tile_range,
                                                     const uint32_t* __restrict__ inst_g, uint32_t gmask,
                                                     const float4* __restrict__ rec,
                                                     unsigned long long* __restrict__ best_key, uint32_t order_base,
                                                     float t_floor, float bg, float* __restrict__ tilebuf,
                                                     uint32_t* __restrict__ chunk_work,
                                                     const G2pcCameraJob* __restrict__ job, size_t cs) {
    const unsigned chunk_i = blockIdx.z * gridDim.y + blockIdx.y;       // the blends: camera = blockIdx.x, chunk in (y, z)
    if ((int)chunk_i >= lay.num_chunks) return;
    tile_range = seg_at(tile_range, cs, blockIdx.x); inst_g = seg_at(inst_g, cs, blockIdx.x); rec = seg_at(rec, cs, blockIdx.x);
    if (job) {
        job += blockIdx.x;
        order_base = job->camera_slot << (12 + lay.seq_bits); t_floor = job->t_floor; bg = job->cam.bg[0];
        const unsigned long long tb = ((unsigned long long)job->tilebuf_hi << 32) | job->tilebuf_lo;
        if (tb) tilebuf = (float*)tb;              // one colour buffer per camera (deferred colour resolve)
    }
    const unsigned long long clk0 = chunk_work ? wall_clock64() : 0ull;     // diagnostics only
    __shared__ float4 s_a[2][BL_BATCH + 4];         // A, B, C, Lu
    __shared__ float4 s_b[2][BL_BATCH + 4];         // Lv, K, red, green
    __shared__ float2 s_c[2][BL_BATCH + 4];         // blue, max(running maximum, FLT_MIN)
    __shared__ uint32_t s_g[2][BL_BATCH];
    const int tile = chunk_tile[chunk_i];
    const uint32_t sbpair = (uint32_t)chunk_pix0[chunk_i];    // a | b << 16, b = 0xFFFF: none
    const int ix = tile % lay.nx, iy = tile / lay.nx;
    const int x0 = lay.xs[ix], w = lay.ws[ix], y0 = lay.ys[iy], h = lay.hs[iy];
    const int nsbx = (w + 7) >> 3;
    const uint32_t order_tile = order_base | ((uint32_t)lay.tile_seq[tile] << 12);
    const unsigned lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const float uu = (float)lx - 3.5f, vv = (float)ly - 3.5f;

    int pix[2];
    float T[2], cr[2], cg[2], cb[2], ox[2], oy[2], rx1[2], ry1[2];
    bool done[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sb = (int)((sbpair >> (16 * j)) & 0xFFFFu);
        const bool present = sb != 0xFFFF;
        const int sx = present ? (sb % nsbx) * 8 : 0, sy = present ? (sb / nsbx) * 8 : 0;
        const int x = sx + lx, y = sy + ly;
        const bool valid = present && (x < w) && (y < h);
        pix[j] = valid ? y * w + x : -1;
        T[j] = valid ? 1.0f : 0.0f;
        cr[j] = cg[j] = cb[j] = 0.0f;
        ox[j] = (float)(x0 + sx) + 3.5f;
        oy[j] = (float)(y0 + sy) + 3.5f;
        rx1[j] = (float)(x0 + (sx + 7 > w - 1 ? w - 1 : sx + 7));   // the cull rectangle stops at the tile's edge
        ry1[j] = (float)(y0 + (sy + 7 > h - 1 ? h - 1 : sy + 7));
        done[j] = !present;
    }
    const bool cull = t_floor > 0.0f;        // t_floor = 0 is the to-the-letter mode: nothing is skipped

    const uint2 se = tile_range[tile];        // k_tile_gate: [first, end) of the tile's instances, empty for a gated tile
    const uint32_t start = se.x, end = se.y;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t g_cur = 0, g_nxt = 0;
    bool v_cur = (start + lane) < end, v_nxt = (start + BL_BATCH + lane) < end;
    if (v_cur) g_cur = inst_g[start + lane] & gmask;
    if (v_nxt) g_nxt = inst_g[start + BL_BATCH + lane] & gmask;
    float4 r0 = zero4, r1 = zero4, r2 = zero4;
    uint32_t gmb = 0x7F000000u;
    if (v_cur) {
        r0 = rec[4 * (size_t)g_cur];
        r1 = rec[4 * (size_t)g_cur + 1];
        r2 = rec[4 * (size_t)g_cur + 2];
        gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
    }
    uint32_t processed = 0, visits = 0;
    for (uint32_t b = start; b < end; b += BL_BATCH) {
        processed = b + BL_BATCH - start;
        if (processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(2);
        wave_sync();
        int cnt[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (done[j]) continue;                          // wave-uniform
            const bool keep = v_cur && (!cull || chunk_may_touch(r0, r1, r2.w, ox[j] - 3.5f, rx1[j], oy[j] - 3.5f, ry1[j]));
            const unsigned long long kept = __ballot(keep ? 1 : 0);
            cnt[j] = __popcll(kept);
            if (keep) {
                const int pos = __popcll(kept & ((1ull << lane) - 1ull));
                const float mx = r0.x - ox[j], my = r0.y - oy[j];
                const float A = r0.z, B = r0.w, C = r1.x;
                const float h1 = fmaf(A, mx, B * my);                                     // A mx + B my
                const float Lu = -(fmaf(A, mx, h1)), Lv = -(fmaf(2.0f * C, my, B * mx));
                const float K = fmaf(h1, mx, fmaf(C * my, my, -25.5f - r2.w));            // ... + log2(opacity)
                s_a[j][pos] = make_float4(A, B, C, Lu);
                s_b[j][pos] = make_float4(Lv, K, r2.x, r2.y);
                s_c[j][pos] = make_float2(r2.z, fmaxf(__uint_as_float(gmb), 1.17549435e-38f));
                s_g[j][pos] = g_cur;
            }
            if (lane < (unsigned)U) {                       // the last trip reads up to U - 1 entries past cnt: alpha = 0 ones
                s_a[j][cnt[j] + lane] = zero4;
                s_b[j][cnt[j] + lane] = make_float4(0.f, -INFINITY, 0.f, 0.f);
                s_c[j][cnt[j] + lane] = make_float2(0.f, 1.17549435e-38f);
            }
        }
        g_cur = g_nxt;
        v_cur = v_nxt;
        v_nxt = (b + 2 * BL_BATCH + lane) < end;
        g_nxt = 0;
        if (v_nxt) g_nxt = inst_g[b + 2 * BL_BATCH + lane] & gmask;
        r0 = zero4; r1 = zero4; r2 = zero4; gmb = 0x7F000000u;
        if (v_cur) {
            r0 = rec[4 * (size_t)g_cur];
            r1 = rec[4 * (size_t)g_cur + 1];
            r2 = rec[4 * (size_t)g_cur + 2];
            gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
        }
        wave_sync();
        // One trip = U entries of a list: weights (independent exp chains), then the transmittance recurrence in depth
        // order, then -- rarely, behind one wave-uniform test -- the visibility bookkeeping.  The two lists are walked IN
        // STEP while both have entries (two independent recurrences in one instruction stream: a lone wave, which is
        // what the tail of every launch consists of, is latency-bound), the longer one finishes alone.
        auto weights = [&](auto nn, int j, int k0, float* alpha, float4* qb, float2* qc) {
            constexpr int N = decltype(nn)::value;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const float4 a = s_a[j][k0 + u];
                qb[u] = s_b[j][k0 + u];
                qc[u] = s_c[j][k0 + u];
                float t1 = fmaf(a.x, uu, a.w);
                t1 = fmaf(a.y, vv, t1);
                const float t2 = fmaf(a.z, vv, qb[u].x);
                float pw = fmaf(uu, t1, qb[u].y);
                pw = fmaf(vv, t2, pw);
                alpha[u] = fminf(__builtin_amdgcn_exp2f(pw), 0.99f);
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                G2PC_PIN(alpha[u]);
                G2PC_PIN(qb[u].z); G2PC_PIN(qb[u].w); G2PC_PIN(qc[u].x); G2PC_PIN(qc[u].y);
            }
        };
        auto recur = [&](auto nn, int j, const float* alpha, const float4* qb, const float2* qc, float* contrib) -> bool {
            constexpr int N = decltype(nn)::value;
            bool any_cand = false;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                contrib[u] = T[j] * alpha[u];
                cr[j] = fmaf(contrib[u], qb[u].z, cr[j]);
                cg[j] = fmaf(contrib[u], qb[u].w, cg[j]);
                cb[j] = fmaf(contrib[u], qc[u].x, cb[j]);
                T[j] -= contrib[u];
                any_cand = any_cand || (contrib[u] >= qc[u].y);
            }
            return any_cand;
        };
        auto publish = [&](auto nn, int j, int k0, const float* contrib, const float2* qc) {
            constexpr int N = decltype(nn)::value;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (__any(contrib[u] >= qc[u].y)) {
                    const uint32_t bits = __float_as_uint(contrib[u]);
                    const uint32_t m = wave_max_u32_dpp(bits);
                    // the pixel index grows with the lane inside a sub-block: the lowest lane at the maximum owns it;
                    // ties between the two sub-blocks are settled by the packed key itself (lower pixel = larger key)
                    const unsigned long long at_max = __ballot(bits == m);
                    const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane(pix[j], __ffsll(at_max) - 1);
                    if (lane == 0) {
                        unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~(order_tile | pm));
                        atomicMax(&best_key[s_g[j][k0 + u]], key);
                    }
                }
            }
        };
        constexpr int UF = U / 2;                        // in-step trips: UF entries of each list (same number of exp chains in flight)
        const std::integral_constant<int, UF> nf;
        const std::integral_constant<int, U> nu;
        visits += (uint32_t)(cnt[0] + cnt[1]);                                       // wave-uniform (diagnostics)
        const int c0 = (cnt[0] + U - 1) / U * U, c1 = (cnt[1] + U - 1) / U * U;      // entries up to the next multiple of U are neutral
        const int cboth = c0 < c1 ? c0 : c1;
        for (int k0 = 0; k0 < cboth; k0 += UF) {
            float al0[UF], al1[UF], ct0[UF], ct1[UF];
            float4 qb0[UF], qb1[UF];
            float2 qc0[UF], qc1[UF];
            weights(nf, 0, k0, al0, qb0, qc0);
            weights(nf, 1, k0, al1, qb1, qc1);
            const bool a0 = recur(nf, 0, al0, qb0, qc0, ct0);
            const bool a1 = recur(nf, 1, al1, qb1, qc1, ct1);
            if (__any((a0 || a1) ? 1 : 0)) {
                publish(nf, 0, k0, ct0, qc0);
                publish(nf, 1, k0, ct1, qc1);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cj = j == 0 ? c0 : c1;
            for (int k0 = cboth; k0 < cj; k0 += U) {
                float al[U], ct[U];
                float4 qb[U];
                float2 qc[U];
                weights(nu, j, k0, al, qb, qc);
                if (__any(recur(nu, j, al, qb, qc, ct) ? 1 : 0)) publish(nu, j, k0, ct, qc);
            }
            if (!done[j]) done[j] = __all(T[j] <= t_floor ? 1 : 0) != 0;      // see k_blend_py
        }
        if (done[0] && done[1]) break;
        if (lay.walk_cap && processed >= (uint32_t)lay.walk_cap * BL_BATCH) break;      // diagnostic: truncated walk
    }
    if (chunk_work && lane == 0) {                 // diagnostics (+ when and where this wave ran: 100 MHz clock, HW_ID, XCC_ID)
        uint32_t* cw = chunk_work + 8 * (size_t)chunk_i;
        cw[0] = end - start;
        cw[1] = processed;
        cw[2] = (uint32_t)clk0;
        cw[3] = (uint32_t)(wall_clock64() - clk0);
        cw[4] = g2pc_hw_id();
        cw[5] = g2pc_xcc_id();
        cw[6] = visits;                              // (Gaussian, sub-block) pairs that survived the cull
    }
    float* out = tilebuf + 3 * (size_t)lay.tile_pix_off[tile];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (pix[j] >= 0) {
            out[3 * (size_t)pix[j] + 0] = fmaf(T[j], bg, cr[j]);
            out[3 * (size_t)pix[j] + 1] = fmaf(T[j], bg, cg[j]);
            out[3 * (size_t)pix[j] + 2] = fmaf(T[j], bg, cb[j]);
        }
    }
}

// K6 (PY), scalar-gather form of the dual-list kernel (round 4).  In k_blend_py_dl every (Gaussian, sub-block) visit reads
// 40 bytes per lane from LDS as three broadcast reads, and at the VALU-bound visit rate those broadcasts alone would keep the
// LDS pipe ~95 % busy (13.1 cycles per broadcast ds_read_b128 and SIMD, profiles/archive/r02c_valu_rates.json): the kernel sits at
// 57 % of the VALU issue rate with the LDS at 55 % -- two nearly critical resources and five waves per SIMD.  Here only what
// is specific to the (Gaussian, sub-block) pair -- Lu, Lv, K and the running maximum -- is staged in LDS (16 bytes); the
// Gaussian's own A, B, C and colour are read from its 64-byte record with SCALAR loads (the list entry's index is wave-
// uniform; the record array is read through the constant address space: s_load_dword* through the scalar cache) and enter
// the FMAs as SGPR operands (one per instruction: the gfx9 constant-bus limit is met by the operand order below).  The
// scalar loads of trip t + 1 are issued before trip t is blended (two SGPR sets, ping-pong).  Same arithmetic in the same
// order per pixel and list as k_blend_py_dl: bit-identical results.  LDS traffic per visit 40 -> 20 bytes, 70 VGPRs.
template <int U>
__global__ __launch_bounds__(BL_T) void k_blend_py_sg(Layout lay, const int32_t* __restrict__ chunk_tile,
                                                     const int32_t* __restrict__ chunk_pix0,
                                                     const uint2* __restrict__ tile_range,
                                                     const uint32_t* __restrict__ inst_g, uint32_t gmask,
                                                     const float4* __restrict__ rec,
                                                     unsigned long long* __restrict__ best_key, uint32_t order_base,
                                                     float t_floor, float bg, float* __restrict__ tilebuf,
                                                     uint32_t* __restrict__ chunk_work,
                                                     const G2pcCameraJob* __restrict__ job, size_t cs) {
    const unsigned chunk_i = blockIdx.z * gridDim.y + blockIdx.y;       // the blends: camera = blockIdx.x, chunk in (y, z)
    if ((int)chunk_i >= lay.num_chunks) return;
    tile_range = seg_at(tile_range, cs, blockIdx.x); inst_g = seg_at(inst_g, cs, blockIdx.x); rec = seg_at(rec, cs, blockIdx.x);
    if (job) {
        job += blockIdx.x;
        order_base = job->camera_slot << (12 + lay.seq_bits); t_floor = job->t_floor; bg = job->cam.bg[0];
        const unsigned long long tb = ((unsigned long long)job->tilebuf_hi << 32) | job->tilebuf_lo;
        if (tb) tilebuf = (float*)tb;              // one colour buffer per camera (deferred colour resolve)
    }
    const float G2PC_CONSTANT* crecf = (const float G2PC_CONSTANT*)rec;        // written by k_preprocess_py, read-only here
    const unsigned long long clk0 = chunk_work ? wall_clock64() : 0ull;     // diagnostics only
    __shared__ float4 s_a[2][BL_BATCH + 2 * U];     // Lu, Lv, K, max(running maximum, FLT_MIN)
    __shared__ uint32_t s_g[2][BL_BATCH + 2 * U];   // the Gaussian (its record holds A, B, C and the colour)
    const int tile = chunk_tile[chunk_i];
    const uint32_t sbpair = (uint32_t)chunk_pix0[chunk_i];    // a | b << 16, b = 0xFFFF: none
    const int ix = tile % lay.nx, iy = tile / lay.nx;
    const int x0 = lay.xs[ix], w = lay.ws[ix], y0 = lay.ys[iy], h = lay.hs[iy];
    const int nsbx = (w + 7) >> 3;
    const uint32_t order_tile = order_base | ((uint32_t)lay.tile_seq[tile] << 12);
    const unsigned lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const float uu = (float)lx - 3.5f, vv = (float)ly - 3.5f;

    int pix[2];
    float T[2], cr[2], cg[2], cb[2], ox[2], oy[2], rx1[2], ry1[2];
    bool done[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sb = (int)((sbpair >> (16 * j)) & 0xFFFFu);
        const bool present = sb != 0xFFFF;
        const int sx = present ? (sb % nsbx) * 8 : 0, sy = present ? (sb / nsbx) * 8 : 0;
        const int x = sx + lx, y = sy + ly;
        const bool valid = present && (x < w) && (y < h);
        pix[j] = valid ? y * w + x : -1;
        T[j] = valid ? 1.0f : 0.0f;
        cr[j] = cg[j] = cb[j] = 0.0f;
        ox[j] = (float)(x0 + sx) + 3.5f;
        oy[j] = (float)(y0 + sy) + 3.5f;
        rx1[j] = (float)(x0 + (sx + 7 > w - 1 ? w - 1 : sx + 7));   // the cull rectangle stops at the tile's edge
        ry1[j] = (float)(y0 + (sy + 7 > h - 1 ? h - 1 : sy + 7));
        done[j] = !present;
    }
    const bool cull = t_floor > 0.0f;        // t_floor = 0 is the to-the-letter mode: nothing is skipped

    const uint2 se = tile_range[tile];        // k_tile_gate: [first, end) of the tile's instances, empty for a gated tile
    const uint32_t start = se.x, end = se.y;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t g_pad = start < end ? (inst_g[start] & gmask) : 0u;
    uint32_t g_cur = 0, g_nxt = 0;
    bool v_cur = (start + lane) < end, v_nxt = (start + BL_BATCH + lane) < end;
    if (v_cur) g_cur = inst_g[start + lane] & gmask;
    if (v_nxt) g_nxt = inst_g[start + BL_BATCH + lane] & gmask;
    float4 r0 = zero4, r1 = zero4, r2 = zero4;
    uint32_t gmb = 0x7F000000u;
    if (v_cur) {
        r0 = rec[4 * (size_t)g_cur];
        r1 = rec[4 * (size_t)g_cur + 1];
        r2 = rec[4 * (size_t)g_cur + 2];
        gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
    }
    uint32_t processed = 0, visits = 0;
    for (uint32_t b = start; b < end; b += BL_BATCH) {
        processed = b + BL_BATCH - start;
        if (processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(2);
        wave_sync();
        int cnt[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (done[j]) continue;                          // wave-uniform
            const bool keep = v_cur && (!cull || chunk_may_touch(r0, r1, r2.w, ox[j] - 3.5f, rx1[j], oy[j] - 3.5f, ry1[j]));
            const unsigned long long kept = __ballot(keep ? 1 : 0);
            cnt[j] = __popcll(kept);
            if (keep) {
                const int pos = __popcll(kept & ((1ull << lane) - 1ull));
                const float mx = r0.x - ox[j], my = r0.y - oy[j];
                const float A = r0.z, B = r0.w, C = r1.x;
                const float h1 = fmaf(A, mx, B * my);                                     // A mx + B my
                const float Lu = -(fmaf(A, mx, h1)), Lv = -(fmaf(2.0f * C, my, B * mx));
                const float K = fmaf(h1, mx, fmaf(C * my, my, -25.5f - r2.w));            // ... + log2(opacity)
                s_a[j][pos] = make_float4(Lu, Lv, K, fmaxf(__uint_as_float(gmb), 1.17549435e-38f));
                s_g[j][pos] = g_cur;
            }
            if (lane < (unsigned)(2 * U)) {                 // the last trip reads up to U - 1 entries past cnt, the prefetch U more: alpha = 0 ones
                s_a[j][cnt[j] + lane] = make_float4(0.f, 0.f, -INFINITY, 1.17549435e-38f);
                s_g[j][cnt[j] + lane] = g_pad;               // any valid record: K = -inf makes alpha 0
            }
        }
        g_cur = g_nxt;
        v_cur = v_nxt;
        v_nxt = (b + 2 * BL_BATCH + lane) < end;
        g_nxt = 0;
        if (v_nxt) g_nxt = inst_g[b + 2 * BL_BATCH + lane] & gmask;
        r0 = zero4; r1 = zero4; r2 = zero4; gmb = 0x7F000000u;
        if (v_cur) {
            r0 = rec[4 * (size_t)g_cur];
            r1 = rec[4 * (size_t)g_cur + 1];
            r2 = rec[4 * (size_t)g_cur + 2];
            gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
        }
        wave_sync();
        // One trip = U entries of ONE list: the scalar loads of the NEXT trip's records first, then the weights of this trip
        // (U independent exp chains), the transmittance recurrence in depth order and -- rarely, behind one wave-uniform
        // test -- the visibility bookkeeping.
        struct RecS { float A, B, C, r, g, b; };
        // LDS and scalar-memory operations share one counter (lgkmcnt) and scalar loads return out of order, so ANY wait for an
        // LDS read also drains the scalar loads in flight.  A trip therefore does all its LDS reads first -- this trip's
        // (Lu, Lv, K, maximum) and the NEXT trip's list indices -- waits once, then issues the next trip's scalar loads and
        // blends without touching LDS again (the rare publish excepted).
        auto lds_part = [&](int j, int k0, float4* a, uint32_t* idn) {
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = s_a[j][k0 + u]; idn[u] = s_g[j][k0 + U + u]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { G2PC_PIN(a[u].x); G2PC_PIN(a[u].y); G2PC_PIN(a[u].z); G2PC_PIN(a[u].w); G2PC_PIN(idn[u]); }
        };
        auto fetch = [&](const uint32_t* ids, RecS* out) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // wave-uniform index -> scalar loads: (A, B) = dwords 2..3, C = dword 4, colour = dwords 8..10 of the record
                const size_t gu = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ids[u]);
                const float G2PC_CONSTANT* q = crecf + 16 * gu;
                const g2pc_f2v ab = *(const g2pc_f2v G2PC_CONSTANT*)(q + 2);
                const g2pc_f4v col = *(const g2pc_f4v G2PC_CONSTANT*)(q + 8);
                out[u].A = ab[0]; out[u].B = ab[1]; out[u].C = q[4];
                out[u].r = col[0]; out[u].g = col[1]; out[u].b = col[2];
            }
        };
        auto trip = [&](auto jj, int k0, const RecS* rs, const float4* a) {
            constexpr int j = decltype(jj)::value;
            float alpha[U], contrib[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float t1 = fmaf(rs[u].A, uu, a[u].x);
                t1 = fmaf(rs[u].B, vv, t1);
                const float t2 = fmaf(rs[u].C, vv, a[u].y);
                float pw = fmaf(uu, t1, a[u].z);
                pw = fmaf(vv, t2, pw);
                alpha[u] = fminf(__builtin_amdgcn_exp2f(pw), 0.99f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) G2PC_PIN(alpha[u]);
            bool any_cand = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                contrib[u] = T[j] * alpha[u];
                cr[j] = fmaf(contrib[u], rs[u].r, cr[j]);
                cg[j] = fmaf(contrib[u], rs[u].g, cg[j]);
                cb[j] = fmaf(contrib[u], rs[u].b, cb[j]);
                T[j] -= contrib[u];
                any_cand = any_cand || (contrib[u] >= a[u].w);
            }
            if (__any(any_cand ? 1 : 0)) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (__any(contrib[u] >= a[u].w)) {
                        const uint32_t bits = __float_as_uint(contrib[u]);
                        const uint32_t m = wave_max_u32_dpp(bits);
                        // the pixel index grows with the lane inside a sub-block: the lowest lane at the maximum owns it;
                        // ties between the two sub-blocks are settled by the packed key itself (lower pixel = larger key)
                        const unsigned long long at_max = __ballot(bits == m);
                        const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane(pix[j], __ffsll(at_max) - 1);
                        if (lane == 0) {
                            unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~(order_tile | pm));
                            atomicMax(&best_key[s_g[j][k0 + u]], key);
                        }
                    }
                }
            }
        };
        visits += (uint32_t)(cnt[0] + cnt[1]);                                       // wave-uniform (diagnostics)
        auto walk = [&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const int cj = (cnt[j] + U - 1) / U * U;                                 // entries up to the next multiple of U are neutral
            if (cj == 0) return;
            RecS ra[U], rb[U];
            float4 a[U];
            uint32_t idn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) idn[u] = s_g[j][u];
            fetch(idn, ra);
            for (int k0 = 0; k0 < cj; k0 += 2 * U) {                                 // (the list is padded by 2 U neutral entries)
                lds_part(j, k0, a, idn);
                fetch(idn, rb);
                trip(jj, k0, ra, a);
                if (k0 + U >= cj) break;
                lds_part(j, k0 + U, a, idn);
                fetch(idn, ra);
                trip(jj, k0 + U, rb, a);
            }
        };
        if (!done[0]) walk(std::integral_constant<int, 0>());
        if (!done[1]) walk(std::integral_constant<int, 1>());
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (!done[j]) done[j] = __all(T[j] <= t_floor ? 1 : 0) != 0;      // see k_blend_py
        if (done[0] && done[1]) break;
    }
    if (chunk_work && lane == 0) {                 // diagnostics (+ when and where this wave ran: 100 MHz clock, HW_ID, XCC_ID)
        uint32_t* cw = chunk_work + 8 * (size_t)chunk_i;
        cw[0] = end - start;
        cw[1] = processed;
        cw[2] = (uint32_t)clk0;
        cw[3] = (uint32_t)(wall_clock64() - clk0);
        cw[4] = g2pc_hw_id();
        cw[5] = g2pc_xcc_id();
        cw[6] = visits;                              // (Gaussian, sub-block) pairs that survived the cull
    }
    float* out = tilebuf + 3 * (size_t)lay.tile_pix_off[tile];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (pix[j] >= 0) {
            out[3 * (size_t)pix[j] + 0] = fmaf(T[j], bg, cr[j]);
            out[3 * (size_t)pix[j] + 1] = fmaf(T[j], bg, cg[j]);
            out[3 * (size_t)pix[j] + 2] = fmaf(T[j], bg, cb[j]);
        }
    }
}


// K6 (PY), two-wave form of the dual-list kernel (round 4): one 128-thread block per chunk, wave w blends sub-block w.
// A batch is 128 list entries, one per thread; each is tested against BOTH sub-blocks and appended -- in depth order:
// wave 0's survivors before wave 1's -- to the lists it can touch; then every wave walks only ITS list.  Loads, tests,
// (pixel, Gaussian) visits and every floating-point operation are those of k_blend_py_dl (results bit-identical), but a
// chunk's serial chain is half as long: a lone wave issues one VALU instruction per ~6 cycles whatever its instruction-level
// parallelism (profiles/archive/r02c_valu_rates.json: 6.1 cycles with one wave per SIMD, 3.4 with two, 2.45 with eight), and a
// launch lasts as long as its longest walk -- the tail of the single-wave kernel is 2 waves per SIMD on
// average, here the same work is spread over twice the waves.  Two barriers per batch (lists complete / lists consumed).
template <int U>
__global__ __launch_bounds__(2 * BL_T) __attribute__((amdgpu_waves_per_eu(5))) void k_blend_py_2w(Layout lay, const int32_t* __restrict__ chunk_tile,
                                                         const int32_t* __restrict__ chunk_pix0,
                                                         const uint2* __restrict__ tile_range,
                                                         const uint32_t* __restrict__ inst_g, uint32_t gmask,
                                                         const float4* __restrict__ rec,
                                                         unsigned long long* __restrict__ best_key, uint32_t order_base,
                                                         float t_floor, float bg, float* __restrict__ tilebuf,
                                                         uint32_t* __restrict__ chunk_work,
                                                         const G2pcCameraJob* __restrict__ job, size_t cs) {
    const unsigned chunk_i = blockIdx.z * gridDim.y + blockIdx.y;       // camera = blockIdx.x, chunk in (y, z)
    if ((int)chunk_i >= lay.num_chunks) return;
    tile_range = seg_at(tile_range, cs, blockIdx.x); inst_g = seg_at(inst_g, cs, blockIdx.x); rec = seg_at(rec, cs, blockIdx.x);
    if (job) {
        job += blockIdx.x;
        order_base = job->camera_slot << (12 + lay.seq_bits); t_floor = job->t_floor; bg = job->cam.bg[0];
        const unsigned long long tb = ((unsigned long long)job->tilebuf_hi << 32) | job->tilebuf_lo;
        if (tb) tilebuf = (float*)tb;
    }
    const unsigned long long clk0 = chunk_work ? wall_clock64() : 0ull;     // diagnostics only
    constexpr int NB = 2 * BL_BATCH;                // list entries per batch
    __shared__ float4 s_a[2][NB + 4];               // A, B, C, Lu
    __shared__ float4 s_b[2][NB + 4];               // Lv, K, red, green
    __shared__ float2 s_c[2][NB + 4];               // blue, max(running maximum, FLT_MIN)
    __shared__ uint32_t s_g[2][NB];
    __shared__ int s_cnt[2][2][2];                  // [parity of the batch][wave][list] survivors
    __shared__ int s_done[2];                       // sub-block saturated (or absent)
    const int tile = chunk_tile[chunk_i];
    const uint32_t sbpair = (uint32_t)chunk_pix0[chunk_i];    // a | b << 16, b = 0xFFFF: none
    const int ix = tile % lay.nx, iy = tile / lay.nx;
    const int x0 = lay.xs[ix], w = lay.ws[ix], y0 = lay.ys[iy], h = lay.hs[iy];
    const int nsbx = (w + 7) >> 3;
    const uint32_t order_tile = order_base | ((uint32_t)lay.tile_seq[tile] << 12);
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const int wv = (int)(tid >> 6);                 // this wave's sub-block / list
    const int lx = lane & 7, ly = lane >> 3;
    const float uu = (float)lx - 3.5f, vv = (float)ly - 3.5f;

    float ox[2], oy[2], rx1[2], ry1[2];
    bool dn[2];                                     // block-uniform view of s_done, one batch old
    int mypix = -1;
    float T = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sb = (int)((sbpair >> (16 * j)) & 0xFFFFu);
        const bool present = sb != 0xFFFF;
        const int sx = present ? (sb % nsbx) * 8 : 0, sy = present ? (sb / nsbx) * 8 : 0;
        ox[j] = (float)(x0 + sx) + 3.5f;
        oy[j] = (float)(y0 + sy) + 3.5f;
        rx1[j] = (float)(x0 + (sx + 7 > w - 1 ? w - 1 : sx + 7));   // the cull rectangle stops at the tile's edge
        ry1[j] = (float)(y0 + (sy + 7 > h - 1 ? h - 1 : sy + 7));
        dn[j] = !present;
        if (j == wv) {
            const int x = sx + lx, y = sy + ly;
            const bool valid = present && (x < w) && (y < h);
            mypix = valid ? y * w + x : -1;
            T = valid ? 1.0f : 0.0f;
        }
    }
    bool mydone = dn[wv];
    if (lane == 0) s_done[wv] = mydone ? 1 : 0;
    const bool cull = t_floor > 0.0f;        // t_floor = 0 is the to-the-letter mode: nothing is skipped

    const uint2 se = tile_range[tile];        // k_tile_gate: [first, end) of the tile's instances, empty for a gated tile
    const uint32_t start = se.x, end = se.y;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t g_cur = 0, g_nxt = 0;
    bool v_cur = (start + tid) < end, v_nxt = (start + NB + tid) < end;
    if (v_cur) g_cur = inst_g[start + tid] & gmask;
    if (v_nxt) g_nxt = inst_g[start + NB + tid] & gmask;
    float4 r0 = zero4, r1 = zero4, r2 = zero4;
    uint32_t gmb = 0x7F000000u;
    if (v_cur) {
        r0 = rec[4 * (size_t)g_cur];
        r1 = rec[4 * (size_t)g_cur + 1];
        r2 = rec[4 * (size_t)g_cur + 2];
        gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
    }
    uint32_t processed = 0, visits = 0;
    int par = 0;
    for (uint32_t b = start; b < end; b += NB, par ^= 1) {
        processed = b + NB - start;
        // (1) test this thread's entry against both sub-blocks, count the survivors per wave
        bool keep[2];
        unsigned long long kept[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            keep[j] = !dn[j] && v_cur && (!cull || chunk_may_touch(r0, r1, r2.w, ox[j] - 3.5f, rx1[j], oy[j] - 3.5f, ry1[j]));
            kept[j] = __ballot(keep[j] ? 1 : 0);
        }
        if (lane == 0) { s_cnt[par][wv][0] = __popcll(kept[0]); s_cnt[par][wv][1] = __popcll(kept[1]); }
        __syncthreads();                    // counts and s_done published; both waves have left the previous batch's lists
        dn[0] = s_done[0] != 0;
        dn[1] = s_done[1] != 0;
        if (dn[0] && dn[1]) break;          // block-uniform
        int total[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c0 = s_cnt[par][0][j];
            total[j] = c0 + s_cnt[par][1][j];
            if (keep[j] && !dn[j]) {
                const int pos = (wv ? c0 : 0) + __popcll(kept[j] & ((1ull << lane) - 1ull));
                const float mx = r0.x - ox[j], my = r0.y - oy[j];
                const float A = r0.z, B = r0.w, C = r1.x;
                const float h1 = fmaf(A, mx, B * my);                                     // A mx + B my
                const float Lu = -(fmaf(A, mx, h1)), Lv = -(fmaf(2.0f * C, my, B * mx));
                const float K = fmaf(h1, mx, fmaf(C * my, my, -25.5f - r2.w));            // ... + log2(opacity)
                s_a[j][pos] = make_float4(A, B, C, Lu);
                s_b[j][pos] = make_float4(Lv, K, r2.x, r2.y);
                s_c[j][pos] = make_float2(r2.z, fmaxf(__uint_as_float(gmb), 1.17549435e-38f));
                s_g[j][pos] = g_cur;
            }
            if (wv == j && lane < (unsigned)U) {            // the last trip reads up to U - 1 entries past the end: alpha = 0 ones
                s_a[j][total[j] + lane] = zero4;
                s_b[j][total[j] + lane] = make_float4(0.f, -INFINITY, 0.f, 0.f);
                s_c[j][total[j] + lane] = make_float2(0.f, 1.17549435e-38f);
            }
        }
        // loads of the next batch (records) and the one after (ids): they complete under this batch's walk
        g_cur = g_nxt;
        v_cur = v_nxt;
        v_nxt = (b + 2 * NB + tid) < end;
        g_nxt = 0;
        if (v_nxt) g_nxt = inst_g[b + 2 * NB + tid] & gmask;
        r0 = zero4; r1 = zero4; r2 = zero4; gmb = 0x7F000000u;
        if (v_cur) {
            r0 = rec[4 * (size_t)g_cur];
            r1 = rec[4 * (size_t)g_cur + 1];
            r2 = rec[4 * (size_t)g_cur + 2];
            gmb = ((const uint32_t*)best_key)[2 * (size_t)g_cur + 1];   // live running maximum
        }
        __syncthreads();                    // lists complete
        // (2) this wave walks its own list
        if (!mydone) {
            const int cnt = total[wv];
            visits += (uint32_t)cnt;
            for (int k0 = 0; k0 < cnt; k0 += U) {
                float alpha[U], contrib[U];
                float4 qb[U];
                float2 qc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 a = s_a[wv][k0 + u];
                    qb[u] = s_b[wv][k0 + u];
                    qc[u] = s_c[wv][k0 + u];
                    float t1 = fmaf(a.x, uu, a.w);
                    t1 = fmaf(a.y, vv, t1);
                    const float t2 = fmaf(a.z, vv, qb[u].x);
                    float pw = fmaf(uu, t1, qb[u].y);
                    pw = fmaf(vv, t2, pw);
                    alpha[u] = fminf(__builtin_amdgcn_exp2f(pw), 0.99f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    G2PC_PIN(alpha[u]);
                    G2PC_PIN(qb[u].z); G2PC_PIN(qb[u].w); G2PC_PIN(qc[u].x); G2PC_PIN(qc[u].y);
                }
                bool any_cand = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    contrib[u] = T * alpha[u];
                    cr = fmaf(contrib[u], qb[u].z, cr);
                    cg = fmaf(contrib[u], qb[u].w, cg);
                    cb = fmaf(contrib[u], qc[u].x, cb);
                    T -= contrib[u];
                    any_cand = any_cand || (contrib[u] >= qc[u].y);
                }
                if (__any(any_cand ? 1 : 0)) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (__any(contrib[u] >= qc[u].y)) {
                            const uint32_t bits = __float_as_uint(contrib[u]);
                            const uint32_t m = wave_max_u32_dpp(bits);
                            const unsigned long long at_max = __ballot(bits == m);      // lowest lane = lowest pixel index
                            const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane(mypix, __ffsll(at_max) - 1);
                            if (lane == 0) {
                                unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~(order_tile | pm));
                                atomicMax(&best_key[s_g[wv][k0 + u]], key);
                            }
                        }
                    }
                }
            }
            mydone = __all(T <= t_floor ? 1 : 0) != 0;      // see k_blend_py
            if (mydone && lane == 0) s_done[wv] = 1;         // read by both waves after the next batch's first barrier
        }
    }
    if (chunk_work && lane == 0) {
        uint32_t* cw = chunk_work + 8 * (size_t)chunk_i;
        if (wv == 0) {
            cw[0] = end - start;
            cw[1] = processed;
            cw[2] = (uint32_t)clk0;
            cw[4] = g2pc_hw_id();
            cw[5] = g2pc_xcc_id();
        }
        atomicMax(&cw[3], (uint32_t)(wall_clock64() - clk0));
        atomicAdd(&cw[6], visits);
    }
    if (mypix >= 0) {
        float* out = tilebuf + 3 * (size_t)lay.tile_pix_off[tile];
        out[3 * (size_t)mypix + 0] = fmaf(T, bg, cr);
        out[3 * (size_t)mypix + 1] = fmaf(T, bg, cg);
        out[3 * (size_t)mypix + 2] = fmaf(T, bg, cb);
    }
}

// K7 (PY): running update of the per-Gaussian colour: Gaussians whose best key was set by this camera slot take
// the colour of the winning (tile, pixel) from that tile's own rendered colours (gauss_render.py:387-395).
__global__ __launch_bounds__(RA_T) void k_update_colours_py(Layout lay, const unsigned long long* __restrict__ best_key,
                                                           long n, uint32_t slot, const float* __restrict__ tilebuf,
                                                           float* __restrict__ colours_out) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = best_key[i];
    if ((key >> 32) == 0ull) return;
    uint32_t order = ~(uint32_t)key;
    if ((order >> (12 + lay.seq_bits)) != slot) return;
    int seq = (int)((order >> 12) & ((1u << lay.seq_bits) - 1u)) - lay.seq_base, pix = order & 0xFFF;
    if ((unsigned)seq >= (unsigned)lay.seq_count) return;      // a key of another pass of this camera (quad-tree passes)
    int tile = lay.seq_tile[seq];
    const float* src = tilebuf + 3 * ((size_t)lay.tile_pix_off[tile] + pix);
    colours_out[3 * i + 0] = src[0];
    colours_out[3 * i + 1] = src[1];
    colours_out[3 * i + 2] = src[2];
}

// deferred form of K7: every Gaussian takes its colour from the buffer of the camera that holds its key (tilebufs[slot],
// device addresses; 0 = that camera updated on its own)
__global__ __launch_bounds__(RA_T) void k_resolve_colours_py(Layout lay, const unsigned long long* __restrict__ best_key,
                                                            long n, const unsigned long long* __restrict__ tilebufs,
                                                            float* __restrict__ colours_out) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = best_key[i];
    if ((key >> 32) == 0ull) return;
    uint32_t order = ~(uint32_t)key;
    const float* tb = (const float*)tilebufs[order >> (12 + lay.seq_bits)];
    if (!tb) return;
    int seq = (int)((order >> 12) & ((1u << lay.seq_bits) - 1u)) - lay.seq_base, pix = order & 0xFFF;
    if ((unsigned)seq >= (unsigned)lay.seq_count) return;      // set by a quad-tree pass of that camera, which updated on its own
    int tile = lay.seq_tile[seq];
    const float* src = tb + 3 * ((size_t)lay.tile_pix_off[tile] + pix);
    colours_out[3 * i + 0] = src[0];
    colours_out[3 * i + 1] = src[1];
    colours_out[3 * i + 2] = src[2];
}

// keys older than the current epoch: forget their order (order 0 = "earliest possible") but keep the value
__global__ __launch_bounds__(RA_T) void k_rebase_keys(unsigned long long* __restrict__ best_key, long n) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = best_key[i];
    if ((key >> 32) != 0ull) best_key[i] = key | 0xFFFFFFFFull;
}

// multi-GPU winner selection: after the all-reduce(MAX) of the packed keys every rank that holds the winning key
// nominates itself (k_key_owner); an all-reduce(MIN) of the nominations elects exactly ONE rank per Gaussian -- several
// ranks hold the same key after an earlier exchange or a rebase --, the others zero their colours
// (k_keep_winner_colours), and the all-reduce(SUM) of the colours has exactly one non-zero term per Gaussian whatever
// was exchanged before: the exchange is idempotent.
__global__ __launch_bounds__(RA_T) void k_key_owner(const unsigned long long* __restrict__ local_key,
                                                   const unsigned long long* __restrict__ global_key, long n, int rank,
                                                   int32_t* __restrict__ owner) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const unsigned long long g = global_key[i];
    owner[i] = ((g >> 32) != 0ull && local_key[i] == g) ? rank : 0x7FFFFFFF;
}
__global__ __launch_bounds__(RA_T) void k_keep_winner_colours(const int32_t* __restrict__ owner, long n, int rank,
                                                             float* __restrict__ colours) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    if (owner[i] != rank) { colours[3 * i + 0] = 0.0f; colours[3 * i + 1] = 0.0f; colours[3 * i + 2] = 0.0f; }
}

__global__ __launch_bounds__(RA_T) void k_contributions(const unsigned long long* __restrict__ best_key, long n,
                                                       float* __restrict__ out) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i < n) out[i] = __uint_as_float((uint32_t)(best_key[i] >> 32));
}

// image[y][W-1-x] = colour of the LAST tile (processing order) that covers pixel (x,y); flip = gauss_render.py:402
__global__ __launch_bounds__(RA_T) void k_assemble_image_py(Layout lay, int W, int H, const float* __restrict__ tilebuf,
                                                           float* __restrict__ image) {
    long t = (long)blockIdx.x * RA_T + threadIdx.x;
    if (t >= (long)W * H) return;
    int x = (int)(t % W), y = (int)(t / W);
    int bx[2], by[2], nbx = 0, nby = 0;
    for (int i = 0; i < lay.nx && nbx < 2; ++i) if (x >= lay.xs[i] && x < lay.xs[i] + lay.ws[i]) bx[nbx++] = i;
    for (int i = 0; i < lay.ny && nby < 2; ++i) if (y >= lay.ys[i] && y < lay.ys[i] + lay.hs[i]) by[nby++] = i;
    int best_seq = -1, best_tile = -1;
    for (int a = 0; a < nby; ++a)
        for (int b = 0; b < nbx; ++b) {
            int tile = by[a] * lay.nx + bx[b];
            if (lay.tile_mask && !lay.tile_mask[tile]) continue;    // not painted by this pass
            int s = lay.tile_seq[tile];
            if (s > best_seq) { best_seq = s; best_tile = tile; }
        }
    if (lay.tile_mask && best_tile < 0) return;   // compose mode: the pixel keeps what earlier passes / fills left there
    float r = 1.0f, g = 1.0f, bl = 1.0f;          // torch.ones init (gauss_render.py:287)
    if (best_tile >= 0) {
        int ix = best_tile % lay.nx, iy = best_tile / lay.nx;
        int lp = (y - lay.ys[iy]) * lay.ws[ix] + (x - lay.xs[ix]);
        const float* src = tilebuf + 3 * ((size_t)lay.tile_pix_off[best_tile] + lp);
        r = src[0]; g = src[1]; bl = src[2];
    }
    float* dst = image + 3 * ((size_t)y * W + (W - 1 - x));
    dst[0] = r; dst[1] = g; dst[2] = bl;
}

// =========================================================================================================
// Semantics "CU" = the reference's native rasteriser (renderer_type="cuda"), deterministic spec of SURVEY §8(a.5).
// =========================================================================================================
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// forward.cu:153-271 (preprocessCUDA) + :22-73 (computeColorFromSH) + :76-111 (computeCov2D); 16x16 tile rect of
// auxiliary.h:45-55.  Depth-sort input is written in ASCENDING index order: the reference's stable radix sort of
// (tile << 32 | depth bits) keeps equal depths in ascending Gaussian index.
__global__ __launch_bounds__(RA_T) void k_preprocess_cu(Cam cam, int grid_x, int grid_y,
                                                       const float* __restrict__ means3D,
                                                       const float* __restrict__ cov6,
                                                       const float* __restrict__ opacity,
                                                       const float* __restrict__ colours_precomp,
                                                       const float* __restrict__ shs, int sh_degree, int sh_coeffs,
                                                       float3 campos, long n, uint32_t* __restrict__ depth_key,
                                                       uint32_t* __restrict__ index, uint32_t* __restrict__ tiles_touched,
                                                       float4* __restrict__ rec, uint32_t* __restrict__ rect,
                                                       int32_t* __restrict__ radii, int wide) {
#pragma clang fp contract(off)
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    const float* V = cam.V;
    const float* P = cam.P;
    uint32_t key = 0xFFFFFFFFu, touched = 0, rc = 0, rc_hi = 0;
    int rad = 0;
    // Everything that decides an INTEGER of the reference (radius, tile rectangle, depth bits -> order) is evaluated
    // below with the reference's own expressions, operation by operation in source order, every operation rounded on its
    // own (contraction is off for this kernel).  That is the one evaluation of the reference's text that does not depend
    // on a compiler's choice of which product to fuse (gcc fuses the FIRST product of transformPoint4x3 but the LAST two of
    // transformPoint4x4, tools/cu_preprocess_exactness.py; nvcc's choices cannot be observed here), and it is what
    // oracle/_ref -- the reference's .cu files compiled with -ffp-contract=off -- computes: radii, tile rectangles,
    // num_rendered, depths and projected means equal bit for bit (tests/golden/render_cu_*).
    const float tz0 = V[2] * x + V[6] * y + V[10] * z + V[14];
    if (tz0 > 0.2f) {                                                   // in_frustum (auxiliary.h:166)
        float hx = P[0] * x + P[4] * y + P[8] * z + P[12];
        float hy = P[1] * x + P[5] * y + P[9] * z + P[13];
        float hw = P[3] * x + P[7] * y + P[11] * z + P[15];
        float pw = 1.0f / (hw + 0.0000001f);
        const float focal_x = (float)cam.W / (2.0f * cam.tan_fovx), focal_y = (float)cam.H / (2.0f * cam.tan_fovy);
        float tx = V[0] * x + V[4] * y + V[8] * z + V[12];
        float ty = V[1] * x + V[5] * y + V[9] * z + V[13];
        float tz = tz0;
        float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
        tx = fminf(limx, fmaxf(-limx, tx / tz)) * tz;
        ty = fminf(limy, fmaxf(-limy, ty / tz)) * tz;
        // T = W J with glm's column-major constructors (forward.cu:91-101): column 0 of J is (fx/tz, 0, -fx tx/tz^2),
        // column 1 is (0, fy/tz, -fy ty/tz^2), column 2 is zero.  The products with those zeros add +-0 and are left out.
        float j00 = focal_x / tz, j11 = focal_y / tz, j02 = -(focal_x * tx) / (tz * tz), j12 = -(focal_y * ty) / (tz * tz);
        float T0[3], T1[3];                                              // columns 0 and 1 of T, indexed by row
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T0[r] = V[4 * r + 0] * j00 + V[4 * r + 2] * j02;
            T1[r] = V[4 * r + 1] * j11 + V[4 * r + 2] * j12;
        }
        const float* c = cov6 + 6 * i;
        float S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
        float a0[3], a1[3];                                              // rows 0,1 of T^T Vrk^T
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a0[k] = T0[0] * S[k][0] + T0[1] * S[k][1] + T0[2] * S[k][2];
            a1[k] = T1[0] * S[k][0] + T1[1] * S[k][1] + T1[2] * S[k][2];
        }
        float cxx = a0[0] * T0[0] + a0[1] * T0[1] + a0[2] * T0[2] + 0.3f;
        float cxy = a1[0] * T0[0] + a1[1] * T0[1] + a1[2] * T0[2];
        float cyy = a1[0] * T1[0] + a1[1] * T1[1] + a1[2] * T1[2] + 0.3f;
        float det = cxx * cyy - cxy * cxy;
        if (det != 0.0f) {
            float di = 1.0f / det;
            float kx = cyy * di, ky = -cxy * di, kz = cxx * di;
            float mid = 0.5f * (cxx + cyy);
            float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            float l1 = mid + sq, l2 = mid - sq;
            float my_radius = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
            float px = (float)((((double)(hx * pw) + 1.0) * cam.W - 1.0) * 0.5);     // ndc2Pix in double (auxiliary.h:40-43)
            float py = (float)((((double)(hy * pw) + 1.0) * cam.H - 1.0) * 0.5);
            int r = (int)my_radius;
            int x0 = (int)((px - r) / 16), y0 = (int)((py - r) / 16);
            int x1 = (int)((px + r + 15) / 16), y1 = (int)((py + r + 15) / 16);
            x0 = min(grid_x, max(0, x0)); y0 = min(grid_y, max(0, y0));
            x1 = min(grid_x, max(0, x1)); y1 = min(grid_y, max(0, y1));
            if ((x1 - x0) * (y1 - y0) != 0) {
                touched = (uint32_t)((x1 - x0) * (y1 - y0));
                if (wide) {                    // grids beyond 256 tiles per axis: 16-bit tile coordinates in two words
                    rc = (uint32_t)x0 | ((uint32_t)(x1 - 1) << 16);
                    rc_hi = (uint32_t)y0 | ((uint32_t)(y1 - 1) << 16);
                } else {
                    rc = (uint32_t)x0 | ((uint32_t)(x1 - 1) << 8) | ((uint32_t)y0 << 16) | ((uint32_t)(y1 - 1) << 24);
                }
                key = __float_as_uint(tz0);
                rad = r;
                const float sc = LOG2E;
                const float qa = -0.5f * sc * kx, qb = -sc * ky, qc = -0.5f * sc * kz;
                rec[4 * i + 0] = make_float4(px, py, qa, qb);                           // one 64-byte record per Gaussian,
                rec[4 * i + 1] = make_float4(qc, opacity[i], tz0, my_radius);           // as on the PY path
                // per-wave cull of k_blend_cu (rect_may_touch): slopes of the exponent's edge maxima and the exponent below
                // which alpha < 1/255 (with a 0.7 % margin for the different rounding of the bound)
                rec[4 * i + 3] = make_float4(-qb / (2.0f * qc), -qb / (2.0f * qa), -8.00435f - log2f(opacity[i]), 0.0f);
                float cr, cg, cb;
                if (colours_precomp) {
                    cr = colours_precomp[3 * i]; cg = colours_precomp[3 * i + 1]; cb = colours_precomp[3 * i + 2];
                } else {
                    float dx = x - campos.x, dy = y - campos.y, dz = z - campos.z;
                    float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dx /= len; dy /= len; dz /= len;
                    const float* sh = shs + (size_t)i * sh_coeffs * 3;
                    float res[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.28209479177387814f * sh[ch];
                        if (sh_degree > 0) {
                            v = v - 0.4886025119029199f * dy * sh[3 + ch] + 0.4886025119029199f * dz * sh[6 + ch] -
                                0.4886025119029199f * dx * sh[9 + ch];
                            if (sh_degree > 1) {
                                float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                                v = v + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                                    kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                                    kSH_C2[4] * (xx - yy) * sh[24 + ch];
                                if (sh_degree > 2) {
                                    v = v + kSH_C3[0] * dy * (3.0f * xx - yy) * sh[27 + ch] + kSH_C3[1] * xy * dz * sh[30 + ch] +
                                        kSH_C3[2] * dy * (4.0f * zz - xx - yy) * sh[33 + ch] +
                                        kSH_C3[3] * dz * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                                        kSH_C3[4] * dx * (4.0f * zz - xx - yy) * sh[39 + ch] +
                                        kSH_C3[5] * dz * (xx - yy) * sh[42 + ch] + kSH_C3[6] * dx * (xx - 3.0f * yy) * sh[45 + ch];
                                }
                            }
                        }
                        v += 0.5f;
                        res[ch] = v < 0.0f ? 0.0f : v;
                    }
                    cr = res[0]; cg = res[1]; cb = res[2];
                }
                rec[4 * i + 2] = make_float4(cr, cg, cb, 0.0f);
            }
        }
    }
    depth_key[i] = key;
    index[i] = (uint32_t)i;
    tiles_touched[i] = touched;
    if (wide) { rect[2 * i] = rc; rect[2 * i + 1] = rc_hi; } else rect[i] = rc;
    radii[i] = rad;
}

// forward.cu:303-497 (renderCUDA).  One 256-thread block per 16x16 tile, one pixel per lane (thread rank t -> pixel
// (t % 16, t / 16), as in the reference); the tile's list is staged 256 instances at a time (= the reference's batches:
// the unit of the "everyone done" test and of the surface-distance pass).  Inside a batch the four waves run
// independently: 4 Gaussians per trip (independent exp chains), wave64 DPP reductions for the per-Gaussian maximum and
// for the surface distance, each guarded by a cheap "can any lane improve the staged value" ballot.
constexpr int CU_T = 256;

__global__ __launch_bounds__(CU_T) void k_blend_cu(int W, int H, int grid_x, int tile_first, int tile_step,
                                                  const uint32_t* __restrict__ tile_start,
                                                  const uint32_t* __restrict__ inst_g, uint32_t gmask, const float4* __restrict__ rec,
                                                  const int32_t* __restrict__ mask, float3 bg, int calc_surf,
                                                  unsigned long long* __restrict__ cam_key,
                                                  uint32_t* __restrict__ cam_surf, float* __restrict__ out_color,
                                                  float* __restrict__ out_depth, float* __restrict__ out_invdepth) {
    __shared__ float4 s_p0[CU_T + 1];             // slot CU_T: a neutral entry (opacity 0) the per-wave lists are padded with
    __shared__ float4 s_p1[CU_T + 1];
    __shared__ float4 s_p2[CU_T + 1];
    __shared__ uint32_t s_g[CU_T + 1];
    __shared__ uint32_t s_surf[CU_T];             // surface distance known when the batch was staged (filter only)
    __shared__ unsigned short s_list[4][CU_T + 4];   // per wave: the batch entries that can reach its 16x4 pixels, in depth order
    __shared__ int s_wc[4][4];                    // [list][staging wave] survivors
    const int tile = tile_first + (int)blockIdx.x * tile_step;       // (first, step) != (0, 1): this rank's share of the tiles
    const int tx = tile % grid_x, ty = tile / grid_x;
    const unsigned t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int x = tx * 16 + (int)(t & 15), y = ty * 16 + (int)(t >> 4);
    const bool inside = (x < W) && (y < H);
    const bool masked = inside && mask && (mask[(size_t)W * y + x] == 0);
    const bool part = inside && !masked;              // takes part in blending
    const bool surf_part = !inside || part;           // out-of-image threads take part (E = 0), masked pixels do not
    bool done = !part;
    const float px = (float)x, py = (float)y;
    const uint32_t pixid = (uint32_t)(W * y + x);
    float T = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, E = 0.f, Ei = 0.f;
    const uint32_t* key_hi = (const uint32_t*)cam_key + 1;
    const uint32_t start = tile_start[tile], end = tile_start[tile + 1];
    // pixel rectangle of wave w inside the image (the four waves of a tile own four 16x4 strips)
    const float rx0 = (float)(tx * 16), rx1 = (float)min(tx * 16 + 15, W - 1);
    if (t == 0) {
        s_p0[CU_T] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_p1[CU_T] = make_float4(0.f, 0.f, 1.f, 0.f);
        s_p2[CU_T] = make_float4(0.f, 0.f, 0.f, 3.0e38f);
        s_g[CU_T] = 0;
    }
    for (uint32_t b = start; b < end; b += CU_T) {
        if (__syncthreads_and(done ? 1 : 0)) break;                       // forward.cu:373-375 (also: LDS is free again)
        // Stage entry t and decide, for each of the tile's four waves, whether this Gaussian's alpha can reach 1/255 on
        // that wave's pixels (rect_may_touch).  Below it the reference's loop body does nothing for the pixel (forward.cu:
        // 411-413 `continue`), so a Gaussian that fails for all 64 pixels of a wave is not walked by that wave at all --
        // same results bit for bit, ~half the (pixel, Gaussian) pairs of a 16x16 tile never evaluated.
        bool keep[4] = {false, false, false, false};
        if (b + t < end) {
            uint32_t g = inst_g[b + t] & gmask;
            const float4 r0 = rec[4 * (size_t)g], r1 = rec[4 * (size_t)g + 1], r3 = rec[4 * (size_t)g + 3];
            s_p0[t] = r0;
            s_p1[t] = r1;
            const float4 c3 = rec[4 * (size_t)g + 2];
            float gm = fmaxf(__uint_as_float(key_hi[2 * (size_t)g]), 1.17549435e-38f);
            s_p2[t] = make_float4(c3.x, c3.y, c3.z, gm);
            s_g[t] = g;
            if (calc_surf) s_surf[t] = cam_surf[g];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int wy0 = ty * 16 + 4 * w;
                keep[w] = wy0 < H && rect_may_touch(r0.x, r0.y, r0.z, r0.w, r1.x, r3.x, r3.y, r3.z, rx0, rx1, (float)wy0,
                                                    (float)min(wy0 + 3, H - 1));
            }
        } else {                                   // padding: opacity 0 -> alpha 0 < 1/255 -> skipped
            s_p0[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_p1[t] = make_float4(0.f, 0.f, 1.f, 0.f);
            s_p2[t] = make_float4(0.f, 0.f, 0.f, 3.0e38f);
            s_g[t] = 0;
            s_surf[t] = 0;
        }
        unsigned long long kept[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            kept[w] = __ballot(keep[w] ? 1 : 0);
            if (lane == 0) s_wc[w][wv] = __popcll(kept[w]);
        }
        __syncthreads();
        int lcnt = 0;                                                   // survivors on this wave's list
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int off = 0, tot = 0;
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) { const int c = s_wc[w][sw]; if (sw < (int)wv) off += c; tot += c; }
            if (keep[w]) s_list[w][off + __popcll(kept[w] & ((1ull << lane) - 1ull))] = (unsigned short)t;
            if ((int)wv == w) {
                lcnt = tot;
                if (lane < 4) s_list[w][tot + lane] = (unsigned short)CU_T;      // the last trip reads up to 3 entries past the end
            }
        }
        __syncthreads();
        const int cnt = (end - b) < (uint32_t)CU_T ? (int)(end - b) : CU_T;
        // wave-uniform early out inside the batch: nothing left to blend for these 64 pixels
        for (int i0 = 0; i0 < lcnt && !__all(done ? 1 : 0); i0 += 4) {
            float alpha[4], power[4], dep[4];
            float4 cc[4];
            int kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kk[u] = (int)s_list[wv][i0 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = s_p2[kk[u]];       // read with the rest: the serial part never waits on LDS
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 a = s_p0[kk[u]], q = s_p1[kk[u]];
                float dx = a.x - px, dy = a.y - py;
                power[u] = fmaf(dx, fmaf(a.w, dy, a.z * dx), (q.x * dy) * dy);
                alpha[u] = fminf(0.99f, q.y * __builtin_amdgcn_exp2f(power[u]));
                dep[u] = q.z;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                G2PC_PIN(alpha[u]); G2PC_PIN(dep[u]);
                G2PC_PIN(cc[u].x); G2PC_PIN(cc[u].y); G2PC_PIN(cc[u].z); G2PC_PIN(cc[u].w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 c = cc[u];
                const float depth = dep[u];
                float test_T = T * (1.0f - alpha[u]);
                bool live = !done && !(power[u] > 0.0f) && !(alpha[u] < 1.0f / 255.0f);
                bool stop = live && (test_T < 0.0001f);
                done = done || stop;
                bool blend = live && !stop;
                float contrib = blend ? alpha[u] * T : 0.0f;
                cr = fmaf(c.x, contrib, cr);
                cg = fmaf(c.y, contrib, cg);
                cb = fmaf(c.z, contrib, cb);
                Ei = fmaf(1.0f / depth, contrib, Ei);
                E = fmaf(depth, contrib, E);
                T = blend ? test_T : T;
                if (__any(contrib >= c.w)) {
                    uint32_t bits = __float_as_uint(contrib);
                    uint32_t m = wave_max_u32_dpp(bits);
                    // pixel id grows with the lane inside a wave (4 rows of the 16x16 tile): lowest lane at the maximum
                    const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane((int)pixid, __ffsll(__ballot(bits == m)) - 1);
                    if (lane == 0) {
                        unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(uint32_t)(~pm);
                        atomicMax(&cam_key[s_g[kk[u]]], key);
                    }
                }
            }
        }
        if (calc_surf) {                                                   // forward.cu:460-477
            __syncthreads();                                               // E of the whole batch is final for this wave
            // min over this wave's pixels of |depth_k - E_p| for every Gaussian k of the batch.  E_p >= 0 and almost every
            // depth_k lies above (or below) ALL 64 expected depths -- the minimum is then |depth_k - Emax| (or Emin), the
            // very subtraction the pixel holding that extreme would do -- so the lanes first go through the batch 64
            // Gaussians at a time, one k per lane, and only a depth strictly inside (Emin, Emax) needs the per-pixel pass.
            const uint32_t ebits = __float_as_uint(E);                     // non-negative floats order like their bits
            const uint32_t emax_b = wave_max_u32_dpp(surf_part ? ebits : 0u);
            const uint32_t emin_b = wave_min_u32_dpp(surf_part ? ebits : 0xFFFFFFFFu);
            if (emin_b != 0xFFFFFFFFu) {                                   // some pixel of this wave takes part
                const float emax = __uint_as_float(emax_b), emin = __uint_as_float(emin_b);
                for (int k0 = 0; k0 < cnt; k0 += 64) {
                    const int k = k0 + (int)lane;
                    bool inside = false;
                    if (k < cnt) {
                        const float z = s_p1[k].z;
                        if (z >= emax || z <= emin) {
                            const uint32_t bits = __float_as_uint(fabsf(z - (z >= emax ? emax : emin)));
                            if (bits < s_surf[k]) atomicMin(&cam_surf[s_g[k]], bits);
                        } else {
                            inside = true;
                        }
                    }
                    unsigned long long todo = __ballot(inside);
                    while (todo) {
                        const int kk = k0 + __ffsll(todo) - 1;
                        todo &= todo - 1;
                        float d = fabsf(s_p1[kk].z - E);
                        const uint32_t bits = surf_part ? __float_as_uint(d) : 0x7F7FFFFFu;
                        if (__any(bits < s_surf[kk])) {
                            uint32_t m = wave_min_u32_dpp(bits);
                            if (lane == 0) atomicMin(&cam_surf[s_g[kk]], m);
                        }
                    }
                }
            }
        }
    }
    if (part) {
        const size_t plane = (size_t)W * H;
        out_color[pixid] = fmaf(T, bg.x, cr);
        out_color[plane + pixid] = fmaf(T, bg.y, cg);
        out_color[2 * plane + pixid] = fmaf(T, bg.z, cb);
        out_invdepth[pixid] = Ei;
        out_depth[pixid] = E;
    }
}

// tile_start[t] = first sorted instance of tile t (exclusive offsets, tile_start[T] = L): boundary detection on the
// sorted tile ids (rasterizer_impl.cu:115-137 identifyTileRanges), no histogram, no scan.
__global__ __launch_bounds__(RA_T) void k_tile_ranges(const uint32_t* __restrict__ tile_sorted, long L, int T,
                                                     uint32_t* __restrict__ tile_start,
                                                     const uint32_t* __restrict__ l_dev, int gshift, size_t cs) {
    tile_sorted = seg(tile_sorted, cs); tile_start = seg(tile_start, cs); l_dev = seg(l_dev, cs);
    if (l_dev) L = (long)*l_dev;                 // capacity-sized launch, count on the device
    long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l > L) return;
    int prev = l == 0 ? -1 : (int)(tile_sorted[l - 1] >> gshift);      // gshift > 0: packed (tile << gshift | Gaussian) instances
    int cur = l == L ? T : (int)(tile_sorted[l] >> gshift);
    for (int t = prev + 1; t <= cur; ++t) tile_start[t] = (uint32_t)l;      // every t in [0, T] is written exactly once:
    if (l == L) tile_start[T + 1] = (uint32_t)L;                              // no memset of tile_start is needed
}

// The pixel rectangle of Gaussian i as k_preprocess_py forms it (gauss_render.py:151-193, 435-436): the same device
// functions in the same order, so the floats are the preprocess's own.  false: outside projection_ndc's in_mask.
__device__ __forceinline__ bool py_rect(const Cam& cam, const float* __restrict__ means3D, const float* __restrict__ cov9,
                                        long i, float r[4]) {
    float pv[4];
    py_view(cam.V, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], pv);
    if (!(pv[2] <= -0.000001f)) return false;
    float cv[4], ph[4], det;
    py_cov2d(cam.V, pv, cam.lim_x, cam.lim_y, cam.focal_x, cam.focal_y, cov9 + 9 * i, cv);
    py_hom(cam.P, pv, ph);
    const float pw = 1.0f / (ph[3] + 0.000001f);
    const float ndx = ph[0] * pw, ndy = ph[1] * pw;
    const float mx = ((ndx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;
    const float my = ((ndy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
    const float radius = py_radius(cv, det);
    const float wmax = (float)cam.W - 1.0f, hmax = (float)cam.H - 1.0f;
    r[0] = fminf(fmaxf(mx - radius, 0.0f), wmax); r[1] = fminf(fmaxf(mx + radius, 0.0f), wmax);
    r[2] = fminf(fmaxf(my - radius, 0.0f), hmax); r[3] = fminf(fmaxf(my + radius, 0.0f), hmax);
    return (mx == mx) && (my == my) && (det == det);      // NaNs belong to no tile (k_preprocess_py's `ok`)
}
// membership of a rectangle in the node [x0, x1] x [y0, y1] (inclusive pixels): gauss_render.py:306-309
__device__ __forceinline__ bool rect_in_node(const float r[4], int x0, int x1, int y0, int y1) {
    const float tlx = fmaxf(r[0], (float)x0), brx = fminf(r[1], (float)x1);
    const float tly = fmaxf(r[2], (float)y0), bry = fminf(r[3], (float)y1);
    return (brx > tlx) && (bry > tly);
}

// The reference's quad-tree is data dependent in two ways the fixed leaf grid does not show (gauss_render.py:311-335):
//  * a leaf holding more than max_gaussians_per_tile Gaussians is split further (:319) -- state 1: the leaf is NOT blended
//    here, the host renders its children in further passes (GaussHipRenderer._quadtree_passes);
//  * a node without any Gaussian is painted with the background and its children are never visited (:311-314), while a
//    child reaches up to one pixel per odd split beyond its parent (:321-334): a leaf whose members all live in that
//    strip of an otherwise empty ancestor is never blended by the reference -- state 2 | level << 8.  A leaf that lies
//    inside an ancestor proves that ancestor non-empty by having a member, so only the levels a leaf sticks out of
//    (tile_stick, a handful of border leaves) are examined: first the other leaves of the ancestor's block, and only if all
//    of those inside it are empty the members of the sticking-out ones, with the reference's own predicate.
// tile_range[t] = the instances the blend walks: [first, end), empty for a gated leaf.  One thread per leaf.
// (Cost beside the blends of the other streams: 16 us per launch against 6 for the count check it replaces -- not its loads,
// which go out in one round, but its 94 VGPRs: five blend waves leave 32 of a SIMD's 512 free, so its waves start when a
// blend wave retires.  The job time does not see it, profiles/r03zo_*.)
constexpr int GATE_T = 64;    // one wave per block: its 94 VGPRs find a slot wherever ONE blend wave retires (a 256-thread block
                              // waits for a free slot on all four SIMDs of one CU at once: 44 - 83 us beside the blends)
__global__ __launch_bounds__(GATE_T) void k_tile_gate(Layout lay, Cam cam_val, const Cam* __restrict__ cam_dev,
                                                   const float* __restrict__ means3D, const float* __restrict__ cov9,
                                                   const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ inst_g,
                                                   uint32_t gmask, int T, uint32_t limit, uint2* __restrict__ tile_range,
                                                   uint32_t* __restrict__ tile_state, uint32_t* __restrict__ flag,
                                                   uint32_t* __restrict__ count_host, size_t cs) {
    tile_start = seg(tile_start, cs); inst_g = seg(inst_g, cs); tile_range = seg(tile_range, cs); tile_state = seg(tile_state, cs);
    const int t = blockIdx.x * GATE_T + threadIdx.x;
    if (t >= T) return;
    // the usual answer with ONE round of independent loads: the first leaf of an ancestor's block lies inside that ancestor,
    // so its having members settles the level (bit k of `vacant`: it has none)
    const int ix = t % lay.nx, iy = t / lay.nx;
    const uint32_t sticks = (uint32_t)lay.tile_stick[t];         // (depth 0: any readable table, no bit is looked at)
    uint32_t lo[8], hi[8], vacant = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // (levels beyond the tree read the leaf itself: every load is unconditional)
        const int sh = k < lay.depth ? lay.depth - k : 0, u = ((iy >> sh) << sh) * lay.nx + ((ix >> sh) << sh);
        lo[k] = tile_start[u]; hi[k] = tile_start[u + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) vacant |= ((k < lay.depth && lo[k] == hi[k]) ? 1u : 0u) << k;
    uint32_t first = tile_start[t], end = tile_start[t + 1];
    G2PC_PIN(vacant); G2PC_PIN(first);            // keep all loads in ONE round (else they sink behind the leaf's own count)
    const uint32_t cnt = end - first;
    uint32_t state = 0u;
    if (cnt && (sticks & vacant)) {               // (first: a leaf the queue never reaches is not split either)
        const Cam& cam = cam_dev ? *(const Cam*)((const char*)cam_dev + (size_t)blockIdx.y * sizeof(G2pcCameraJob)) : cam_val;
        for (int k = 0; k < lay.depth && !state; ++k) {
            if (!(((sticks & vacant) >> k) & 1u)) continue;
            const int sh = lay.depth - k, ax = ix >> sh, ay = iy >> sh, node = (1 << k) - 1;
            const int x0 = lay.inner_x[2 * (node + ax)], x1 = lay.inner_x[2 * (node + ax) + 1];
            const int y0 = lay.inner_y[2 * (node + ay)], y1 = lay.inner_y[2 * (node + ay) + 1];
            bool occupied = false;
            for (int pass = 0; pass < 2 && !occupied; ++pass)          // 0: leaves inside the node, 1: members of the others
                for (int by = ay << sh; by < ((ay + 1) << sh) && !occupied; ++by)
                    for (int bx = ax << sh; bx < ((ax + 1) << sh) && !occupied; ++bx) {
                        const int u = by * lay.nx + bx;
                        const uint32_t u0 = tile_start[u], u1 = tile_start[u + 1];
                        if (u1 == u0) continue;
                        const bool inside = !((((uint32_t)lay.tile_stick[u]) >> k) & 1u);
                        if (pass == 0) { occupied = inside; continue; }
                        if (inside) continue;
                        for (uint32_t m = u0; m < u1 && !occupied; ++m) {
                            float r[4];
                            occupied = py_rect(cam, means3D, cov9, (long)(inst_g[m] & gmask), r) && rect_in_node(r, x0, x1, y0, y1);
                        }
                    }
            if (!occupied) state = 2u | ((uint32_t)k << 8);
        }
    }
    // a tile of a child level that is no child of a split node (the level's layout is the PRODUCT of the child intervals) is
    // not part of the tree: it is never blended (its chunks are not in the work list) and must not report a load either
    bool in_tree = !lay.tile_mask || lay.tile_mask[t] != 0;
    // Static child pass (round 4): a pass over a parent layout leaves "this node holds a Gaussian" per tile in the job's `alive`
    // array; the child level's pass skips the children of nodes that held none (:311-314: never visited -- a Gaussian that
    // reaches only into the pixel a child extends beyond its odd-sized parent must not be blended there)
    uint8_t* alive = nullptr;
    if (cam_dev) {
        const G2pcCameraJob* jb = (const G2pcCameraJob*)((const char*)cam_dev + (size_t)blockIdx.y * sizeof(G2pcCameraJob));
        alive = (uint8_t*)(((unsigned long long)jb->alive_hi << 32) | jb->alive_lo);
    }
    if (alive) {
        if (lay.tile_parent && in_tree && !child_exists(lay.tile_parent, alive, t)) in_tree = false;
    }
    // ... and a node the size rule has not finished with (tile_force) is split whenever it holds a Gaussian (:319: `or` of the two)
    const uint8_t force = lay.tile_force ? lay.tile_force[t] : (uint8_t)0;
    const bool over = limit && cnt > limit, forced = force != 0 && cnt > 0;
    // "this node is split for this camera": its children exist (tile_mask does not matter here: the static pass A masks the
    // very nodes whose children follow)
    if (alive && !lay.tile_parent) alive[t] = (!state && (over || forced)) ? 1 : 0;
    // (a node whose children follow statically, force == 2, is masked out of pass A's tile_mask -- its chunks are not in the work
    // list --, so it is recognised by its force byte, not by in_tree: it gets state 3 and an empty range as include/g2pc.h says)
    const bool follows = force == 2 && !over && cnt > 0;   // its children come with the camera's static child pass
    if (!state && (over || forced) && (in_tree || follows)) {
        state = follows ? 3u : 1u;
        if (flag && over) atomicMax(flag, cnt);
        if (count_host && !follows) count_host[4 * blockIdx.y + 2] = cnt;   // pinned, through its device mapping: "some leaf of this camera"
    }
    if (!in_tree && lay.tile_parent) state = 4u;           // child of an empty node: never visited
    tile_range[t] = make_uint2(first, state ? first : end);
    tile_state[t] = state;
}

// Gaussians per node for a list of pixel rectangles (x0, y0, w, h) -- the reference's `tile_mask.sum()` of gauss_render.py:309
// for nodes that are not tiles of a layout (interior nodes of the quad-tree).  One block = 256 Gaussians x all nodes.
__global__ __launch_bounds__(RA_T) void k_node_counts(Cam cam, const float* __restrict__ means3D, const float* __restrict__ cov9,
                                                     long n, const int32_t* __restrict__ nodes, int m,
                                                     uint32_t* __restrict__ counts) {
    const long i = (long)blockIdx.x * RA_T + threadIdx.x;
    float r[4];
    const bool live = i < n && py_rect(cam, means3D, cov9, i, r);
    for (int j = 0; j < m; ++j) {
        const int x0 = nodes[4 * j], y0 = nodes[4 * j + 1], w = nodes[4 * j + 2], h = nodes[4 * j + 3];
        const bool in = live && rect_in_node(r, x0, x0 + w - 1, y0, y0 + h - 1);
        const unsigned long long b = __ballot(in ? 1 : 0);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[j], (uint32_t)__popcll(b));
    }
}

__global__ __launch_bounds__(RA_T) void k_adjacent_diff(const uint32_t* __restrict__ start, int T, uint32_t* __restrict__ out) {
    const int t = blockIdx.x * RA_T + threadIdx.x;
    if (t < T) out[t] = start[t + 1] - start[t];
}

// keys packed with an `ob`-bit tile field -> an `nb`-bit one (same camera slot, tile sequence and pixel)
__global__ __launch_bounds__(RA_T) void k_repack_keys(unsigned long long* __restrict__ best_key, long n, int ob, int nb) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = best_key[i];
    if ((key >> 32) == 0ull) return;
    const uint32_t order = ~(uint32_t)key;
    const uint32_t slot = order >> (12 + ob), seq = (order >> 12) & ((1u << ob) - 1u), pix = order & 0xFFFu;
    best_key[i] = (key & 0xFFFFFFFF00000000ull) | (uint32_t)~((slot << (12 + nb)) | (seq << 12) | pix);
}

// capacity-sized launches: the instance count stays on the device.  l_eff = L if it fits the buffers, else 0 (the
// camera is then skipped altogether and the host, which receives L asynchronously, renders it again with more room)
__global__ void k_resolve_count(const uint32_t* __restrict__ total, uint32_t capacity, uint32_t* __restrict__ l_eff,
                                uint32_t* __restrict__ count_host, const uint32_t* __restrict__ depth_overflow, size_t cs) {
    total = seg(total, cs); l_eff = seg(l_eff, cs); depth_overflow = seg(depth_overflow, cs);
    if (count_host) count_host += 4 * blockIdx.y;                                  // [camera][instances, unsorted, overloaded leaf, -]
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t unsorted = depth_overflow ? *depth_overflow : 0u;       // the depth bucket sort gave up: skip the camera
        l_eff[0] = (total[0] <= capacity && !unsorted) ? total[0] : 0u;
        // pinned host memory, through its device mapping ([2] is raised by k_tile_gate later in the sequence)
        if (count_host) { count_host[0] = total[0]; count_host[1] = unsorted; count_host[2] = 0u; }
    }
}
// the pinned host job -> device memory, by a kernel rather than a copy node (see g2pc_raster_camera_py)
// ... and, when the depth order comes from the bucket sort, the (zeroed) header its range partials are collected in
__global__ void k_fetch_job(const uint32_t* __restrict__ job_host, uint32_t* __restrict__ job_dev, BucketHdr* __restrict__ hdr,
                            size_t cs, BucketPlan plan) {
    const unsigned o = blockIdx.x * (unsigned)(sizeof(G2pcCameraJob) / 4);          // one block per camera of the batch
    if (job_host && threadIdx.x < sizeof(G2pcCameraJob) / 4) job_dev[o + threadIdx.x] = job_host[o + threadIdx.x];
    if (hdr) bucket_hdr_init((BucketHdr*)((char*)hdr + (size_t)blockIdx.x * cs), plan, threadIdx.x, blockDim.x);
}

// binding-side reductions (gaussian_pointcloud_rasterization/__init__.py:128-158): gather the colour of the arg-max
// pixel from the final image, strict-> running max (earliest camera wins ties), running SUM of the per-camera
// maxima, running min of the surface distance.
__global__ __launch_bounds__(RA_T) void k_update_cu(const unsigned long long* __restrict__ cam_key,
                                                   const uint32_t* __restrict__ cam_surf, long n, int W, int H,
                                                   const float* __restrict__ out_color,
                                                   float* __restrict__ max_contrib, float* __restrict__ total_contrib,
                                                   float* __restrict__ colours, float* __restrict__ min_surf,
                                                   int32_t* __restrict__ winner_cam, int32_t cam_index,
                                                   float* __restrict__ cur_contrib, int32_t* __restrict__ cur_pixels,
                                                   float* __restrict__ cur_surf) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = cam_key[i];
    float c = __uint_as_float((uint32_t)(key >> 32));
    uint32_t pix = c > 0.0f ? ~(uint32_t)key : 0u;                      // never blended: pixel 0, contribution 0
    // strictly larger wins; a tie goes to the EARLIER camera whatever the order the cameras are applied in (a camera that
    // outgrew its capacity is rendered again after later ones; multi-GPU ranks apply their shards independently)
    const float mc = max_contrib[i];
    if (c > mc || (c == mc && c > 0.0f && winner_cam && cam_index < winner_cam[i])) {
        const size_t plane = (size_t)W * H;
        max_contrib[i] = c;
        if (winner_cam) winner_cam[i] = cam_index;
        colours[3 * i + 0] = out_color[pix];
        colours[3 * i + 1] = out_color[plane + pix];
        colours[3 * i + 2] = out_color[2 * plane + pix];
    }
    total_contrib[i] += c;
    float sd = __uint_as_float(cam_surf[i]);
    if (sd < min_surf[i]) min_surf[i] = sd;
    if (cur_contrib) cur_contrib[i] = c;
    if (cur_pixels) cur_pixels[i] = (int32_t)pix;
    if (cur_surf) cur_surf[i] = sd;
}

// _C.mark_visible (rasterize_points.cu:147-166 -> checkFrustum -> in_frustum, auxiliary.h:151-176): z_view > 0.2
struct View16 { float m[16]; };
__global__ __launch_bounds__(RA_T) void k_mark_visible(View16 V, const float* __restrict__ means3D, long n,
                                                      uint8_t* __restrict__ present) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    present[i] = (V.m[2] * x + V.m[6] * y + V.m[10] * z + V.m[14]) > 0.2f ? 1 : 0;
}

__global__ __launch_bounds__(RA_T) void k_fill_u32(uint32_t* __restrict__ p, long n, uint32_t v) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i < n) p[i] = v;
}
// per-camera state of the native-semantics blend in one launch: packed (contribution, ~pixel) keys = 0, surface distance = FLT_MAX
__global__ __launch_bounds__(RA_T) void k_init_camera_state_cu(unsigned long long* __restrict__ cam_key, uint32_t* __restrict__ cam_surf,
                                                              long n) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i < n) { cam_key[i] = 0ull; cam_surf[i] = 0x7F7FFFFFu; }
}

static uint32_t* g_chunk_work = nullptr;      // diagnostics hook (g2pc_raster_debug_chunk_work)

static Cam to_cam(const G2pcCamera* c) {
    Cam k;
    for (int i = 0; i < 16; ++i) { k.V[i] = c->view[i]; k.P[i] = c->proj[i]; }
    k.tan_fovx = c->tan_fovx; k.tan_fovy = c->tan_fovy; k.focal_x = c->focal_x; k.focal_y = c->focal_y;
    k.W = c->width; k.H = c->height;
    k.bg[0] = c->bg[0]; k.bg[1] = c->bg[1]; k.bg[2] = c->bg[2];
    k.lim_x = c->lim_x; k.lim_y = c->lim_y;
    return k;
}
static int g_walk_cap = 0;        // diagnostic only, see Layout::walk_cap
static Layout to_layout(const G2pcTileLayout* l) {
    Layout k;
    k.walk_cap = g_walk_cap;
    k.nx = l->nx; k.ny = l->ny; k.num_chunks = l->num_chunks; k.seq_bits = l->seq_bits ? l->seq_bits : 12; k.xs = l->xs; k.ws = l->ws; k.ys = l->ys; k.hs = l->hs;
    k.tile_seq = l->tile_seq; k.seq_tile = l->seq_tile; k.tile_pix_off = l->tile_pix_off;
    k.seq_base = l->seq_count ? l->seq_base : 0; k.seq_count = l->seq_count ? l->seq_count : l->nx * l->ny;
    k.tile_mask = l->tile_mask;
    k.tile_force = l->tile_force;
    k.tile_parent = l->tile_parent;
    const bool tree = l->depth > 0 && l->inner_x && l->inner_y && l->tile_stick && l->nx == (1 << l->depth) && l->ny == (1 << l->depth);
    k.depth = tree ? l->depth : 0; k.inner_x = l->inner_x; k.inner_y = l->inner_y; k.tile_stick = tree ? l->tile_stick : nullptr;
    return k;
}
// native-semantics tile grids beyond 256 x 256 (images beyond 4 096 pixels a side): tile rectangles take two words per Gaussian
static bool cu_wide_grid(int gx, int gy) { return gx > 256 || gy > 256; }
static int bits_for_tiles(unsigned t) { int b = 1; while ((1u << b) < t && b < 31) ++b; return b; }
// packed visibility keys: the tile-sequence field is seq_bits wide (12 .. 14), the camera slot gets the 20 - seq_bits above it
static bool layout_keys_ok(const G2pcTileLayout* l) {
    const int sb = l->seq_bits ? l->seq_bits : 12;
    const long top = l->seq_count ? (long)l->seq_base + l->seq_count : (long)l->nx * l->ny;     // largest sequence number + 1
    return sb >= 12 && sb <= 14 && l->seq_base >= 0 && l->seq_count >= 0 && top <= (1l << sb);
}
static uint32_t max_camera_slot(const G2pcTileLayout* l) { return (1u << (20 - (l->seq_bits ? l->seq_bits : 12))) - 1u; }

// ---- host side of the PY path, shared by the two-call API (count read back by the host) and the single-call,
// capture-safe API (count stays on the device, launch geometry fixed by a capacity) -------------------------------
struct PyFrontBuffers { float4* rec; uint32_t *rect, *sorted_idx, *offsets; };      // rec: 4 x float4 per Gaussian

static size_t py_front_ws(long n) {
    return align_up((size_t)n * 4) * 6 + sort_workspace(n) + scan_workspace(n) + bucket_sort_workspace(n) + 4096;
}
static int g_blend_variant = 1;               // 2 sub-blocks per chunk: 2 = two-wave dual-list kernel (k_blend_py_2w, one wave per sub-block), 1 = dual-list kernel (k_blend_py_dl), 0 = packed kernel (k_blend_py_pk)
static int g_depth_bucket_sort = 1;           // captured camera path: 1 = bucket sort of the depth keys, 0 = radix (g2pc_set_depth_sort)
#ifndef G2PC_PREPROCESS_MULTI
#define G2PC_PREPROCESS_MULTI 1
#endif
static const int g_preprocess_multi = G2PC_PREPROCESS_MULTI;   // build-time A/B switch: one thread per Gaussian for all cameras of a batch
#ifndef G2PC_FUSED_EMIT
#define G2PC_FUSED_EMIT 1
#endif
static const int g_fused_emit = G2PC_FUSED_EMIT;   // build-time A/B switch: 0 = scan + k_duplicate + k_resolve_count as until round 4
// Packed tile-sort instances: when the tile id and the Gaussian index share one 32-bit word (tile << gshift | index) the
// stable sort by tile moves keys only -- half the traffic of the two passes -- and the blend masks the index out.
// Returns gshift (0: they do not fit, separate arrays as before).
static int packed_instance_shift(long n, int T) {
    int gbits = 1;
    while (((long)1 << gbits) < n) ++gbits;
    return (gbits + bits_for_tiles((unsigned)T) <= 32) ? gbits : 0;
}
static size_t py_back_ws(long L, int T) {
    return align_up((size_t)(L + 1) * 4) * 6 + sort_workspace(L) + scan_workspace(T + 1) + align_up((size_t)(T + 2) * 4) +
           align_up((size_t)(T + 1) * 8) + align_up((size_t)(T + 1) * 4) + 4096 + 256;
}
// the arena of the back half (shared by py_back and g2pc_raster_tile_states)
struct PyBackArena {
    uint32_t *inst_tile, *inst_g, *tile_sorted, *g_sorted, *tile_tmp, *g_tmp, *tile_start, *tile_state;
    uint2* tile_range;
    char* sort_ws;
    size_t sort_bytes;
    bool ok;
    PyBackArena(void* ws, size_t ws_bytes, long L, int T) {
        Arena ar(ws, ws_bytes);
        inst_tile = ar.get<uint32_t>((size_t)L + 1);
        inst_g = ar.get<uint32_t>((size_t)L + 1);
        tile_sorted = ar.get<uint32_t>((size_t)L + 1);
        g_sorted = ar.get<uint32_t>((size_t)L + 1);
        tile_tmp = ar.get<uint32_t>((size_t)L + 1);       // ping-pong scratch of the multi-pass sort: must NOT
        g_tmp = ar.get<uint32_t>((size_t)L + 1);          // alias its input (pass 0 writes here when #passes is even)
        tile_start = ar.get<uint32_t>((size_t)T + 2);
        tile_range = ar.get<uint2>((size_t)T + 1);        // k_tile_gate: what the blend walks per tile
        tile_state = ar.get<uint32_t>((size_t)T + 1);     // k_tile_gate: 0 blended, 1 overloaded, 2 | level << 8 under an empty node
        sort_bytes = sort_workspace(L);
        sort_ws = ar.get<char>(sort_bytes);
        ok = ar.ok();
    }
};

// depth_overflow != nullptr: the depth order comes from the bucket sort (prims.hip) and *depth_overflow points at its
// overflow word afterwards (non-zero = NOT sorted: the caller must discard the camera and repeat it with the radix path)
// emit != nullptr (captured path, bucket sort): the depth sort's last kernel writes the (tile, Gaussian) instances itself and
// the instance count is settled inside the sort (BucketEmit, prims.hip) -- no tiles-touched scan here, no k_duplicate and no
// k_resolve_count afterwards; fb.sorted_idx / fb.offsets are then NOT written.
static int py_front(const Cam& cam_val, const Cam* cam_dev, const G2pcTileLayout* layout, const float* means3D,
                    const float* cov9, const float* opacity, const float* colours, long n, const PyFrontBuffers& fb,
                    void* ws, size_t ws_bytes, hipStream_t s, uint32_t** depth_overflow = nullptr, Batch bt = Batch(),
                    const G2pcCameraJob* jobs_host = nullptr, G2pcCameraJob* jobs_dev = nullptr, BucketEmit* emit = nullptr) {
    Arena ar(ws, ws_bytes);
    uint32_t* key_rev = ar.get<uint32_t>((size_t)n);
    uint32_t* idx_rev = ar.get<uint32_t>((size_t)n);
    uint32_t* key_sorted = ar.get<uint32_t>((size_t)n);
    uint32_t* ktmp = ar.get<uint32_t>((size_t)n);
    uint32_t* vtmp = ar.get<uint32_t>((size_t)n);
    uint32_t* touched = ar.get<uint32_t>((size_t)n);
    size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace(n);
    char* sort_ws = ar.get<char>(sort_bytes);
    char* scan_ws = ar.get<char>(scan_bytes);
    const size_t bucket_bytes = depth_overflow ? bucket_sort_workspace(n) : 0;
    char* bucket_ws = ar.get<char>(bucket_bytes);
    if (!ar.ok()) { set_error("raster_front_py", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    // device-resident cameras whose depth order comes from the bucket sort: the sort's range pass is folded into the
    // preprocess (header zeroed by the job fetch) and its values are the reversed positions themselves (no index array)
    const bool fold = cam_dev && depth_overflow;
    BucketHdr* hdr = fold ? bucket_sort_header(bucket_ws) : nullptr;
    const BucketPlan plan = bucket_plan(n);
    if (cam_dev && (jobs_host || hdr))
        hipLaunchKernelGGL(k_fetch_job, dim3((unsigned)bt.n), dim3(64), 0, s, (const uint32_t*)jobs_host, (uint32_t*)jobs_dev, hdr, bt.cs, plan);
    if (cam_dev && bt.n > 1 && g_preprocess_multi)
        hipLaunchKernelGGL((k_preprocess_py<true, true>), dim3(cdiv(n, g_head_threads), 1u), dim3(g_head_threads), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, fold ? (uint32_t*)nullptr : idx_rev, touched, colours, fb.rec, fb.rect,
                           bt.cs, hdr, plan.nminmax, bt.n);
    else if (cam_dev)
        hipLaunchKernelGGL((k_preprocess_py<true, false>), dim3(cdiv(n, g_head_threads), (unsigned)bt.n), dim3(g_head_threads), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, fold ? (uint32_t*)nullptr : idx_rev, touched, colours, fb.rec, fb.rect,
                           bt.cs, hdr, plan.nminmax, 1);
    else
        hipLaunchKernelGGL((k_preprocess_py<false, false>), dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, idx_rev, touched, colours, fb.rec, fb.rect, (size_t)0,
                           (BucketHdr*)nullptr, 1u, 1);
    for (int k = 0; k < g_extra_launches; ++k) hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, (uint32_t*)nullptr);
    if (emit && !(fold && depth_overflow)) { set_error("raster_front_py", "fused emission without the folded bucket sort"); return G2PC_ERR_ARG; }
    if (emit) { emit->weight = touched; emit->rect = fb.rect; }
    int rc = depth_overflow ? bucket_sort_u32(key_rev, fold ? nullptr : idx_rev, fb.sorted_idx, nullptr, n, bucket_ws, bucket_bytes,
                                              depth_overflow, s, bt, fold, fold, emit)
                            : sort_pairs_u32(key_rev, idx_rev, key_sorted, fb.sorted_idx, ktmp, vtmp, n, 0, 32, sort_ws, sort_bytes, s, nullptr, bt);
    if (rc) return rc;
    if (emit) return G2PC_OK;
    // exclusive scan of the tiles touched, taken in depth order (the gather rides in the scan's first kernel)
    return scan_exclusive_u32(touched, fb.offsets, n, scan_ws, scan_bytes, s, fb.sorted_idx, bt);
}

struct PyBlendArgs {                  // by value ...                      ... or device resident (job != nullptr)
    uint32_t camera_slot; float t_floor; float bg; const G2pcCameraJob* job;
};
struct PyScene {                      // what k_tile_gate re-derives a member's rectangle from (means3D == nullptr: it does not look)
    Cam cam_val; const Cam* cam_dev; const float* means3D; const float* cov9; uint32_t* count_host;
};

// L: the instance count, or (l_eff != nullptr) the capacity of the buffers with the count in device memory
static int py_back(const G2pcTileLayout* layout, long n, long L, const uint32_t* l_eff,
                   const PyBlendArgs& ba, int W, int H, const PyFrontBuffers& fb, unsigned long long* best_key,
                   float* colours_out, float* tilebuf, float* image, int phases, uint32_t max_per_tile,
                   uint32_t* overflow_flag, void* ws, size_t ws_bytes, hipStream_t s, Batch bt = Batch(),
                   const PyScene& sc = PyScene{}, bool emitted = false) {      // emitted: the instances are there already (BucketEmit)
    const int T = layout->nx * layout->ny;
    PyBackArena A(ws, ws_bytes, L, T);
    if (!A.ok) { set_error("raster_back_py", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    uint32_t *inst_tile = A.inst_tile, *inst_g = A.inst_g, *tile_sorted = A.tile_sorted, *g_sorted = A.g_sorted;
    uint32_t *tile_tmp = A.tile_tmp, *g_tmp = A.g_tmp, *tile_start = A.tile_start;
    char* sort_ws = A.sort_ws;
    const size_t sort_bytes = A.sort_bytes;
    Layout lay = to_layout(layout);
    const int gshift = packed_instance_shift(n, T);
    const uint32_t gmask = gshift ? ((1u << gshift) - 1u) : 0xFFFFFFFFu;
    const uint32_t* blend_list = gshift ? tile_sorted : g_sorted;
    if (phases & 1) {
        if (L > 0) {
            if (!emitted)
                hipLaunchKernelGGL(k_duplicate<false>, dim3(cdiv(n, g_head_threads), (unsigned)bt.n), dim3(g_head_threads), 0, s, fb.sorted_idx, fb.offsets, fb.rect, n, lay.nx,
                                   inst_tile, inst_g, l_eff, gshift, bt.cs, sc.cam_dev ? lay.tile_parent : (const int32_t*)nullptr,
                                   (const G2pcCameraJob*)sc.cam_dev);
            int rc = gshift ? sort_pairs_u32(inst_tile, nullptr, tile_sorted, nullptr, tile_tmp, nullptr, L, gshift,
                                             gshift + bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s, l_eff, bt)
                            : sort_pairs_u32(inst_tile, inst_g, tile_sorted, g_sorted, tile_tmp, g_tmp, L, 0,
                                             bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s, l_eff, bt);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(L + 1, g_head_threads), (unsigned)bt.n), dim3(g_head_threads), 0, s, tile_sorted, L, T, tile_start, l_eff, gshift, bt.cs);
        Layout glay = lay;
        // no scene / no tree tables: leaves under empty nodes are not looked for (depth 0; the kernel's loads stay unconditional)
        if (!sc.means3D || !lay.tile_stick) { glay.depth = 0; glay.tile_stick = lay.tile_seq; }
        hipLaunchKernelGGL(k_tile_gate, dim3(cdiv(T, GATE_T), (unsigned)bt.n), dim3(GATE_T), 0, s, glay, sc.cam_val, sc.cam_dev, sc.means3D,
                           sc.cov9, tile_start, blend_list, gmask, T, max_per_tile, A.tile_range, A.tile_state, overflow_flag,
                           sc.count_host, bt.cs);
    }
    if ((phases & 2) && layout->num_chunks > 0) {
        const unsigned chunks_y = layout->num_chunks < 32768 ? (unsigned)layout->num_chunks : 32768u;   // grid.y is 16 bits wide
#define G2PC_BLEND(...)                                                                                                 \
    hipLaunchKernelGGL((__VA_ARGS__), dim3((unsigned)bt.n, chunks_y, cdiv(layout->num_chunks, chunks_y)), dim3(BL_T), 0, s, lay, layout->chunk_tile, \
                       layout->chunk_pix0, A.tile_range, blend_list, gmask, (const float4*)fb.rec, best_key,           \
                       ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, g_chunk_work, ba.job, bt.cs)
        switch (layout->chunk_subblocks) {
            case 1: G2PC_BLEND(k_blend_py<1, 4>); break;
            case 2: {
                // t_floor == 0 is the to-the-letter mode: it takes the kernel that evaluates the exponent in the reference's
                // operation order (k_blend_py_pk); the dual-list kernel's expanded exponent differs by up to ~2e-5 relative
                // in alpha.  A captured camera reads t_floor from its device job: the caller says so with phase bit 8.
                const bool exact = ba.job ? ((phases & 8) != 0) : (ba.t_floor == 0.0f);
                if (g_blend_variant == 6 && !exact) G2PC_BLEND(k_blend_py_dl<2>);
                else if (g_blend_variant == 4 && !exact) G2PC_BLEND(k_blend_py_sg<4>);
                else if (g_blend_variant == 5 && !exact) G2PC_BLEND(k_blend_py_sg<2>);
                else if (g_blend_variant == 3 && !exact)
                    hipLaunchKernelGGL((k_blend_py_2w<2>), dim3((unsigned)bt.n, chunks_y, cdiv(layout->num_chunks, chunks_y)), dim3(2 * BL_T), 0, s,
                                       lay, layout->chunk_tile, layout->chunk_pix0, A.tile_range, blend_list, gmask, (const float4*)fb.rec,
                                       best_key, ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, g_chunk_work, ba.job, bt.cs);
                else if (g_blend_variant == 2 && !exact)
                    hipLaunchKernelGGL((k_blend_py_2w<4>), dim3((unsigned)bt.n, chunks_y, cdiv(layout->num_chunks, chunks_y)), dim3(2 * BL_T), 0, s,
                                       lay, layout->chunk_tile, layout->chunk_pix0, A.tile_range, blend_list, gmask, (const float4*)fb.rec,
                                       best_key, ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, g_chunk_work, ba.job, bt.cs);
                else if (g_blend_variant == 1 && !exact) G2PC_BLEND(k_blend_py_dl<4>);
                else G2PC_BLEND(k_blend_py_pk<4>);
                break;
            }
            case 4: G2PC_BLEND(k_blend_py<4, 1>); break;
            default: set_error("raster_back_py", "chunk_subblocks must be 1, 2 or 4"); return G2PC_ERR_ARG;
        }
#undef G2PC_BLEND
    }
    if (phases & 4) {
        hipLaunchKernelGGL(k_update_colours_py, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, lay, best_key, n, ba.camera_slot,
                           tilebuf, colours_out);
        if (image)
            hipLaunchKernelGGL(k_assemble_image_py, dim3(cdiv((long)W * H, RA_T)), dim3(RA_T), 0, s, lay, W, H, tilebuf, image);
    }
    return G2PC_OK;
}

}  // namespace g2pc

extern "C" {

size_t g2pc_raster_front_workspace(int64_t n) { return g2pc::py_front_ws(n); }

// Front half of one camera: preprocess -> depth sort -> tiles-touched scan.  Leaves sorted_idx u32[n] and
// offsets u32[n+1] (offsets[n] = L, the number of (tile, Gaussian) instances) for the back half.
int g2pc_raster_front_py(const G2pcCamera* cam, const G2pcTileLayout* layout, const float* means3D, const float* cov9,
                         const float* opacity, const float* colours, int64_t n, float* rec, uint32_t* rect,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && layout && means3D && cov9 && opacity && colours && rec && rect && sorted_idx && offsets && ws && n > 0,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(layout->nx <= 256 && layout->ny <= 256, G2PC_ERR_UNSUPPORTED, "more than 256 tile intervals per axis");
    hipStream_t s = (hipStream_t)stream;
    PyFrontBuffers fb{(float4*)rec, rect, sorted_idx, offsets};
    int rc = py_front(to_cam(cam), nullptr, layout, means3D, cov9, opacity, colours, (long)n, fb, ws, ws_bytes, s);
    if (rc) return rc;
    if (count_host) hipMemcpyAsync(count_host, offsets + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    return check_launch("g2pc_raster_front_py");
}

size_t g2pc_raster_back_workspace(int64_t num_instances, int32_t num_tiles) {
    return g2pc::py_back_ws((long)num_instances, num_tiles);
}

// Back half: duplicate -> stable sort by tile id -> tile ranges -> blend + visibility -> colour update.
int g2pc_raster_back_py(const G2pcCamera* cam, const G2pcTileLayout* layout, int64_t n, int64_t num_instances,
                        const float* rec, const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                        const float* means3D, const float* cov9, uint32_t camera_slot, float t_floor,
                        unsigned long long* best_key, float* colours_out, float* tilebuf, float* image, int phases,
                        uint32_t max_per_tile, uint32_t* overflow_flag, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && layout && rec && rect && sorted_idx && offsets && best_key && colours_out && tilebuf && ws && n > 0,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(layout_keys_ok(layout), G2PC_ERR_UNSUPPORTED, "seq_bits must be 12 .. 14 and hold every tile (at most 16384 tiles)");
    G2PC_REQUIRE(camera_slot >= 1 && camera_slot <= max_camera_slot(layout), G2PC_ERR_ARG, "camera_slot must be in [1, (1 << (20 - seq_bits)) - 1]");
    PyFrontBuffers fb{(float4*)rec, (uint32_t*)rect, (uint32_t*)sorted_idx, (uint32_t*)offsets};
    PyBlendArgs ba{camera_slot, t_floor, cam->bg[0], nullptr};
    PyScene scene{to_cam(cam), nullptr, cov9 ? means3D : nullptr, cov9, nullptr};
    int rc = py_back(layout, (long)n, (long)num_instances, nullptr, ba, cam->width, cam->height, fb, best_key,
                     colours_out, tilebuf, image, phases, max_per_tile, overflow_flag, ws, ws_bytes, (hipStream_t)stream,
                     Batch(), scene);
    if (rc) return rc;
    return check_launch("g2pc_raster_back_py");
}

size_t g2pc_raster_camera_workspace(int64_t n, int64_t capacity, int32_t num_tiles) {
    using namespace g2pc;
    // a multiple of 256: batched launches place one such arena per camera back to back (g2pc_raster_cameras_py)
    return align_up(align_up((size_t)n * 64) + align_up((size_t)n * 4) * 2 + align_up((size_t)(n + 1) * 4) + 256 +
                    py_front_ws((long)n) + py_back_ws((long)capacity, num_tiles) + 4096);
}

// One camera up to and including the blend, without any host round trip: the camera, its slot and the transmittance
// floor are read from device memory (job_dev) and the instance count never leaves the device, so the launch sequence
// depends on (n, capacity, layout) only and can be captured once into a hipGraph and replayed for every camera.
int g2pc_raster_cameras_py(const G2pcCameraJob* jobs_dev, const G2pcCameraJob* jobs_host, int32_t batch,
                           const G2pcTileLayout* layout, const float* means3D, const float* cov9, const float* opacity,
                           const float* colours, int64_t n, int64_t capacity, unsigned long long* best_key, float* tilebuf,
                           uint32_t* count_host, uint32_t max_per_tile, uint32_t* overflow_flag, int phases, void* ws,
                           size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(jobs_dev && layout && means3D && cov9 && opacity && colours && best_key && tilebuf && ws && n > 0 &&
                     capacity > 0,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(batch >= 1 && batch <= G2PC_MAX_CAMERA_BATCH, G2PC_ERR_ARG, "batch must be in [1, G2PC_MAX_CAMERA_BATCH]");
    G2PC_REQUIRE(layout->nx <= 256 && layout->ny <= 256, G2PC_ERR_UNSUPPORTED, "more than 256 tile intervals per axis");
    G2PC_REQUIRE(layout_keys_ok(layout), G2PC_ERR_UNSUPPORTED, "seq_bits must be 12 .. 14 and hold every tile (at most 16384 tiles)");
    G2PC_REQUIRE(capacity < (1ll << 31), G2PC_ERR_ARG, "capacity must be below 2^31 instances");
    static_assert(sizeof(Cam) == sizeof(G2pcCamera), "Cam mirrors G2pcCamera");
    hipStream_t s = (hipStream_t)stream;
    const int T = layout->nx * layout->ny;
    // one arena per camera, all with the same internal layout, `cs` bytes apart: the pointers below are camera 0's and
    // every kernel moves them by blockIdx.y * cs (g2pc_internal.h: seg)
    Batch bt;
    bt.n = batch;
    bt.cs = g2pc_raster_camera_workspace(n, capacity, T);
    G2PC_REQUIRE(ws_bytes >= bt.cs * (size_t)batch, G2PC_ERR_WORKSPACE, "workspace too small");
    Arena ar(ws, bt.cs);
    PyFrontBuffers fb;
    fb.rec = ar.get<float4>((size_t)n * 4);
    fb.rect = ar.get<uint32_t>((size_t)n);
    fb.sorted_idx = ar.get<uint32_t>((size_t)n);
    fb.offsets = ar.get<uint32_t>((size_t)n + 1);
    uint32_t* l_eff = ar.get<uint32_t>(1);
    const size_t front_bytes = py_front_ws((long)n), back_bytes = py_back_ws((long)capacity, T);
    char* front_ws = ar.get<char>(front_bytes);
    char* back_ws = ar.get<char>(back_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    int rc;
    // Bucket-sorted cameras emit their instances from inside the sort (BucketEmit): the depth sort's last kernel, one wave per
    // depth bucket, writes what k_duplicate would write, the count is settled by the sort's own scan kernel.
    const bool bucket = g_depth_bucket_sort && bucket_sort_pays((long)n);
    const bool fused = bucket && g_fused_emit && bucket_emit_supported((long)n);
    if (phases & 1) {
        // Both hand-overs with the host go through kernels that touch the PINNED buffers via their device mapping, not
        // through copy nodes: a graph whose first node is a host-to-device copy replayed with ~0.1 ms of extra latency
        // per camera for the lifetime of the first buffers a process pinned (25.9 -> 29 ms per 50-camera job).
        uint32_t* depth_overflow = nullptr;
        BucketEmit em{};
        if (fused) {
            PyBackArena A(back_ws, back_bytes, (long)capacity, T);
            G2PC_REQUIRE(A.ok, G2PC_ERR_WORKSPACE, "workspace too small");
            em.inst_tile = A.inst_tile; em.inst_g = A.inst_g;
            em.gshift = packed_instance_shift((long)n, T); em.nx = layout->nx;
            em.capacity = (uint32_t)capacity; em.l_eff = l_eff; em.count_host = count_host;
            em.tile_parent = layout->tile_parent; em.jobs = jobs_dev;
        }
        rc = py_front(Cam{}, (const Cam*)&jobs_dev->cam, layout, means3D, cov9, opacity, colours, (long)n, fb, front_ws,
                      front_bytes, s, bucket ? &depth_overflow : nullptr, bt, jobs_host,
                      (G2pcCameraJob*)jobs_dev, fused ? &em : nullptr);
        if (rc) return rc;
        if (!fused)
            hipLaunchKernelGGL(k_resolve_count, dim3(1, (unsigned)batch), dim3(64), 0, s, fb.offsets + n, (uint32_t)capacity, l_eff,
                               count_host, (const uint32_t*)depth_overflow, bt.cs);
    }
    PyBlendArgs ba{0u, 0.0f, 0.0f, jobs_dev};
    PyScene scene{Cam{}, (const Cam*)&jobs_dev->cam, means3D, cov9, count_host};
    rc = py_back(layout, (long)n, (long)capacity, l_eff, ba, 0, 0, fb, best_key, nullptr, tilebuf, nullptr,
                 phases & (3 | 8), max_per_tile, overflow_flag, back_ws, back_bytes, s, bt, scene, fused);
    if (rc) return rc;
    return check_launch("g2pc_raster_cameras_py");
}

int g2pc_raster_camera_py(const G2pcCameraJob* job_dev, const G2pcCameraJob* job_host, const G2pcTileLayout* layout,
                          const float* means3D, const float* cov9, const float* opacity, const float* colours, int64_t n,
                          int64_t capacity, unsigned long long* best_key, float* tilebuf, uint32_t* count_host,
                          uint32_t max_per_tile, uint32_t* overflow_flag, int phases, void* ws, size_t ws_bytes,
                          void* stream) {
    return g2pc_raster_cameras_py(job_dev, job_host, 1, layout, means3D, cov9, opacity, colours, n, capacity, best_key, tilebuf,
                                  count_host, max_per_tile, overflow_flag, phases, ws, ws_bytes, stream);
}

/* colour update of a camera rendered with g2pc_raster_camera_py (to be issued in camera order, see g2pc_raster_back_py) */
int g2pc_raster_camera_update_py(const G2pcTileLayout* layout, int64_t n, uint32_t camera_slot,
                                 const unsigned long long* best_key, const float* tilebuf, float* colours_out, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(layout && best_key && tilebuf && colours_out && n > 0, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(layout_keys_ok(layout), G2PC_ERR_UNSUPPORTED, "seq_bits must be 12 .. 14 and hold every tile");
    G2PC_REQUIRE(camera_slot >= 1 && camera_slot <= max_camera_slot(layout), G2PC_ERR_ARG, "camera_slot must be in [1, (1 << (20 - seq_bits)) - 1]");
    hipLaunchKernelGGL(k_update_colours_py, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, to_layout(layout),
                       best_key, (long)n, camera_slot, tilebuf, colours_out);
    return check_launch("g2pc_raster_camera_update_py");
}

int g2pc_raster_resolve_colours_py(const G2pcTileLayout* layout, int64_t n, const unsigned long long* best_key,
                                   const unsigned long long* tilebufs, float* colours_out, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(layout && best_key && tilebufs && colours_out && n > 0, G2PC_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(k_resolve_colours_py, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, to_layout(layout),
                       best_key, (long)n, tilebufs, colours_out);
    return check_launch("g2pc_raster_resolve_colours_py");
}

/* what k_tile_gate decided for the tiles of the camera last binned in `ws` (g2pc_raster_back_py phase 1) */
int g2pc_raster_tile_states(const void* ws, size_t ws_bytes, int64_t num_instances, int32_t num_tiles, uint32_t* counts,
                            uint32_t* states, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(ws && num_tiles > 0 && num_instances >= 0 && (counts || states), G2PC_ERR_ARG, "bad arguments");
    PyBackArena A(const_cast<void*>(ws), ws_bytes, (long)num_instances, num_tiles);
    G2PC_REQUIRE(A.ok, G2PC_ERR_WORKSPACE, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (counts)
        hipLaunchKernelGGL(k_adjacent_diff, dim3(cdiv(num_tiles, RA_T)), dim3(RA_T), 0, s, (const uint32_t*)A.tile_start, num_tiles, counts);
    if (states) hipMemcpyAsync(states, A.tile_state, (size_t)num_tiles * 4, hipMemcpyDeviceToDevice, s);
    return check_launch("g2pc_raster_tile_states");
}

int g2pc_raster_node_counts(const G2pcCamera* cam, const float* means3D, const float* cov9, int64_t n, const int32_t* nodes,
                            int32_t num_nodes, uint32_t* counts, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && means3D && cov9 && nodes && counts && n > 0 && num_nodes > 0, G2PC_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(counts, 0, (size_t)num_nodes * 4, s);
    hipLaunchKernelGGL(k_node_counts, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, to_cam(cam), means3D, cov9, (long)n, nodes, num_nodes, counts);
    return check_launch("g2pc_raster_node_counts");
}

int g2pc_raster_repack_keys(unsigned long long* best_key, int64_t n, int32_t old_seq_bits, int32_t new_seq_bits, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(best_key && n > 0, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE(old_seq_bits >= 12 && new_seq_bits >= old_seq_bits && new_seq_bits <= 14, G2PC_ERR_ARG, "seq_bits must grow within 12 .. 14");
    if (new_seq_bits != old_seq_bits)
        hipLaunchKernelGGL(k_repack_keys, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, best_key, (long)n, old_seq_bits, new_seq_bits);
    return check_launch("g2pc_raster_repack_keys");
}

/* diagnostics: see g2pc.h */
int g2pc_raster_debug_chunk_work(uint32_t* buf) { g2pc::g_chunk_work = buf; return G2PC_OK; }
int g2pc_debug_set_extra_launches(int n) { g2pc::g_extra_launches = n > 0 ? n : 0; return G2PC_OK; }
int g2pc_debug_set_head_threads(int threads) {
    if (threads != 64 && threads != 128 && threads != 256) return G2PC_ERR_ARG;
    g2pc::g_head_threads = threads;
    return G2PC_OK;
}
int g2pc_debug_set_walk_cap(int batches) { g2pc::g_walk_cap = batches > 0 ? batches : 0; return G2PC_OK; }

/* depth order of the capture-safe camera call: 1 = range-normalised bucket sort + in-LDS bitonic (default), 0 = 4-pass
 * radix.  Identical results; a camera whose depths pile up (bucket overflow) is skipped and reported through
 * count_host[1] -- the caller repeats it with g2pc_raster_front_py / _back_py, which always use the radix sort. */
int g2pc_set_blend_variant(int variant) { g2pc::g_blend_variant = variant; return G2PC_OK; }
int g2pc_set_depth_sort(int bucket) { g2pc::g_depth_bucket_sort = bucket ? 1 : 0; return G2PC_OK; }

int g2pc_raster_rebase_keys(unsigned long long* best_key, int64_t n, void* stream) {
    using namespace g2pc;
    if (n <= 0) return G2PC_OK;
    hipLaunchKernelGGL(k_rebase_keys, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, best_key, (long)n);
    return check_launch("g2pc_raster_rebase_keys");
}

int g2pc_raster_key_owner(const unsigned long long* local_key, const unsigned long long* global_key, int64_t n,
                          int32_t rank, int32_t* owner, void* stream) {
    using namespace g2pc;
    if (n <= 0) return G2PC_OK;
    G2PC_REQUIRE(local_key && global_key && owner, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_key_owner, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, local_key, global_key, (long)n,
                       (int)rank, owner);
    return check_launch("g2pc_raster_key_owner");
}

int g2pc_raster_keep_winner_colours(const int32_t* owner, int64_t n, int32_t rank, float* colours, void* stream) {
    using namespace g2pc;
    if (n <= 0) return G2PC_OK;
    G2PC_REQUIRE(owner && colours, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_keep_winner_colours, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, owner, (long)n,
                       (int)rank, colours);
    return check_launch("g2pc_raster_keep_winner_colours");
}

int g2pc_mark_visible(const float* means3D, int64_t n, const float* viewmatrix, uint8_t* present, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(means3D && viewmatrix && present && n > 0, G2PC_ERR_ARG, "bad arguments");
    View16 V;
    for (int i = 0; i < 16; ++i) V.m[i] = viewmatrix[i];
    hipLaunchKernelGGL(k_mark_visible, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, V, means3D, (long)n, present);
    return check_launch("g2pc_mark_visible");
}

int g2pc_raster_contributions(const unsigned long long* best_key, int64_t n, float* out, void* stream) {
    using namespace g2pc;
    if (n <= 0) return G2PC_OK;
    hipLaunchKernelGGL(k_contributions, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, best_key, (long)n, out);
    return check_launch("g2pc_raster_contributions");
}
}

extern "C" {
// CU semantics, front half: preprocess (+SH) -> depth sort (ascending index on ties) -> tiles-touched scan.
int g2pc_raster_front_cu(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                         const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                         const float* campos, int64_t n, float* rec, uint32_t* rect, int32_t* radii,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && means3D && cov6 && opacity && campos && rec && rect && radii && sorted_idx && offsets && ws && n > 0,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE((colours_precomp != nullptr) != (shs != nullptr), G2PC_ERR_ARG,
                 "provide exactly one of precomputed colours or SHs");       // __init__.py:42-43
    G2PC_REQUIRE(!shs || (sh_degree >= 0 && sh_degree <= 3 && sh_coeffs >= (sh_degree + 1) * (sh_degree + 1)), G2PC_ERR_ARG,
                 "SH degree / coefficient count mismatch");
    const int gx = (cam->width + 15) / 16, gy = (cam->height + 15) / 16;
    G2PC_REQUIRE(gx <= 65535 && gy <= 65535, G2PC_ERR_UNSUPPORTED, "image larger than 1048560 pixels per side");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    uint32_t* key = ar.get<uint32_t>((size_t)n);
    uint32_t* idx = ar.get<uint32_t>((size_t)n);
    uint32_t* key_sorted = ar.get<uint32_t>((size_t)n);
    uint32_t* ktmp = ar.get<uint32_t>((size_t)n);
    uint32_t* vtmp = ar.get<uint32_t>((size_t)n);
    uint32_t* touched = ar.get<uint32_t>((size_t)n);
    size_t sort_bytes = sort_workspace(n), scan_bytes = scan_workspace(n);
    char* sort_ws = ar.get<char>(sort_bytes);
    char* scan_ws = ar.get<char>(scan_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_preprocess_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, to_cam(cam), gx, gy, means3D, cov6, opacity,
                       colours_precomp, shs, (int)sh_degree, (int)sh_coeffs, make_float3(campos[0], campos[1], campos[2]),
                       (long)n, key, idx, touched, (float4*)rec, rect, radii, cu_wide_grid(gx, gy) ? 1 : 0);
    int rc = sort_pairs_u32(key, idx, key_sorted, sorted_idx, ktmp, vtmp, n, 0, 32, sort_ws, sort_bytes, s);
    if (rc) return rc;
    rc = scan_exclusive_u32(touched, offsets, n, scan_ws, scan_bytes, s, sorted_idx);
    if (rc) return rc;
    if (count_host) hipMemcpyAsync(count_host, offsets + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    return check_launch("g2pc_raster_front_cu");
}

// CU semantics, back half: duplicate -> tile sort -> ranges -> blend -> running-state update.
// out_color f32[3,H,W], out_depth / out_invdepth f32[H,W] are zero-filled here.  cam_key u64[n], cam_surf u32[n] are
// per-camera scratch.  cur_* (optional) receive this camera's gauss_contributions / gauss_pixels / surface distances.
int g2pc_raster_back_cu_tiles(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t num_instances, const float* rec,
                        const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets, int calculate_surface_distance, unsigned long long* cam_key,
                        uint32_t* cam_surf, float* out_color, float* out_depth, float* out_invdepth,
                        float* max_contrib, float* total_contrib, float* colours, float* min_surf,
                        int32_t* winner_cam, int32_t cam_index, float* cur_contrib, int32_t* cur_pixels, float* cur_surf,
                        int phases, int32_t tile_first, int32_t tile_step, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && rec && rect && sorted_idx && offsets && cam_key && cam_surf && out_color &&
                     out_depth && out_invdepth && max_contrib && total_contrib && colours && min_surf && ws && n > 0,
                 G2PC_ERR_ARG, "bad arguments");
    const int W = cam->width, H = cam->height;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    G2PC_REQUIRE(tile_step >= 1 && tile_first >= 0 && tile_first < tile_step, G2PC_ERR_ARG, "bad tile shard");
    const bool sharded = tile_step > 1;                 // the images then hold this rank's tiles only (zero elsewhere)
    hipStream_t s = (hipStream_t)stream;
    const long L = num_instances;
    Arena ar(ws, ws_bytes);
    uint32_t* inst_tile = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* inst_g = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_sorted = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* g_sorted = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_tmp = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* g_tmp = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_start = ar.get<uint32_t>((size_t)T + 2);
    size_t sort_bytes = sort_workspace(L), scan_bytes = scan_workspace(T + 1);
    char* sort_ws = ar.get<char>(sort_bytes);
    char* scan_ws = ar.get<char>(scan_bytes);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    const int gshift = packed_instance_shift((long)n, T);
    const uint32_t gmask = gshift ? ((1u << gshift) - 1u) : 0xFFFFFFFFu;
    if (phases & 1) {
    // (k_tile_ranges writes every entry of tile_start: no memset)
    if (mask || sharded) {                        // without a mask every pixel is written by the blend kernel
        hipMemsetAsync(out_color, 0, (size_t)3 * W * H * 4, s);
        hipMemsetAsync(out_depth, 0, (size_t)W * H * 4, s);
        hipMemsetAsync(out_invdepth, 0, (size_t)W * H * 4, s);
    }
    hipLaunchKernelGGL(k_init_camera_state_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, (unsigned long long*)cam_key, (uint32_t*)cam_surf, (long)n);
    if (L > 0) {
        if (cu_wide_grid(gx, gy))
            hipLaunchKernelGGL(k_duplicate<true>, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, sorted_idx, offsets, rect, (long)n, gx,
                               inst_tile, inst_g, (const uint32_t*)nullptr, gshift, (size_t)0, (const int32_t*)nullptr, (const G2pcCameraJob*)nullptr);
        else
            hipLaunchKernelGGL(k_duplicate<false>, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, sorted_idx, offsets, rect, (long)n, gx,
                               inst_tile, inst_g, (const uint32_t*)nullptr, gshift, (size_t)0, (const int32_t*)nullptr, (const G2pcCameraJob*)nullptr);
        int rc = gshift ? sort_pairs_u32(inst_tile, nullptr, tile_sorted, nullptr, tile_tmp, nullptr, L, gshift,
                                         gshift + bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s)
                        : sort_pairs_u32(inst_tile, inst_g, tile_sorted, g_sorted, tile_tmp, g_tmp, L, 0,
                                         bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s);
        if (rc) return rc;
    }
    (void)scan_ws; (void)scan_bytes;
    hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(L + 1, RA_T)), dim3(RA_T), 0, s, tile_sorted, L, T, tile_start, (const uint32_t*)nullptr, gshift, (size_t)0);
    }
    if ((phases & 2) && tile_first < T)
    hipLaunchKernelGGL(k_blend_cu, dim3((unsigned)((T - tile_first + tile_step - 1) / tile_step)), dim3(CU_T), 0, s, W, H, gx,
                       (int)tile_first, (int)tile_step, tile_start, gshift ? tile_sorted : g_sorted, gmask, (const float4*)rec,
                       mask, make_float3(cam->bg[0], cam->bg[1], cam->bg[2]),
                       calculate_surface_distance, cam_key, cam_surf, out_color, out_depth, out_invdepth);
    if (phases & 4)
    hipLaunchKernelGGL(k_update_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, cam_key, cam_surf, (long)n, W, H, out_color,
                       max_contrib, total_contrib, colours, min_surf, winner_cam, cam_index, cur_contrib, cur_pixels, cur_surf);
    return check_launch("g2pc_raster_back_cu");
}

// CU semantics, bin + blend of one camera WITHOUT the host in the loop (the python-semantics path's scheme): the launches are
// sized for `capacity` instances, the true count stays on the device (k_resolve_count -> l_eff; the L-dependent kernels
// read it) and travels to the pinned count_host[0] on its own.  A camera that does not fit is skipped as a whole (empty
// tile lists: every pixel gets the background, no Gaussian a contribution) and the caller, who sees count_host[0] >
// capacity later, renders it again with g2pc_raster_back_cu[_tiles].  Follow with g2pc_raster_back_cu_tiles(phases = 4,
// num_instances = capacity) for the running-state update.
int g2pc_raster_back_cu_dev(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t capacity, const float* rec,
                            const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets,
                            int calculate_surface_distance, unsigned long long* cam_key, uint32_t* cam_surf, float* out_color,
                            float* out_depth, float* out_invdepth, uint32_t* count_host, int32_t tile_first, int32_t tile_step,
                            void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && rec && rect && sorted_idx && offsets && cam_key && cam_surf && out_color && out_depth && out_invdepth &&
                     ws && n > 0 && capacity > 0, G2PC_ERR_ARG, "bad arguments");
    const int W = cam->width, H = cam->height;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    G2PC_REQUIRE(tile_step >= 1 && tile_first >= 0 && tile_first < tile_step, G2PC_ERR_ARG, "bad tile shard");
    const bool sharded = tile_step > 1;
    hipStream_t s = (hipStream_t)stream;
    const long L = capacity;
    Arena ar(ws, ws_bytes);
    uint32_t* inst_tile = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* inst_g = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_sorted = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* g_sorted = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_tmp = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* g_tmp = ar.get<uint32_t>((size_t)L + 1);
    uint32_t* tile_start = ar.get<uint32_t>((size_t)T + 2);
    size_t sort_bytes = sort_workspace(L);
    char* sort_ws = ar.get<char>(sort_bytes);
    uint32_t* l_eff = ar.get<uint32_t>(1);
    G2PC_REQUIRE(ar.ok(), G2PC_ERR_WORKSPACE, "workspace too small");
    hipLaunchKernelGGL(k_resolve_count, dim3(1), dim3(64), 0, s, offsets + n, (uint32_t)capacity, l_eff, count_host,
                       (const uint32_t*)nullptr, (size_t)0);
    if (mask || sharded) {                        // without a mask every pixel is written by the blend kernel
        hipMemsetAsync(out_color, 0, (size_t)3 * W * H * 4, s);
        hipMemsetAsync(out_depth, 0, (size_t)W * H * 4, s);
        hipMemsetAsync(out_invdepth, 0, (size_t)W * H * 4, s);
    }
    hipLaunchKernelGGL(k_init_camera_state_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, (unsigned long long*)cam_key, (uint32_t*)cam_surf, (long)n);
    const int gshift = packed_instance_shift((long)n, T);
    const uint32_t gmask = gshift ? ((1u << gshift) - 1u) : 0xFFFFFFFFu;
    if (cu_wide_grid(gx, gy))
        hipLaunchKernelGGL(k_duplicate<true>, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, sorted_idx, offsets, rect, (long)n, gx, inst_tile, inst_g,
                           (const uint32_t*)l_eff, gshift, (size_t)0, (const int32_t*)nullptr, (const G2pcCameraJob*)nullptr);
    else
        hipLaunchKernelGGL(k_duplicate<false>, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, sorted_idx, offsets, rect, (long)n, gx, inst_tile, inst_g,
                           (const uint32_t*)l_eff, gshift, (size_t)0, (const int32_t*)nullptr, (const G2pcCameraJob*)nullptr);
    int rc = gshift ? sort_pairs_u32(inst_tile, nullptr, tile_sorted, nullptr, tile_tmp, nullptr, L, gshift,
                                     gshift + bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s, l_eff)
                    : sort_pairs_u32(inst_tile, inst_g, tile_sorted, g_sorted, tile_tmp, g_tmp, L, 0, bits_for_tiles((unsigned)T),
                                     sort_ws, sort_bytes, s, l_eff);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(L + 1, RA_T)), dim3(RA_T), 0, s, tile_sorted, L, T, tile_start, (const uint32_t*)l_eff, gshift, (size_t)0);
    if (tile_first < T)
        hipLaunchKernelGGL(k_blend_cu, dim3((unsigned)((T - tile_first + tile_step - 1) / tile_step)), dim3(CU_T), 0, s, W, H, gx,
                           (int)tile_first, (int)tile_step, tile_start, gshift ? tile_sorted : g_sorted, gmask, (const float4*)rec, mask,
                           make_float3(cam->bg[0], cam->bg[1], cam->bg[2]), calculate_surface_distance, cam_key, cam_surf,
                           out_color, out_depth, out_invdepth);
    return check_launch("g2pc_raster_back_cu_dev");
}

int g2pc_raster_back_cu(const G2pcCamera* cam, const int32_t* mask, int64_t n, int64_t num_instances, const float* rec,
                        const uint32_t* rect, const uint32_t* sorted_idx, const uint32_t* offsets, int calculate_surface_distance,
                        unsigned long long* cam_key, uint32_t* cam_surf, float* out_color, float* out_depth, float* out_invdepth,
                        float* max_contrib, float* total_contrib, float* colours, float* min_surf, int32_t* winner_cam,
                        int32_t cam_index, float* cur_contrib, int32_t* cur_pixels, float* cur_surf, int phases, void* ws,
                        size_t ws_bytes, void* stream) {
    return g2pc_raster_back_cu_tiles(cam, mask, n, num_instances, rec, rect, sorted_idx, offsets, calculate_surface_distance,
                                     cam_key, cam_surf, out_color, out_depth, out_invdepth, max_contrib, total_contrib, colours,
                                     min_surf, winner_cam, cam_index, cur_contrib, cur_pixels, cur_surf, phases, 0, 1, ws,
                                     ws_bytes, stream);
}
}
