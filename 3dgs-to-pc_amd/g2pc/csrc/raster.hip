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
#include "raster_common.h"
#include "py_project.inl"
#include <type_traits>

namespace g2pc {

// build-time A/B switches (tools/experiments/build_variant.sh): wave priority of long blend walks / of the head kernels
#ifndef G2PC_BLEND_PRIO
#define G2PC_BLEND_PRIO 2
#endif
#ifndef G2PC_HEAD_PRIO
#define G2PC_HEAD_PRIO 0
#endif
#ifdef G2PC_EXPERIMENTS
Knobs g_knobs;
__global__ void k_nothing(uint32_t* __restrict__ p) { if (p && threadIdx.x == 1000) p[0] = 0; }
#endif

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
                                                       BucketHdr* __restrict__ mm0, uint32_t mm_slots, int ncam, long reload) {
    // device-resident camera: lets ONE captured launch sequence serve every camera (scalar loads, see below).
    // Batched launch (grid.y cameras, or MULTI): camera c's job is the c-th G2pcCameraJob, its outputs live in the c-th arena.
    // mm != nullptr: the depth keys go to the bucket sort (prims.hip), whose first pass -- the range of the keys -- is folded
    // in here: every block leaves (max ~key, max key) in slot blockIdx.x % mm_slots of the (zeroed) header.
    if (G2PC_HEAD_PRIO) __builtin_amdgcn_s_setprio(G2PC_HEAD_PRIO);
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
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (block size: RA_T, or G2PC_KNOB(head_threads, RA_T) in the camera pipeline)
    // ALL of the Gaussian's inputs are requested at once (17 loads in one round): behind the in-front-of-the-camera test the
    // covariance, opacity and colour were a second round trip, which a wave with one or two neighbours on its SIMD (the rest
    // of the registers belong to another camera's blend) sits out in full
    // MULTI: the inputs are read again for every camera -- from the L1 / L2, where the first camera's reads left them -- rather
    // than held across the loop: 17 live registers more put the kernel at 70 VGPRs, and beside the blends of the other streams
    // (5 waves x 96 VGPRs allocated per SIMD) a wave's register count decides how many of them a retiring blend wave makes
    // room for (40: three; 72: one).  `reload` is 0: an offset the compiler cannot see through (it would hoist the loads).
    __syncthreads();
    const unsigned c_first = MULTI ? 0u : blockIdx.y, c_end = MULTI ? (unsigned)ncam : blockIdx.y + 1u;
    for (unsigned c = c_first; c < c_end; ++c) {
    float x = 0.f, y = 0.f, z = 0.f, S9[9], op_in = 0.f, col_r = 0.f, col_g = 0.f, col_b = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) S9[k] = 0.f;
    if (i < n) {
        const long ii = i + (long)c * reload;
        x = means3D[3 * ii]; y = means3D[3 * ii + 1]; z = means3D[3 * ii + 2];
#pragma unroll
        for (int k = 0; k < 9; ++k) S9[k] = cov9[9 * ii + k];
        op_in = opacity[ii];
        col_r = colours[3 * ii]; col_g = colours[3 * ii + 1]; col_b = colours[3 * ii + 2];
    }
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
    if (tiles_touched) tiles_touched[i] = touched;   // (nullptr: the fused emission derives it from the rect -- no child pass)
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

// ---------------------------------------------------------------------------------------------------------
// K6 (PY): blend.  One single-wave block per (tile, 256-pixel chunk); 64 lanes x 4 pixels; Gaussians staged through
// wave-private LDS in batches of 64 (no s_barrier: 32 independent waves per CU, fine-grained early exit).  For every Gaussian the wave reduces (max T*alpha, lowest pixel among the maxima) and
// lane 0 publishes  key = contribution_bits << 32 | ~(slot << 24 | tile_seq << 12 | pixel)  with one 64-bit
// atomicMax -- but only when some lane can beat the value staged from the running maximum.
// ---------------------------------------------------------------------------------------------------------
constexpr int BL_T = 64, BL_BATCH = 64;

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
        if (G2PC_BLEND_PRIO && processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(G2PC_BLEND_PRIO);
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
#ifdef G2PC_EXPERIMENTS
    if (chunk_work && lane == 0) {
        chunk_work[8 * chunk_i] = end - start;
        chunk_work[8 * chunk_i + 1] = processed;
    }
#endif
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
#ifdef G2PC_EXPERIMENTS
    const unsigned long long clk0 = chunk_work ? wall_clock64() : 0ull;     // diagnostics only
#endif
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
    uint32_t processed = 0, visits = 0;                     // (visits: diagnostics of the experiments build)
    (void)visits;
    for (uint32_t b = start; b < end; b += BL_BATCH) {
        processed = b + BL_BATCH - start;
        if (G2PC_BLEND_PRIO && processed == 16 * BL_BATCH) __builtin_amdgcn_s_setprio(G2PC_BLEND_PRIO);
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
#ifdef G2PC_EXPERIMENTS
        if (lay.walk_cap && processed >= (uint32_t)lay.walk_cap * BL_BATCH) break;      // diagnostic: truncated walk
#endif
    }
#ifdef G2PC_EXPERIMENTS
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
#endif
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

#ifdef G2PC_EXPERIMENTS
#include "experiments/blend_variants.inl"
#endif

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

// ---- host side of the PY path, shared by the two-call API (count read back by the host) and the single-call,
// capture-safe API (count stays on the device, launch geometry fixed by a capacity) -------------------------------
struct PyFrontBuffers { float4* rec; uint32_t *rect, *sorted_idx, *offsets; };      // rec: 4 x float4 per Gaussian

static size_t py_front_ws(long n) {
    return align_up((size_t)n * 4) * 6 + sort_workspace(n) + scan_workspace(n) + bucket_sort_workspace(n) + 4096;
}
#ifndef G2PC_PREPROCESS_MULTI
#define G2PC_PREPROCESS_MULTI 1
#endif
static const int g_preprocess_multi = G2PC_PREPROCESS_MULTI;   // build-time A/B switch: one thread per Gaussian for all cameras of a batch
#ifndef G2PC_FUSED_EMIT
#define G2PC_FUSED_EMIT 1
#endif
static const int g_fused_emit = G2PC_FUSED_EMIT;   // build-time A/B switch: 0 = scan + k_duplicate + k_resolve_count as until round 4
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
    // (without a child pass a Gaussian's instance count IS its rect's area: the counts are then neither written nor read)
    const bool need_counts = !emit || layout->tile_parent != nullptr;
    if (emit) { emit->weight = need_counts ? touched : nullptr; emit->rect = fb.rect; }
    if (cam_dev && bt.n > 1 && g_preprocess_multi)
        hipLaunchKernelGGL((k_preprocess_py<true, true>), dim3(cdiv(n, G2PC_KNOB(head_threads, RA_T)), 1u), dim3(G2PC_KNOB(head_threads, RA_T)), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, fold ? (uint32_t*)nullptr : idx_rev, need_counts ? touched : (uint32_t*)nullptr, colours, fb.rec, fb.rect,
                           bt.cs, hdr, plan.nminmax, bt.n, 0l);
    else if (cam_dev)
        hipLaunchKernelGGL((k_preprocess_py<true, false>), dim3(cdiv(n, G2PC_KNOB(head_threads, RA_T)), (unsigned)bt.n), dim3(G2PC_KNOB(head_threads, RA_T)), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, fold ? (uint32_t*)nullptr : idx_rev, need_counts ? touched : (uint32_t*)nullptr, colours, fb.rec, fb.rect,
                           bt.cs, hdr, plan.nminmax, 1, 0l);
    else
        hipLaunchKernelGGL((k_preprocess_py<false, false>), dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, cam_val, cam_dev, to_layout(layout),
                           means3D, cov9, opacity, n, key_rev, idx_rev, touched, colours, fb.rec, fb.rect, (size_t)0,
                           (BucketHdr*)nullptr, 1u, 1, 0l);
#ifdef G2PC_EXPERIMENTS
    for (int k = 0; k < g_knobs.extra_launches; ++k) hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, (uint32_t*)nullptr);
#endif
    if (emit && !(fold && depth_overflow)) { set_error("raster_front_py", "fused emission without the folded bucket sort"); return G2PC_ERR_ARG; }
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
                hipLaunchKernelGGL(k_duplicate<false>, dim3(cdiv(n, G2PC_KNOB(head_threads, RA_T)), (unsigned)bt.n), dim3(G2PC_KNOB(head_threads, RA_T)), 0, s, fb.sorted_idx, fb.offsets, fb.rect, n, lay.nx,
                                   inst_tile, inst_g, l_eff, gshift, bt.cs, sc.cam_dev ? lay.tile_parent : (const int32_t*)nullptr,
                                   (const G2pcCameraJob*)sc.cam_dev);
            int rc = gshift ? sort_pairs_u32(inst_tile, nullptr, tile_sorted, nullptr, tile_tmp, nullptr, L, gshift,
                                             gshift + bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s, l_eff, bt)
                            : sort_pairs_u32(inst_tile, inst_g, tile_sorted, g_sorted, tile_tmp, g_tmp, L, 0,
                                             bits_for_tiles((unsigned)T), sort_ws, sort_bytes, s, l_eff, bt);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(L + 1, G2PC_KNOB(head_threads, RA_T)), (unsigned)bt.n), dim3(G2PC_KNOB(head_threads, RA_T)), 0, s, tile_sorted, L, T, tile_start, l_eff, gshift, bt.cs);
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
                       ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, G2PC_KNOB(chunk_work, (uint32_t*)nullptr), ba.job, bt.cs)
        // t_floor == 0 is the to-the-letter mode: it takes the kernel that evaluates the exponent in the reference's
        // operation order (k_blend_py_pk); the dual-list kernel's expanded exponent differs by up to ~2e-5 relative
        // in alpha.  A captured camera reads t_floor from its device job: the caller says so with phase bit 8.
        const bool exact = ba.job ? ((phases & 8) != 0) : (ba.t_floor == 0.0f);
#ifdef G2PC_EXPERIMENTS
        const int variant = g_knobs.blend_variant;
        switch (layout->chunk_subblocks) {
            case 1: G2PC_BLEND(k_blend_py<1, 4>); break;
            case 2: {
                if (variant == 6 && !exact) G2PC_BLEND(k_blend_py_dl<2>);
                else if (variant == 4 && !exact) G2PC_BLEND(k_blend_py_sg<4>);
                else if (variant == 5 && !exact) G2PC_BLEND(k_blend_py_sg<2>);
                else if (variant == 3 && !exact)
                    hipLaunchKernelGGL((k_blend_py_2w<2>), dim3((unsigned)bt.n, chunks_y, cdiv(layout->num_chunks, chunks_y)), dim3(2 * BL_T), 0, s,
                                       lay, layout->chunk_tile, layout->chunk_pix0, A.tile_range, blend_list, gmask, (const float4*)fb.rec,
                                       best_key, ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, g_knobs.chunk_work, ba.job, bt.cs);
                else if (variant == 2 && !exact)
                    hipLaunchKernelGGL((k_blend_py_2w<4>), dim3((unsigned)bt.n, chunks_y, cdiv(layout->num_chunks, chunks_y)), dim3(2 * BL_T), 0, s,
                                       lay, layout->chunk_tile, layout->chunk_pix0, A.tile_range, blend_list, gmask, (const float4*)fb.rec,
                                       best_key, ba.camera_slot << (12 + lay.seq_bits), ba.t_floor, ba.bg, tilebuf, g_knobs.chunk_work, ba.job, bt.cs);
                else if (variant == 1 && !exact) G2PC_BLEND(k_blend_py_dl<4>);
                else G2PC_BLEND(k_blend_py_pk<4>);
                break;
            }
            case 4: G2PC_BLEND(k_blend_py<4, 1>); break;
            default: set_error("raster_back_py", "chunk_subblocks must be 1, 2 or 4"); return G2PC_ERR_ARG;
        }
#else
        // one wave per pair of adjacent 8x8 sub-blocks (g2pc/tiles.py: chunk_subblocks = 2) is the layout the product blends
        if (layout->chunk_subblocks != 2) { set_error("raster_back_py", "chunk_subblocks must be 2"); return G2PC_ERR_UNSUPPORTED; }
        if (exact) G2PC_BLEND(k_blend_py_pk<4>);
        else G2PC_BLEND(k_blend_py_dl<4>);
#endif
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
    const bool bucket = G2PC_KNOB(depth_bucket_sort, 1) && bucket_sort_pays((long)n);
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

int g2pc_raster_contributions(const unsigned long long* best_key, int64_t n, float* out, void* stream) {
    using namespace g2pc;
    if (n <= 0) return G2PC_OK;
    hipLaunchKernelGGL(k_contributions, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, best_key, (long)n, out);
    return check_launch("g2pc_raster_contributions");
}
}

#ifdef G2PC_EXPERIMENTS
#include "experiments/knobs.inl"
#endif
