// Shared by the two translation units of the rasteriser -- raster.hip (python-renderer semantics, the parity target) and
// raster_cu.hip (native-rasteriser semantics): the device copies of the camera and the tile layout, the instance
// duplication kernel, the host helpers around the packed keys, and the build-time experiment knobs.
#pragma once
#include "g2pc_internal.h"

namespace g2pc {

constexpr int RA_T = 256;
constexpr float LOG2E = 1.4426950408889634f;

// ---- experiments -------------------------------------------------------------------------------------------------
// libg2pc.so has NO process-global tuning state: what rounds 2-4 exposed as g2pc_set_* / g2pc_debug_* entry points (blend
// kernel variants, walk caps, launch-geometry and sort-tuning knobs, per-chunk clocks) exists only in builds with
// -DG2PC_EXPERIMENTS (tools/experiments/build_variant.sh exp -DG2PC_EXPERIMENTS -> libg2pc_exp.so; the test suite's "_exp"
// emulator build), together with the kernels nothing in the product selects (experiments/blend_variants.inl).
#ifdef G2PC_EXPERIMENTS
struct Knobs {
    int head_threads = RA_T;          // block size of k_preprocess_py / k_duplicate / k_tile_ranges (64, 128, 256)
    int extra_launches = 0;           // empty kernels after every camera batch's preprocess (what does a kernel boundary cost?)
    int walk_cap = 0;                 // the dual-list blend stops a walk after this many 64-entry batches (WRONG results)
    int blend_variant = 1;            // 0 packed, 1 dual-list (the product's choice), 2 / 3 two-wave, 4 / 5 scalar-gather, 6 dual-list unroll 2
    int depth_bucket_sort = 1;        // captured camera path: 0 = radix depth sort
    uint32_t* chunk_work = nullptr;   // per-chunk walk statistics and clocks (tools/chunk_work.py)
};
extern Knobs g_knobs;
#define G2PC_KNOB(name, product_value) (::g2pc::g_knobs.name)
#else
#define G2PC_KNOB(name, product_value) (product_value)
#endif

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
#ifdef G2PC_EXPERIMENTS
    int walk_cap;                     // DIAGNOSTIC (g2pc_debug_set_walk_cap): the dual-list blend stops a walk after this many batches (0 = never; results are then WRONG)
#endif
};


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

__global__ void k_tile_ranges(const uint32_t* __restrict__ tile_sorted, long L, int T, uint32_t* __restrict__ tile_start,
                              const uint32_t* __restrict__ l_dev, int gshift, size_t cs);
__global__ void k_resolve_count(const uint32_t* __restrict__ total, uint32_t capacity, uint32_t* __restrict__ l_eff,
                                uint32_t* __restrict__ count_host, const uint32_t* __restrict__ depth_overflow, size_t cs);

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


static inline Cam to_cam(const G2pcCamera* c) {
    Cam k;
    for (int i = 0; i < 16; ++i) { k.V[i] = c->view[i]; k.P[i] = c->proj[i]; }
    k.tan_fovx = c->tan_fovx; k.tan_fovy = c->tan_fovy; k.focal_x = c->focal_x; k.focal_y = c->focal_y;
    k.W = c->width; k.H = c->height;
    k.bg[0] = c->bg[0]; k.bg[1] = c->bg[1]; k.bg[2] = c->bg[2];
    k.lim_x = c->lim_x; k.lim_y = c->lim_y;
    return k;
}
static inline Layout to_layout(const G2pcTileLayout* l) {
    Layout k;
#ifdef G2PC_EXPERIMENTS
    k.walk_cap = g_knobs.walk_cap;
#endif
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
static inline bool cu_wide_grid(int gx, int gy) { return gx > 256 || gy > 256; }
static inline int bits_for_tiles(unsigned t) { int b = 1; while ((1u << b) < t && b < 31) ++b; return b; }
// packed visibility keys: the tile-sequence field is seq_bits wide (12 .. 14), the camera slot gets the 20 - seq_bits above it
static inline bool layout_keys_ok(const G2pcTileLayout* l) {
    const int sb = l->seq_bits ? l->seq_bits : 12;
    const long top = l->seq_count ? (long)l->seq_base + l->seq_count : (long)l->nx * l->ny;     // largest sequence number + 1
    return sb >= 12 && sb <= 14 && l->seq_base >= 0 && l->seq_count >= 0 && top <= (1l << sb);
}
static inline uint32_t max_camera_slot(const G2pcTileLayout* l) { return (1u << (20 - (l->seq_bits ? l->seq_bits : 12))) - 1u; }


// Packed tile-sort instances: when the tile id and the Gaussian index share one 32-bit word (tile << gshift | index) the
// stable sort by tile moves keys only -- half the traffic of the two passes -- and the blend masks the index out.
// Returns gshift (0: they do not fit, separate arrays as before).
static inline int packed_instance_shift(long n, int T) {
    int gbits = 1;
    while (((long)1 << gbits) < n) ++gbits;
    return (gbits + bits_for_tiles((unsigned)T) <= 32) ? gbits : 0;
}

}  // namespace g2pc
