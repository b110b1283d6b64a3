// Native-rasteriser ("cuda") semantics of the reference (renderer_type="cuda": forward.cu, rasterizer_impl.cu,
// rasterize_points.cu, gaussian_pointcloud_rasterization/__init__.py) -- deterministic spec of SURVEY.md §8(a.5).  Shares the
// sort / scan primitives, the duplication kernel and the tile-range kernel with the python-semantics path (raster.hip).
#include "raster_common.h"

namespace g2pc {

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
                                                       int32_t* __restrict__ radii, int wide, int antialiasing,
                                                       unsigned long long* __restrict__ cam_key, uint32_t* __restrict__ cam_surf,
                                                       BucketHdr* __restrict__ mm, uint32_t mm_slots) {
    // cam_key / cam_surf (optional): the camera's per-Gaussian visibility state is initialised here (packed key 0, surface
    // distance FLT_MAX) instead of by a launch of its own; mm (optional): the depth bucket sort's header -- the range of the
    // keys is noted here (one atomic pair per block) instead of by a pass over the keys (k_bk_minmax), as on the PY path
#pragma clang fp contract(off)
    __shared__ uint32_t s_mm[2];
    if (mm && threadIdx.x < 2) s_mm[threadIdx.x] = 0u;
    if (mm) __syncthreads();
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    uint32_t key = 0xFFFFFFFFu;
    if (i < n) {
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    const float* V = cam.V;
    const float* P = cam.P;
    uint32_t touched = 0, rc = 0, rc_hi = 0;
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
        float cxx = a0[0] * T0[0] + a0[1] * T0[1] + a0[2] * T0[2];
        float cxy = a1[0] * T0[0] + a1[1] * T0[1] + a1[2] * T0[2];
        float cyy = a1[0] * T1[0] + a1[1] * T1[1] + a1[2] * T1[2];
        const float det_cov = cxx * cyy - cxy * cxy;                    // forward.cu:217-225: before the 0.3-pixel dilation
        cxx += 0.3f;
        cyy += 0.3f;
        float det = cxx * cyy - cxy * cxy;
        // antialiasing (forward.cu:224-225,264): the opacity is scaled by sqrt(max(0.000025, det(cov) / det(cov + 0.3 I)))
        const float h_scale = antialiasing ? sqrtf(fmaxf(0.000025f, det_cov / det)) : 1.0f;
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
                const float op = antialiasing ? opacity[i] * h_scale : opacity[i];
                rec[4 * i + 1] = make_float4(qc, op, tz0, my_radius);                    // as on the PY path
                // per-wave cull of k_blend_cu (rect_may_touch): slopes of the exponent's edge maxima and the exponent below
                // which alpha < 1/255 (with a 0.7 % margin for the different rounding of the bound)
                rec[4 * i + 3] = make_float4(-qb / (2.0f * qc), -qb / (2.0f * qa), -8.00435f - log2f(op), 0.0f);
                float cr, cg, cb;
                if (colours_precomp) {
                    cr = colours_precomp[3 * i]; cg = colours_precomp[3 * i + 1]; cb = colours_precomp[3 * i + 2];
                } else {
                    float dx = x - campos.x, dy = y - campos.y, dz = z - campos.z;
                    float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dx /= len; dy /= len; dz /= len;
                    // The Gaussian's coefficients [K][3] are one contiguous block (192 bytes at K = 16): read as 16-byte vectors when
                    // the block is 16-byte aligned (K a multiple of 4) -- a quarter of the load instructions, each of which costs the
                    // address unit one step per cache line it touches (until round 6: 48 single-dword loads, 64 lines apiece) --, band by
                    // band, so that only one band's coefficients are live.  Per channel the sum runs left to right exactly as
                    // computeColorFromSH writes it (forward.cu:22-73): bit-identical.
                    // sh_coeffs < 0: PLANE-MAJOR coefficients (g2pc_sh_planes: f32[3K/4][n][4], vector v of Gaussian i at v * n + i) --
                    // the lanes of a wave then read 1 KB contiguous per vector instead of one 16-byte piece every 192 bytes; the scene
                    // is static over a job's cameras, so the binding transposes once per job.
                    const bool planes = sh_coeffs < 0;
                    if (planes) sh_coeffs = -sh_coeffs;
                    const float* sh = shs + (size_t)i * sh_coeffs * 3;
#ifndef G2PC_CU_SH_VEC
#define G2PC_CU_SH_VEC 1         // build-time A/B switch: 0 = single-dword loads, as until round 6
#endif
                    const bool vec = planes || (G2PC_CU_SH_VEC && ((sh_coeffs & 3) == 0) && ((((size_t)shs) & 15) == 0));
                    float c[48];
                    auto fetch = [&](int v0, int v1) {           // floats [4 v0, 4 v1) of the block
                        if (vec) {
#pragma unroll
                            for (int v = v0; v < v1; ++v) {
                                const float4 q = planes ? ((const float4*)shs)[(size_t)v * (size_t)n + (size_t)i] : ((const float4*)sh)[v];
                                c[4 * v] = q.x; c[4 * v + 1] = q.y; c[4 * v + 2] = q.z; c[4 * v + 3] = q.w;
                            }
                        } else {
#pragma unroll
                            for (int k = 4 * v0; k < 4 * v1; ++k) c[k] = k < sh_coeffs * 3 ? sh[k] : 0.0f;
                        }
                    };
                    float res[3];
                    if (sh_degree > 0) fetch(0, 3); else fetch(0, 1);      // bands 0 (3 floats) and 1 (9); constant bounds: c[] stays in registers
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.28209479177387814f * c[ch];
                        if (sh_degree > 0)
                            v = v - 0.4886025119029199f * dy * c[3 + ch] + 0.4886025119029199f * dz * c[6 + ch] -
                                0.4886025119029199f * dx * c[9 + ch];
                        res[ch] = v;
                    }
                    if (sh_degree > 1) {
                        const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                        fetch(3, 7);                                       // floats 12 .. 27: band 2 (12 .. 26) and the first of band 3
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            res[ch] = res[ch] + kSH_C2[0] * xy * c[12 + ch] + kSH_C2[1] * yz * c[15 + ch] +
                                      kSH_C2[2] * (2.0f * zz - xx - yy) * c[18 + ch] + kSH_C2[3] * xz * c[21 + ch] +
                                      kSH_C2[4] * (xx - yy) * c[24 + ch];
                        if (sh_degree > 2) {
                            fetch(7, 12);                                  // floats 28 .. 47
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch)
                                res[ch] = res[ch] + kSH_C3[0] * dy * (3.0f * xx - yy) * c[27 + ch] + kSH_C3[1] * xy * dz * c[30 + ch] +
                                          kSH_C3[2] * dy * (4.0f * zz - xx - yy) * c[33 + ch] +
                                          kSH_C3[3] * dz * (2.0f * zz - 3.0f * xx - 3.0f * yy) * c[36 + ch] +
                                          kSH_C3[4] * dx * (4.0f * zz - xx - yy) * c[39 + ch] +
                                          kSH_C3[5] * dz * (xx - yy) * c[42 + ch] + kSH_C3[6] * dx * (xx - 3.0f * yy) * c[45 + ch];
                        }
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float v = res[ch] + 0.5f;
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
    if (tiles_touched) tiles_touched[i] = touched;
    if (wide) { rect[2 * i] = rc; rect[2 * i + 1] = rc_hi; } else rect[i] = rc;
    radii[i] = rad;
    if (cam_key) { cam_key[i] = 0ull; cam_surf[i] = 0x7F7FFFFFu; }             // (k_init_camera_state_cu's values)
    }
    if (mm) {
        uint32_t a = key != 0xFFFFFFFFu ? ~key : 0u, b = key != 0xFFFFFFFFu ? key : 0u;
        a = wave_max_u32(a); b = wave_max_u32(b);
        if ((threadIdx.x & 63) == 0) { atomicMax(&s_mm[0], a); atomicMax(&s_mm[1], b); }
        __syncthreads();
        if (threadIdx.x < 2) atomicMax(&mm->partial[2 * (blockIdx.x % mm_slots) + threadIdx.x], s_mm[threadIdx.x]);
    }
}
// g2pc_sh_planes: shs f32[n][K][3] (the reference's layout, forward.cu:31) -> f32[3K/4][n][4]
__global__ __launch_bounds__(RA_T) void k_sh_planes(const float4* __restrict__ in, long n, int vecs, float4* __restrict__ out) {
    const long t = (long)blockIdx.x * RA_T + threadIdx.x;
    if (t >= n * vecs) return;
    const long v = t / n, i = t - v * n;                 // consecutive threads: consecutive Gaussians of one plane (coalesced stores)
    out[t] = in[i * vecs + v];
}
// the depth bucket sort's header, cleared for the range notes of k_preprocess_cu
__global__ void k_bucket_hdr_init_cu(BucketHdr* __restrict__ h, BucketPlan plan) { bucket_hdr_init(h, plan, threadIdx.x, blockDim.x); }

// forward.cu:303-497 (renderCUDA).  One 256-thread block per 16x16 tile, one pixel per lane (thread rank t -> pixel
// (t % 16, t / 16), as in the reference); the tile's list is staged 256 instances at a time (= the reference's batches:
// the unit of the "everyone done" test and of the surface-distance pass).  Inside a batch the four waves run
// independently: 4 Gaussians per trip (independent exp chains), wave64 DPP reductions for the per-Gaussian maximum and
// for the surface distance, each guarded by a cheap "can any lane improve the staged value" ballot.
constexpr int CU_T = 256;

#ifdef G2PC_CU_BLEND_WAVES          // build-time A/B switch: blocks per CU the register allocation is held to (5: 96 VGPRs, 14 spilled)
#define G2PC_CU_BLEND_BOUNDS __launch_bounds__(CU_T, G2PC_CU_BLEND_WAVES)
#else
#define G2PC_CU_BLEND_BOUNDS __launch_bounds__(CU_T)
#endif
__global__ G2PC_CU_BLEND_BOUNDS void k_blend_cu(int W, int H, int grid_x, int tile_first, int tile_step,
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
    // Build-time A/B switch, OFF: a software pipeline of the staging (instance ids two batches ahead, the records of the NEXT batch
    // requested before the current one is walked -- every batch begins with two dependent round trips to memory that the four
    // waves of the tile sit out behind the barrier; PMC: 55 % of a lone launch's wave life is waiting).  Bit-identical (the staged
    // running maximum / surface distance are filters only), and SLOWER in the pipelined job: 27.2 - 27.5 ms against 24.9 - 25.3 on
    // one box, alternating (profiles/r06i_cmd1.log) -- its 18 extra registers cost a block per CU (114 VGPRs: 4 waves per SIMD;
    // held to 96 it spills 14 and is slower still), and with four cameras in flight other tiles already cover the wait.
#ifndef G2PC_CU_BLEND_PREFETCH
#define G2PC_CU_BLEND_PREFETCH 0
#endif
#if G2PC_CU_BLEND_PREFETCH
    bool v_cur = (start + t) < end, v_nxt = (start + CU_T + t) < end;
    uint32_t g_cur = v_cur ? (inst_g[start + t] & gmask) : 0u;
    uint32_t g_nxt = (G2PC_CU_BLEND_PREFETCH && v_nxt) ? (inst_g[start + CU_T + t] & gmask) : 0u;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q3 = q0, qc = q0;
    uint32_t qkey = 0u, qsurf = 0u;
    auto request = [&](uint32_t g) {
        q0 = rec[4 * (size_t)g]; q1 = rec[4 * (size_t)g + 1]; q3 = rec[4 * (size_t)g + 3]; qc = rec[4 * (size_t)g + 2];
        qkey = key_hi[2 * (size_t)g];
        if (calc_surf) qsurf = cam_surf[g];
    };
    if (G2PC_CU_BLEND_PREFETCH && v_cur) request(g_cur);
#endif
    for (uint32_t b = start; b < end; b += CU_T) {
        if (__syncthreads_and(done ? 1 : 0)) break;                       // forward.cu:373-375 (also: LDS is free again)
        // Stage entry t and decide, for each of the tile's four waves, whether this Gaussian's alpha can reach 1/255 on
        // that wave's pixels (rect_may_touch).  Below it the reference's loop body does nothing for the pixel (forward.cu:
        // 411-413 `continue`), so a Gaussian that fails for all 64 pixels of a wave is not walked by that wave at all --
        // same results bit for bit, ~half the (pixel, Gaussian) pairs of a 16x16 tile never evaluated.
        bool keep[4] = {false, false, false, false};
#if !G2PC_CU_BLEND_PREFETCH
        if (b + t < end) {
            uint32_t g = inst_g[b + t] & gmask;
            const float4 r0 = rec[4 * (size_t)g], r1 = rec[4 * (size_t)g + 1], r3 = rec[4 * (size_t)g + 3];
            s_p0[t] = r0;
            // .w: 1 / depth, formed ONCE per staged entry (the record's radius is not used by the blend).  The inverse-depth
            // map accumulates contrib / depth per pixel (forward.cu:428-430): the quotient is the same for all 256 pixels, and an
            // IEEE division is a 13-instruction sequence -- until round 6 every lane evaluated it on every visit.
            // Same operands, same operation: bit-identical.
            s_p1[t] = make_float4(r1.x, r1.y, r1.z, 1.0f / r1.z);
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
#else
        if (!G2PC_CU_BLEND_PREFETCH) {
            v_cur = b + t < end;
            if (v_cur) { g_cur = inst_g[b + t] & gmask; request(g_cur); }
        }
        if (v_cur) {
            const uint32_t g = g_cur;
            const float4 r0 = q0, r1 = q1, r3 = q3;
            s_p0[t] = r0;
            // .w: 1 / depth, formed ONCE per staged entry (the record's radius is not used by the blend).  The inverse-depth
            // map accumulates contrib / depth per pixel (forward.cu:428-430): the quotient is the same for all 256 pixels, and an
            // IEEE division is a 13-instruction sequence -- until round 6 every lane evaluated it on every visit.
            // Same operands, same operation: bit-identical.
            s_p1[t] = make_float4(r1.x, r1.y, r1.z, 1.0f / r1.z);
            const float4 c3 = qc;
            float gm = fmaxf(__uint_as_float(qkey), 1.17549435e-38f);
            s_p2[t] = make_float4(c3.x, c3.y, c3.z, gm);
            s_g[t] = g;
            if (calc_surf) s_surf[t] = qsurf;
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
        if (G2PC_CU_BLEND_PREFETCH) {              // next batch's records (its ids arrived a batch ago), the ids of the one after
            g_cur = g_nxt; v_cur = v_nxt;
            v_nxt = (b + 2 * CU_T + t) < end;
            g_nxt = v_nxt ? (inst_g[b + 2 * CU_T + t] & gmask) : 0u;
            if (v_cur) request(g_cur);
        }
#endif
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
            float alpha[4], power[4], dep[4], idep[4];
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
                idep[u] = q.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                G2PC_PIN(alpha[u]); G2PC_PIN(dep[u]); G2PC_PIN(idep[u]);
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
#ifndef G2PC_CU_HOIST_IDEPTH
#define G2PC_CU_HOIST_IDEPTH 1   // build-time A/B switch: 0 = 1 / depth by every lane on every visit, as until round 6
#endif
                Ei = fmaf(G2PC_CU_HOIST_IDEPTH ? idep[u] : 1.0f / depth, contrib, Ei);
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

// computeCov3D (forward.cu:115-150) for g2pc_rasterize_gaussians: Sigma = (S R)^T (S R) with glm's column-major constructors, ACTIVATED
// scales times scale_modifier, the quaternion as given (its normalisation is commented out there); every operation rounded on
// its own, sums over k = 0, 1, 2 in order as glm's operator* writes them.
__global__ __launch_bounds__(RA_T) void k_cov3d_cu(const float* __restrict__ scales, const float* __restrict__ rots, float mod,
                                                  long n, float* __restrict__ cov6) {
#pragma clang fp contract(off)
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const float s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
    const float r = rots[4 * i], x = rots[4 * i + 1], y = rots[4 * i + 2], z = rots[4 * i + 3];
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},     // R[column][row]
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float M[3][3];                                                          // M = S R: M[column][row] = s[row] R[column][row]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[c][k] = s[k] * R[c][k];
    auto dot = [&](int a, int b) { return M[a][0] * M[b][0] + M[a][1] * M[b][1] + M[a][2] * M[b][2]; };
    float* o = cov6 + 6 * i;
    o[0] = dot(0, 0); o[1] = dot(0, 1); o[2] = dot(0, 2); o[3] = dot(1, 1); o[4] = dot(1, 2); o[5] = dot(2, 2);
}

// results 9-11 of _C.rasterize_gaussians out of the camera's packed (contribution, ~pixel) keys and surface distances
__global__ __launch_bounds__(RA_T) void k_unpack_camera_cu(const unsigned long long* __restrict__ cam_key,
                                                          const uint32_t* __restrict__ cam_surf, long n,
                                                          float* __restrict__ contrib, float* __restrict__ surf,
                                                          int32_t* __restrict__ pixels) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = cam_key[i];
    const float c = __uint_as_float((uint32_t)(key >> 32));
    contrib[i] = c;
    pixels[i] = c > 0.0f ? (int32_t)(~(uint32_t)key) : 0;
    surf[i] = __uint_as_float(cam_surf[i]);
}

// per-camera state of the native-semantics blend in one launch: packed (contribution, ~pixel) keys = 0, surface distance = FLT_MAX
__global__ __launch_bounds__(RA_T) void k_init_camera_state_cu(unsigned long long* __restrict__ cam_key, uint32_t* __restrict__ cam_surf,
                                                              long n) {
    long i = (long)blockIdx.x * RA_T + threadIdx.x;
    if (i < n) { cam_key[i] = 0ull; cam_surf[i] = 0x7F7FFFFFu; }
}


}  // namespace g2pc

extern "C" {
int g2pc_sh_planes(const float* shs, int64_t n, int32_t sh_coeffs, float* planes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(shs && planes && n > 0 && sh_coeffs > 0 && (sh_coeffs & 3) == 0 && (((size_t)shs | (size_t)planes) & 15) == 0,
                 G2PC_ERR_ARG, "sh_coeffs must be a multiple of 4 and the arrays 16-byte aligned");
    const int vecs = sh_coeffs * 3 / 4;
    hipLaunchKernelGGL(k_sh_planes, dim3(cdiv((long)n * vecs, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, (const float4*)shs, (long)n, vecs,
                       (float4*)planes);
    return check_launch("g2pc_sh_planes");
}
int g2pc_mark_visible(const float* means3D, int64_t n, const float* viewmatrix, uint8_t* present, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(means3D && viewmatrix && present && n > 0, G2PC_ERR_ARG, "bad arguments");
    View16 V;
    for (int i = 0; i < 16; ++i) V.m[i] = viewmatrix[i];
    hipLaunchKernelGGL(k_mark_visible, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, (hipStream_t)stream, V, means3D, (long)n, present);
    return check_launch("g2pc_mark_visible");
}

// CU semantics, front half: preprocess (+SH) -> depth sort (ascending index on ties) -> tiles-touched scan.
static int front_cu_impl(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                         const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                         const float* campos, int64_t n, float* rec, uint32_t* rect, int32_t* radii,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream, int antialiasing) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && means3D && cov6 && opacity && campos && rec && rect && radii && sorted_idx && offsets && ws && n > 0,
                 G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE((colours_precomp != nullptr) != (shs != nullptr), G2PC_ERR_ARG,
                 "provide exactly one of precomputed colours or SHs");       // __init__.py:42-43
    G2PC_REQUIRE(!shs || (sh_degree >= 0 && sh_degree <= 3 && (sh_coeffs < 0 ? -sh_coeffs : sh_coeffs) >= (sh_degree + 1) * (sh_degree + 1) &&
                          (sh_coeffs > 0 || ((-sh_coeffs) & 3) == 0)), G2PC_ERR_ARG, "SH degree / coefficient count mismatch");
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
                       (long)n, key, idx, touched, (float4*)rec, rect, radii, cu_wide_grid(gx, gy) ? 1 : 0, antialiasing,
                       (unsigned long long*)nullptr, (uint32_t*)nullptr, (BucketHdr*)nullptr, 0u);
    int rc = sort_pairs_u32(key, idx, key_sorted, sorted_idx, ktmp, vtmp, n, 0, 32, sort_ws, sort_bytes, s);
    if (rc) return rc;
    rc = scan_exclusive_u32(touched, offsets, n, scan_ws, scan_bytes, s, sorted_idx);
    if (rc) return rc;
    if (count_host) hipMemcpyAsync(count_host, offsets + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    return check_launch("g2pc_raster_front_cu");
}
int g2pc_raster_front_cu(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                         const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                         const float* campos, int64_t n, float* rec, uint32_t* rect, int32_t* radii,
                         uint32_t* sorted_idx, uint32_t* offsets, uint32_t* count_host, void* ws, size_t ws_bytes,
                         void* stream) {
    return front_cu_impl(cam, means3D, cov6, opacity, colours_precomp, shs, sh_degree, sh_coeffs, campos, n, rec, rect, radii,
                         sorted_idx, offsets, count_host, ws, ws_bytes, stream, 0);
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
                     out_depth && out_invdepth && ws && n > 0 &&
                     (!(phases & 4) || (max_contrib && total_contrib && colours && min_surf)),
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

// CU semantics, ONE camera without the host in the loop and without a separate duplication: preprocess -> depth bucket sort whose
// last kernel emits the (tile, Gaussian) instances (BucketEmit, prims.hip: what g2pc_raster_front_cu's radix sort + scan and
// g2pc_raster_back_cu_dev's k_duplicate + k_resolve_count do in 17 launches) -> tile sort -> ranges -> blend.  Same results bit for
// bit: the bucket sort leaves THE stable ascending order (equal depths in ascending Gaussian index, as the reference's radix
// sort of (tile << 32 | depth bits) does).  count_host (pinned, optional) = [instances, unsorted, -, -]: a camera with more
// instances than `capacity`, or whose depths piled up in one bucket (unsorted != 0), is skipped as a whole -- render it again with
// g2pc_raster_front_cu / _back_cu.  G2PC_ERR_UNSUPPORTED: grids beyond 256 tiles per axis, or more Gaussians than the bucket sort
// pays for (~2 M): use the two-call path.  Follow with g2pc_raster_back_cu_tiles(phases = 4, num_instances = capacity).
size_t g2pc_raster_camera_cu_workspace(int64_t n, int64_t capacity, int32_t num_tiles) {
    using namespace g2pc;
    return align_up((size_t)n * 4) * 2 + bucket_sort_workspace((long)n) + g2pc_raster_back_workspace(capacity, num_tiles) + 4096;
}
int g2pc_raster_camera_cu(const G2pcCamera* cam, const float* means3D, const float* cov6, const float* opacity,
                          const float* colours_precomp, const float* shs, int32_t sh_degree, int32_t sh_coeffs,
                          const float* campos, const int32_t* mask, int64_t n, int64_t capacity, float* rec, uint32_t* rect,
                          int32_t* radii, int calculate_surface_distance, unsigned long long* cam_key, uint32_t* cam_surf,
                          float* out_color, float* out_depth, float* out_invdepth, uint32_t* count_host, int32_t tile_first,
                          int32_t tile_step, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(cam && means3D && cov6 && opacity && campos && rec && rect && radii && cam_key && cam_surf && out_color &&
                     out_depth && out_invdepth && ws && n > 0 && capacity > 0, G2PC_ERR_ARG, "bad arguments");
    G2PC_REQUIRE((colours_precomp != nullptr) != (shs != nullptr), G2PC_ERR_ARG, "provide exactly one of precomputed colours or SHs");
    G2PC_REQUIRE(!shs || (sh_degree >= 0 && sh_degree <= 3 && (sh_coeffs < 0 ? -sh_coeffs : sh_coeffs) >= (sh_degree + 1) * (sh_degree + 1) &&
                          (sh_coeffs > 0 || ((-sh_coeffs) & 3) == 0)), G2PC_ERR_ARG, "SH degree / coefficient count mismatch");
    const int W = cam->width, H = cam->height;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    G2PC_REQUIRE(!cu_wide_grid(gx, gy) && bucket_emit_supported((long)n) && capacity < (1ll << 31), G2PC_ERR_UNSUPPORTED,
                 "the fused camera call takes grids of up to 256 x 256 tiles and as many Gaussians as the bucket sort pays for");
    G2PC_REQUIRE(tile_step >= 1 && tile_first >= 0 && tile_first < tile_step, G2PC_ERR_ARG, "bad tile shard");
    const bool sharded = tile_step > 1;
    hipStream_t s = (hipStream_t)stream;
    const long L = capacity;
    Arena ar(ws, ws_bytes);
    uint32_t* key = ar.get<uint32_t>((size_t)n);
    uint32_t* idx = ar.get<uint32_t>((size_t)n);            // (the identity index the radix path sorts along: unused here)
    const size_t bucket_bytes = bucket_sort_workspace((long)n);
    char* bucket_ws = ar.get<char>(bucket_bytes);
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
    const int gshift = packed_instance_shift((long)n, T);
    const uint32_t gmask = gshift ? ((1u << gshift) - 1u) : 0xFFFFFFFFu;
    // the preprocess also clears the camera's visibility state and notes the depth range in the bucket sort's header
    // (until round 6: k_init_camera_state_cu, 68 us under load, and k_bk_minmax, 22 us, per camera)
#ifndef G2PC_CU_FOLD_INIT
// build-time A/B switch.  1: k_preprocess_cu also clears the camera's visibility state and notes the depth range in the bucket
// sort's header (k_init_camera_state_cu and k_bk_minmax leave the chain).  Measured on one box, alternating, configs[4]
// (profiles/r06d_cmd1.log): 30.64 / 30.65 / 30.30 ms with the fold against 30.47 / 30.19 / 29.73 without -- the two launches it
// removes cost less than the 12 bytes per Gaussian and the block-level atomics it adds to the SH-heavy preprocess.  Off.
#define G2PC_CU_FOLD_INIT 0
#endif
    const BucketPlan bplan = bucket_plan((long)n);
    BucketHdr* hdr = G2PC_CU_FOLD_INIT ? bucket_sort_header(bucket_ws) : nullptr;
    if (hdr) hipLaunchKernelGGL(k_bucket_hdr_init_cu, dim3(1), dim3(256), 0, s, hdr, bplan);
    hipLaunchKernelGGL(k_preprocess_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, to_cam(cam), gx, gy, means3D, cov6, opacity,
                       colours_precomp, shs, (int)sh_degree, (int)sh_coeffs, make_float3(campos[0], campos[1], campos[2]),
                       (long)n, key, idx, (uint32_t*)nullptr /* counts = rect areas */, (float4*)rec, rect, radii, 0, 0 /* antialiasing: g2pc_rasterize_gaussians only */,
                       hdr ? (unsigned long long*)cam_key : (unsigned long long*)nullptr, hdr ? (uint32_t*)cam_surf : (uint32_t*)nullptr,
                       hdr, bplan.nminmax);
    BucketEmit em{};
    em.weight = nullptr; em.rect = rect; em.inst_tile = inst_tile; em.inst_g = inst_g; em.gshift = gshift; em.nx = gx;
    em.capacity = (uint32_t)capacity; em.l_eff = l_eff; em.count_host = count_host;
    uint32_t* depth_overflow = nullptr;
    int rc = bucket_sort_u32(key, nullptr, nullptr, nullptr, (long)n, bucket_ws, bucket_bytes, &depth_overflow, s, Batch(), hdr != nullptr, false, &em);
    if (rc) return rc;
    if (mask || sharded) {                        // without a mask every pixel is written by the blend kernel
        hipMemsetAsync(out_color, 0, (size_t)3 * W * H * 4, s);
        hipMemsetAsync(out_depth, 0, (size_t)W * H * 4, s);
        hipMemsetAsync(out_invdepth, 0, (size_t)W * H * 4, s);
    }
    if (!hdr) hipLaunchKernelGGL(k_init_camera_state_cu, dim3(cdiv(n, RA_T)), dim3(RA_T), 0, s, (unsigned long long*)cam_key, (uint32_t*)cam_surf, (long)n);   // (else: cleared by k_preprocess_cu)
    rc = gshift ? sort_pairs_u32(inst_tile, nullptr, tile_sorted, nullptr, tile_tmp, nullptr, L, gshift,
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
    return check_launch("g2pc_raster_camera_cu");
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
// _C.rasterize_gaussians (rasterize_points.h:18-41, rasterize_points.cu:36-145 -> CudaRasterizer::Rasterizer::forward,
// rasterizer_impl.cu:197-352): include/g2pc.h, ABI 7.
int g2pc_rasterize_gaussians(const G2pcRasterizeArgs* a, const G2pcRasterizeOut* o, int32_t* num_rendered,
                             G2pcResizeFn geometry_buffer, void* geometry_user, G2pcResizeFn binning_buffer, void* binning_user,
                             G2pcResizeFn image_buffer, void* image_user, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(a && o && num_rendered && geometry_buffer && binning_buffer && image_buffer, G2PC_ERR_ARG, "bad arguments");
    const int64_t P = a->P;
    const int W = a->image_width, H = a->image_height;
    G2PC_REQUIRE(P >= 0 && W > 0 && H > 0, G2PC_ERR_ARG, "bad sizes");
    G2PC_REQUIRE(o->out_color && o->out_depth && o->out_invdepth, G2PC_ERR_ARG, "missing image outputs");
    hipStream_t s = (hipStream_t)stream;
    *num_rendered = 0;
    if (P == 0) {                                  // rasterize_points.cu:101: nothing is launched, the images stay zero
        hipMemsetAsync(o->out_color, 0, (size_t)3 * W * H * 4, s);
        hipMemsetAsync(o->out_depth, 0, (size_t)W * H * 4, s);
        hipMemsetAsync(o->out_invdepth, 0, (size_t)W * H * 4, s);
        return check_launch("g2pc_rasterize_gaussians");
    }
    G2PC_REQUIRE(a->background && a->means3D && a->opacity && a->viewmatrix && a->projmatrix && a->campos && o->radii &&
                     o->gauss_contributions && o->gauss_surface_distances && o->gauss_pixels,
                 G2PC_ERR_ARG, "bad arguments");
    // rasterizer_impl.cu:249-252 ("For non-RGB, provide precomputed Gaussian colors!") and the binding's exactly-one-of rules
    G2PC_REQUIRE((a->colors != nullptr) != (a->sh != nullptr), G2PC_ERR_ARG, "provide exactly one of precomputed colours or SHs");
    const bool own_cov = a->cov3D_precomp == nullptr;
    G2PC_REQUIRE(own_cov ? (a->scales && a->rotations) : (!a->scales && !a->rotations), G2PC_ERR_ARG,
                 "provide exactly one of a scale / rotation pair or precomputed 3D covariances");
    G2pcCamera cam{};
    for (int i = 0; i < 16; ++i) { cam.view[i] = a->viewmatrix[i]; cam.proj[i] = a->projmatrix[i]; }
    cam.tan_fovx = a->tan_fovx; cam.tan_fovy = a->tan_fovy; cam.width = W; cam.height = H;
    cam.bg[0] = a->background[0]; cam.bg[1] = a->background[1]; cam.bg[2] = a->background[2];
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    // geometry buffer: blend records, tile rectangles, depth order, instance offsets, own covariances, the front half's scratch
    const size_t front_bytes = g2pc_raster_front_workspace(P);
    const size_t geom_bytes = align_up((size_t)P * 64) + align_up((size_t)P * 8) + align_up((size_t)P * 4) + align_up((size_t)(P + 1) * 4) +
                              (own_cov ? align_up((size_t)P * 24) : 0) + align_up(front_bytes) + 256;
    void* geom = geometry_buffer(geometry_user, geom_bytes);
    G2PC_REQUIRE(geom, G2PC_ERR_WORKSPACE, "the geometry buffer callback returned no memory");
    Arena ga(geom, geom_bytes);
    float* rec = ga.get<float>((size_t)P * 16);
    uint32_t* rect = ga.get<uint32_t>((size_t)P * 2);
    uint32_t* sorted_idx = ga.get<uint32_t>((size_t)P);
    uint32_t* offsets = ga.get<uint32_t>((size_t)P + 1);
    float* cov6 = own_cov ? ga.get<float>((size_t)P * 6) : nullptr;
    char* front_ws = ga.get<char>(front_bytes);
    G2PC_REQUIRE(ga.ok(), G2PC_ERR_WORKSPACE, "geometry buffer too small");
    // image buffer: this camera's per-Gaussian visibility state (packed (contribution, ~pixel) keys, surface distances)
    const size_t img_bytes = align_up((size_t)P * 8) + align_up((size_t)P * 4) + 256;
    void* img = image_buffer(image_user, img_bytes);
    G2PC_REQUIRE(img, G2PC_ERR_WORKSPACE, "the image buffer callback returned no memory");
    Arena ia(img, img_bytes);
    unsigned long long* cam_key = ia.get<unsigned long long>((size_t)P);
    uint32_t* cam_surf = ia.get<uint32_t>((size_t)P);
    if (own_cov)
        hipLaunchKernelGGL(k_cov3d_cu, dim3(cdiv(P, RA_T)), dim3(RA_T), 0, s, a->scales, a->rotations, a->scale_modifier, (long)P, cov6);
    int rc = front_cu_impl(&cam, a->means3D, own_cov ? cov6 : a->cov3D_precomp, a->opacity, a->colors, a->sh, a->sh ? a->degree : 0,
                           a->sh ? a->M : 0, a->campos, P, rec, rect, o->radii, sorted_idx, offsets, nullptr, front_ws, front_bytes,
                           stream, a->antialiasing ? 1 : 0);
    if (rc) return rc;
    uint32_t L = 0;                                    // rasterizer_impl.cu:289: the one blocking read-back
    if (hipMemcpyAsync(&L, offsets + P, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        set_error(__func__, "reading the instance count back failed");
        return G2PC_ERR_LAUNCH;
    }
    *num_rendered = (int32_t)L;
    const size_t bin_bytes = g2pc_raster_back_workspace((int64_t)L, T);
    void* bin = binning_buffer(binning_user, bin_bytes);
    G2PC_REQUIRE(bin, G2PC_ERR_WORKSPACE, "the binning buffer callback returned no memory");
    rc = g2pc_raster_back_cu_tiles(&cam, a->mask, P, (int64_t)L, rec, rect, sorted_idx, offsets, a->calculate_surface_distance ? 1 : 0,
                                   cam_key, cam_surf, o->out_color, o->out_depth, o->out_invdepth, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, 0, nullptr, nullptr, nullptr, 3, 0, 1, bin, bin_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_camera_cu, dim3(cdiv(P, RA_T)), dim3(RA_T), 0, s, cam_key, cam_surf, (long)P, o->gauss_contributions,
                       o->gauss_surface_distances, o->gauss_pixels);
    if (a->debug && hipStreamSynchronize(s) != hipSuccess) {           // CHECK_CUDA(..., debug), auxiliary.h:178-185
        set_error(__func__, "a kernel of the rasterisation failed (debug)");
        return G2PC_ERR_LAUNCH;
    }
    return check_launch("g2pc_rasterize_gaussians");
}
}
