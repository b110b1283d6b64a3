// EXPERIMENTS ONLY (-DG2PC_EXPERIMENTS; not part of libg2pc.so): python-semantics blend kernels that were built to parity,
// measured and NOT selected: the scalar kernel for 1 or 4 sub-blocks per wave (k_blend_py), the
// scalar-gather form (k_blend_py_sg) and the two-wave form (k_blend_py_2w) of the dual-list kernel.  Included by raster.hip
// inside namespace g2pc, after k_blend_py_dl; selected through g2pc_set_blend_variant (experiments/knobs.inl) or a layout
// with chunk_subblocks 1 / 4.
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

