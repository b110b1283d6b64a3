// g2pc device primitives for gfx950 (wave64): error plumbing, exclusive scan, stable LSD radix sort.
//
// These replace the reference's CUB calls (cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:285;
// cub::DeviceRadixSort::SortPairs, rasterizer_impl.cu:311-316) and the torch.unique / boolean-index
// compactions of the sampler (gauss_to_pc.py:225-238,334).  Multi-kernel (histogram -> scan ->
// scatter) formulations are used on purpose: the per-XCD L2s of MI355X are not coherent, so
// single-pass look-back schemes would need agent-scope acquire/release per tile.
#include "g2pc_internal.h"

namespace g2pc {

static thread_local std::string g_err;
void set_error(const char* where, const char* what) { g_err = std::string(where) + ": " + what; }
int check_launch(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(where, hipGetErrorString(e)); return G2PC_ERR_LAUNCH; }
    return G2PC_OK;
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan.  Block = 256 threads x 4 items.  Recursive over block sums.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_T = 256, SCAN_I = 4, SCAN_TILE = SCAN_T * SCAN_I;

// gather (optional): scan in[gather[i]] instead of in[i] (the rasteriser's tiles-touched counts in depth order)
__global__ __launch_bounds__(SCAN_T) void k_scan_tile(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     long n, uint32_t* __restrict__ block_sums,
                                                     const uint32_t* __restrict__ gather, size_t cs) {
    __shared__ uint32_t wsum[SCAN_T / kWave];
    in = seg(in, cs); out = seg(out, cs); block_sums = seg(block_sums, cs); gather = seg(gather, cs);
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_I;
    uint32_t v[SCAN_I];
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        v[i] = (base + i < n) ? (gather ? in[gather[base + i]] : in[base + i]) : 0u;
        tsum += v[i];
    }
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(tsum);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / kWave; ++i) {
        uint32_t s = wsum[i];
        if (i < (int)w) woff += s;
        total += s;
    }
    uint32_t run = woff + incl - tsum;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        if (gridDim.x == 1) out[n] = total;                 // single tile: the scan is complete
    }
}

// second (and last) kernel of the 2-kernel scan: every block sums the block totals in front of it by itself
// (nb <= 2048 values, L2 resident) instead of a separate scan-of-sums launch; the last block also writes the total.
__global__ __launch_bounds__(SCAN_T) void k_scan_add_self(uint32_t* __restrict__ out, long n,
                                                         const uint32_t* __restrict__ block_sums, size_t cs) {
    __shared__ uint32_t wsum[SCAN_T / kWave];
    out = seg(out, cs); block_sums = seg(block_sums, cs);
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_I;
    uint32_t v[SCAN_I];                      // (requested before the block totals are summed: one round of loads, not two)
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) v[i] = (base + i < n) ? out[base + i] : 0u;
    uint32_t acc = 0;
    for (unsigned i = threadIdx.x; i < blockIdx.x; i += SCAN_T) acc += block_sums[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / kWave; ++i) off += wsum[i];
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
        if (base + i < n) out[base + i] = v[i] + off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = off + block_sums[blockIdx.x];
}

__global__ __launch_bounds__(SCAN_T) void k_scan_add(uint32_t* __restrict__ out, long n,
                                                    const uint32_t* __restrict__ block_offs, uint32_t* total_slot) {
    const uint32_t off = block_offs[blockIdx.x];
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_I;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
        if (base + i < n) out[base + i] += off;
    if (total_slot && blockIdx.x == 0 && threadIdx.x == 0) *total_slot = block_offs[gridDim.x];
}

// Batched form: blockIdx.y selects one of `rows` independent scans of n values each (in + row * n -> out + row * (n+1)),
// two launches for all of them (the sampler's per-attempt count scans).  nb <= 2048 tiles per row.
__global__ __launch_bounds__(SCAN_T) void k_scan_tile_rows(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          long n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[SCAN_T / kWave];
    const long row = blockIdx.y;
    in += row * n; out += row * (n + 1); block_sums += row * gridDim.x;
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_I;
    uint32_t v[SCAN_I];
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        tsum += v[i];
    }
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(tsum);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / kWave; ++i) {
        uint32_t s = wsum[i];
        if (i < (int)w) woff += s;
        total += s;
    }
    uint32_t run = woff + incl - tsum;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        if (gridDim.x == 1) out[n] = total;
    }
}
__global__ __launch_bounds__(SCAN_T) void k_scan_add_self_rows(uint32_t* __restrict__ out, long n,
                                                              const uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[SCAN_T / kWave];
    const long row = blockIdx.y;
    out += row * (n + 1); block_sums += row * gridDim.x;
    uint32_t acc = 0;
    for (unsigned i = threadIdx.x; i < blockIdx.x; i += SCAN_T) acc += block_sums[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / kWave; ++i) off += wsum[i];
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_I;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
        if (base + i < n) out[base + i] += off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = off + block_sums[blockIdx.x];
}

size_t scan_rows_workspace(long n, int rows) {
    long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb < 1) nb = 1;
    return align_up((size_t)nb * rows * sizeof(uint32_t)) + 256;
}

// rows independent exclusive scans; returns G2PC_ERR_UNSUPPORTED when a row needs more than 2048 tiles (the caller then
// falls back to one scan_exclusive_u32 per row)
int scan_exclusive_rows_u32(const uint32_t* in, uint32_t* out, long n, int rows, void* ws, size_t ws_bytes, hipStream_t s) {
    if (n <= 0 || rows <= 0) return G2PC_OK;
    long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > 2048) return G2PC_ERR_UNSUPPORTED;
    Arena ar(ws, ws_bytes);
    uint32_t* sums = ar.get<uint32_t>((size_t)nb * rows);
    if (!ar.ok()) { set_error("scan_rows", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    hipLaunchKernelGGL(k_scan_tile_rows, dim3((unsigned)nb, (unsigned)rows), dim3(SCAN_T), 0, s, in, out, n, sums);
    if (nb > 1)
        hipLaunchKernelGGL(k_scan_add_self_rows, dim3((unsigned)nb, (unsigned)rows), dim3(SCAN_T), 0, s, out, n, sums);
    return check_launch("scan_rows");
}

size_t scan_workspace(long n) {
    size_t bytes = 0;
    long m = n;
    while (true) {
        long nb = (m + SCAN_TILE - 1) / SCAN_TILE;
        if (nb < 1) nb = 1;
        bytes += align_up((size_t)(nb + 1) * sizeof(uint32_t));
        if (nb == 1) break;
        m = nb;
    }
    return bytes + 256;
}

static int scan_rec(const uint32_t* in, uint32_t* out, long n, Arena& ar, hipStream_t s,
                    const uint32_t* gather = nullptr, Batch b = Batch()) {
    long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb < 1) nb = 1;
    uint32_t* sums = ar.get<uint32_t>((size_t)nb + 1);
    if (!ar.ok()) { set_error("scan", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    hipLaunchKernelGGL(k_scan_tile, dim3((unsigned)nb, (unsigned)b.n), dim3(SCAN_T), 0, s, in, out, n, sums, gather, b.cs);
    if (nb == 1) return check_launch("scan");
    if (nb <= 2048) {
        hipLaunchKernelGGL(k_scan_add_self, dim3((unsigned)nb, (unsigned)b.n), dim3(SCAN_T), 0, s, out, n, sums, b.cs);
        return check_launch("scan");
    }
    if (b.n > 1) { set_error("scan", "batched scans support at most 2M values per camera"); return G2PC_ERR_UNSUPPORTED; }
    int rc = scan_rec(sums, sums, nb, ar, s);   // sums[0..nb] = exclusive offsets (+ total)
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_T), 0, s, out, n, sums, out + n);
    return check_launch("scan");
}

int scan_exclusive_u32(const uint32_t* in, uint32_t* out, long n, void* ws, size_t ws_bytes, hipStream_t s,
                       const uint32_t* gather, Batch b) {
    if (n <= 0) {
        if (hipMemsetAsync(out, 0, sizeof(uint32_t), s) != hipSuccess) { set_error("scan", "memset failed"); return G2PC_ERR_LAUNCH; }
        return G2PC_OK;
    }
    Arena ar(ws, ws_bytes);
    return scan_rec(in, out, n, ar, s, gather, b);
}

// ------------------------------------------------------------------------------------------------
// Stable LSD radix sort, digits of up to 8 or up to 11 bits.  Tile = 256 threads x ITEMS keys; wave w of a block
// owns a contiguous quarter of the tile and walks it in ITEMS rounds of 64 keys, ranking
// with wave64 ballots (match-any over the digit bits) against a wave-private LDS histogram row.
// ------------------------------------------------------------------------------------------------
constexpr int RS_T = 256, RS_W = RS_T / kWave;

template <int BITS, int ITEMS>
__global__ __launch_bounds__(RS_T) void k_radix_hist(const uint32_t* __restrict__ keys, long n, int shift,
                                                    unsigned mask, uint32_t* __restrict__ ghist, unsigned nb,
                                                    const uint32_t* __restrict__ n_dev, size_t cs) {
    constexpr int BINS = 1 << BITS, TILE = RS_T * ITEMS;
    __shared__ uint32_t hist[BINS];
    keys = seg(keys, cs); ghist = seg(ghist, cs); n_dev = seg(n_dev, cs);
    if (n_dev) { const long m = (long)*n_dev; n = m < n ? m : n; }      // device-side count (n = the launch capacity)
    for (int i = threadIdx.x; i < BINS; i += RS_T) hist[i] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * TILE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        long idx = base + (long)i * RS_T + threadIdx.x;
        if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (unsigned d = threadIdx.x; d <= mask; d += RS_T) ghist[(size_t)d * nb + blockIdx.x] = hist[d];
}

// One launch instead of the generic two-kernel scan over the (digit, block) matrix: block d turns row d of ghist (the
// per-tile counts of digit d, nb entries) into its exclusive prefix IN PLACE and leaves the row total in gtot[d]; the
// scatter kernel adds the exclusive scan of the <= 2048 row totals itself (a few hundred LDS operations per block).
__global__ __launch_bounds__(SCAN_T) void k_radix_rowscan(uint32_t* __restrict__ ghist, unsigned nb, uint32_t* __restrict__ gtot,
                                                         size_t cs) {
    __shared__ uint32_t wsum[SCAN_T / kWave];
    __shared__ uint32_t carry;
    ghist = seg(ghist, cs); gtot = seg(gtot, cs);
    uint32_t* row = ghist + (size_t)blockIdx.x * nb;
    constexpr unsigned RSC_PER = 24;          // rows of up to SCAN_T * RSC_PER tiles: every thread takes `per` consecutive entries,
    if (nb <= SCAN_T * RSC_PER) {             // all requested at once -- one barrier instead of two per 256 entries
        const unsigned per = (nb + SCAN_T - 1) / SCAN_T, first = threadIdx.x * per;
        uint32_t v[RSC_PER], tsum = 0;
#pragma unroll
        for (unsigned k = 0; k < RSC_PER; ++k) {
            v[k] = (k < per && first + k < nb) ? row[first + k] : 0u;
            tsum += v[k];
        }
        const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t incl = wave_incl_scan_u32(tsum);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < SCAN_T / kWave; ++k) { const uint32_t t = wsum[k]; if (k < (int)w) woff += t; total += t; }
        uint32_t run = woff + incl - tsum;
#pragma unroll
        for (unsigned k = 0; k < RSC_PER; ++k) {
            if (k < per && first + k < nb) row[first + k] = run;
            run += v[k];
        }
        if (threadIdx.x == 0) gtot[blockIdx.x] = total;
        return;
    }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (unsigned base = 0; base < nb; base += SCAN_T) {
        const unsigned i = base + threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0u;
        const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t incl = wave_incl_scan_u32(v);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < SCAN_T / kWave; ++k) { const uint32_t t = wsum[k]; if (k < (int)w) woff += t; total += t; }
        const uint32_t c = carry;
        if (i < nb) row[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) gtot[blockIdx.x] = carry;
}

template <int BITS, int ITEMS>
__global__ __launch_bounds__(RS_T) void k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                       const uint32_t* __restrict__ vals_in,
                                                       uint32_t* __restrict__ keys_out,
                                                       uint32_t* __restrict__ vals_out, long n, int shift,
                                                       unsigned mask, int nbits,
                                                       const uint32_t* __restrict__ goffs, unsigned nb,
                                                       const uint32_t* __restrict__ n_dev,
                                                       const uint32_t* __restrict__ gtot, size_t cs) {
    constexpr int BINS = 1 << BITS, TILE = RS_T * ITEMS;
    __shared__ uint32_t whist[RS_W][BINS];      // per-wave running digit counts, then exclusive offsets
    __shared__ uint32_t gbase[BINS];
    __shared__ uint32_t dsum[RS_T / kWave];
    keys_in = seg(keys_in, cs); vals_in = seg(vals_in, cs); keys_out = seg(keys_out, cs); vals_out = seg(vals_out, cs);
    goffs = seg(goffs, cs); n_dev = seg(n_dev, cs); gtot = seg(gtot, cs);
    if (n_dev) {
        const long m = (long)*n_dev;
        n = m < n ? m : n;
        if ((long)blockIdx.x * TILE >= n) return;                        // idle tail block of a capacity-sized launch
    }
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the tile's keys first: their loads are in flight while the digit bases are put together (two more dependent rounds of
    // loads and a barrier) -- beside other kernels a block of this one has little else to hide a round trip behind
    const long wbase = (long)blockIdx.x * TILE + (long)w * (TILE / RS_W);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const long idx = wbase + (long)r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = (valid && vals_in) ? vals_in[idx] : 0u;          // vals_in == nullptr: keys only
    }
    for (int i = threadIdx.x; i < RS_W * BINS; i += RS_T) (&whist[0][0])[i] = 0;
    {   // gbase[d] = (number of keys with a smaller digit) + (keys with digit d in the tiles before this one):
        // exclusive scan of the row totals, BINS / RS_T digits per thread
        constexpr int PER = (BINS + RS_T - 1) / RS_T;
        uint32_t tv[PER], tsum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const unsigned d = threadIdx.x * PER + k;
            tv[k] = d <= mask ? gtot[d] : 0u;
            tsum += tv[k];
        }
        const uint32_t incl = wave_incl_scan_u32(tsum);
        if (lane == 63) dsum[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int k = 0; k < RS_T / kWave; ++k) if (k < (int)w) woff += dsum[k];
        uint32_t run = woff + incl - tsum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const unsigned d = threadIdx.x * PER + k;
            if (d <= mask) gbase[d] = run + goffs[(size_t)d * nb + blockIdx.x];
            run += tv[k];
        }
    }
    __syncthreads();

    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        long idx = wbase + (long)r * 64 + lane;
        bool valid = idx < n;
        unsigned d = (key[r] >> shift) & mask;
        // match-any: lanes holding the same digit
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        unsigned cnt = __popcll(peers);
        unsigned before = __popcll(peers & lt);
        unsigned basec = valid ? whist[w][d] : 0u;
        wave_sync();
        if (valid && before == 0) whist[w][d] = basec + cnt;
        wave_sync();
        rank[r] = basec + before;
    }
    __syncthreads();
    // exclusive prefix over waves for every digit
    for (unsigned d = threadIdx.x; d <= mask; d += RS_T) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < RS_W; ++i) { uint32_t c = whist[i][d]; whist[i][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        long idx = wbase + (long)r * 64 + lane;
        if (idx < n) {
            unsigned d = (key[r] >> shift) & mask;
            uint32_t pos = gbase[d] + whist[w][d] + rank[r];
            keys_out[pos] = key[r];
            if (vals_out) vals_out[pos] = val[r];
        }
    }
}

// pass geometry: digits of up to 8 bits, 4 keys per thread for inputs that would otherwise leave CUs idle (<= 2M keys), else 8.
// (Wider digits -- 11 bits, or ONE 10-bit pass for the tile sort of up to 1 024 leaves -- were measured in rounds 2-4 and lost:
// they exist in -DG2PC_EXPERIMENTS builds only, behind g2pc_set_sort_tuning.)
#ifdef G2PC_EXPERIMENTS
static int g_radix_wide_bits = 8;          // 8 or 11: digit width used when more than 8 bits are sorted; 10: ONE
                                           // 10-bit pass for 9 - 10 bits (the tile sort of up to 1 024 leaves), else as 8
static long g_radix_small_n = 2L << 20;    // inputs up to this size use 4 keys per thread
#else
constexpr int g_radix_wide_bits = 8;
constexpr long g_radix_small_n = 2L << 20;
#endif
#ifndef G2PC_RADIX10_ITEMS
#define G2PC_RADIX10_ITEMS 8      // experiments: keys per thread of the ONE-pass 10-bit tile sort (16: 4 096 keys per block against its 1 024 bins)
#endif
static inline int radix_items(long n) { return n <= g_radix_small_n ? 4 : (g_radix_wide_bits == 10 ? G2PC_RADIX10_ITEMS : 8); }
static inline int radix_maxbits(int total_bits) {
    if (total_bits <= 8) return 8;
    if (g_radix_wide_bits == 10) return total_bits <= 10 ? 10 : 8;
    return g_radix_wide_bits;
}

size_t sort_workspace(long n) {
    long nb = (n + RS_T * 4 - 1) / (RS_T * 4);
    if (nb < 1) nb = 1;
    size_t entries = (size_t)2048 * nb + 1;
    return align_up(entries * sizeof(uint32_t)) + scan_workspace((long)entries) + 2048 * sizeof(uint32_t) + 512;
}

template <int BITS, int ITEMS>
static void radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, long n, int bit,
                       int nbits, uint32_t* ghist, unsigned nb, char* scan_ws, size_t scan_bytes, hipStream_t s, int& rc,
                       const uint32_t* n_dev, Batch b) {
    unsigned mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL((k_radix_hist<BITS, ITEMS>), dim3(nb, (unsigned)b.n), dim3(RS_T), 0, s, kin, n, bit, mask, ghist, nb, n_dev, b.cs);
    uint32_t* gtot = (uint32_t*)scan_ws;                       // (mask + 1) row totals
    if (scan_bytes < (size_t)(mask + 1) * sizeof(uint32_t)) { set_error("sort", "workspace too small"); rc = G2PC_ERR_WORKSPACE; return; }
    hipLaunchKernelGGL(k_radix_rowscan, dim3(mask + 1, (unsigned)b.n), dim3(SCAN_T), 0, s, ghist, nb, gtot, b.cs);
    hipLaunchKernelGGL((k_radix_scatter<BITS, ITEMS>), dim3(nb, (unsigned)b.n), dim3(RS_T), 0, s, kin, vin, kout, vout, n, bit, mask, nbits,
                       ghist, nb, n_dev, gtot, b.cs);
}

int sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                   uint32_t* keys_tmp, uint32_t* vals_tmp, long n, int bit_lo, int bit_hi, void* ws,
                   size_t ws_bytes, hipStream_t s, const uint32_t* n_dev, Batch b) {
    // n_dev (optional, device): the number of keys actually present (<= n).  The launch geometry then depends on n
    // (a capacity) only, so the same sequence of launches -- e.g. a captured hipGraph -- serves any count.
    if (n <= 0) return G2PC_OK;
    // vals_in == vals_out == vals_tmp == nullptr: keys only (half the traffic of a pass)
    const bool keys_only = !vals_in && !vals_out && !vals_tmp;
    if (keys_tmp == keys_in || keys_tmp == keys_out || keys_out == keys_in ||
        (!keys_only && (!vals_in || !vals_out || !vals_tmp || vals_tmp == vals_in || vals_tmp == vals_out || vals_out == vals_in))) {
        set_error("sort", "in / out / tmp buffers must be distinct");
        return G2PC_ERR_ARG;
    }
    int total_bits = bit_hi - bit_lo;
    if (total_bits <= 0) {
        if (b.n > 1) { set_error("sort", "batched sort of zero bits"); return G2PC_ERR_UNSUPPORTED; }
        hipMemcpyAsync(keys_out, keys_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        if (!keys_only) hipMemcpyAsync(vals_out, vals_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        return G2PC_OK;
    }
    const int maxbits = radix_maxbits(total_bits), items = radix_items(n);
    const int passes = (total_bits + maxbits - 1) / maxbits;
    const long tile = (long)RS_T * items;
    const unsigned nb = (unsigned)((n + tile - 1) / tile);
    Arena ar(ws, ws_bytes);
    size_t entries = ((size_t)1 << maxbits) * nb + 1;
    uint32_t* ghist = ar.get<uint32_t>(entries);
    size_t scan_bytes = scan_workspace((long)entries) + 2048 * sizeof(uint32_t);     // holds the per-digit row totals
    char* scan_ws = ar.get<char>(scan_bytes);
    if (!ar.ok()) { set_error("sort", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    // ping-pong so that the last pass lands in *_out
    const uint32_t* kin = keys_in;
    const uint32_t* vin = vals_in;
    int bit = bit_lo, rc = G2PC_OK;
    for (int p = 0; p < passes; ++p) {
        int nbits = total_bits / passes + (p < total_bits % passes ? 1 : 0);
        bool to_out = ((passes - 1 - p) % 2) == 0;
        uint32_t* kout = to_out ? keys_out : keys_tmp;
        uint32_t* vout = to_out ? vals_out : vals_tmp;
        if (maxbits == 8) {
            if (items == 4) radix_pass<8, 4>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
            else radix_pass<8, 8>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
        }
#ifdef G2PC_EXPERIMENTS
        else if (maxbits == 10) {
            if (items == 4) radix_pass<10, 4>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
            else if (items == 16) radix_pass<10, 16>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
            else radix_pass<10, 8>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
        } else {
            if (items == 4) radix_pass<11, 4>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
            else radix_pass<11, 8>(kin, vin, kout, vout, n, bit, nbits, ghist, nb, scan_ws, scan_bytes, s, rc, n_dev, b);
        }
#endif
        if (rc) return rc;
        kin = kout; vin = vout;
        bit += nbits;
    }
    return check_launch("sort");
}

// ------------------------------------------------------------------------------------------------
// Bucket sort of (key, input position) pairs -- the depth order of one camera in six launches instead of the twelve
// of a four-pass LSD radix sort, every key moved ONCE.  Keys are bit patterns of positive floats (monotone as integers)
// and spread smoothly over their range, so one most-significant-digit pass into `nbk` range-normalised buckets
// (~BK_AVG keys each) leaves buckets that a bitonic sort of 64-bit (key, position) composites finishes entirely in LDS
// (32 KB of the CU's 160).  The composite makes the result THE stable ascending order (ties in input order), whatever
// order the scatter left inside a bucket.  Keys 0xFFFFFFFF ("not on screen") go to a tail bucket that is copied, not
// sorted: entries with that key come out in UNSPECIFIED (run-dependent) order -- the rasteriser never reads them
// (tiles touched = 0); a caller that needs them ordered sorts with sort_pairs_u32.  A bucket with more than its room (1024 keys; 4096 when the mean bucket exceeds 256: depths piled up in 1/nbk of their range) raises
// `overflow`: the caller repeats the sort with the radix path.
//   k_bk_minmax   per-block (~min, max) partials
//   k_bk_hist     chunk c of `kpb` keys -> LDS histogram -> column c of table[bucket][chunk]
//   k_bk_colscan  one wave per bucket: exclusive scan of its row of the table, bucket total -> count[]
//   k_bk_scan     one block: exclusive scan of the bucket totals -> start[], overflow test
//   k_bk_scatter  chunk c again: LDS cursors seeded with start[b] + table[b][c]; items[pos] = (key, position)
//   k_bk_sort     one wave per bucket: bitonic sort in LDS, gather of the values
// No global atomics and no memset: a first version reserved places with global atomicAdd cursors (1 M returning atomics
// on 2 K addresses: 0.66 ms per camera, and every co-running kernel slowed 4x) and cleared its header with a
// hipMemsetAsync, whose graph node this runtime did not order against the kernels around it (replays faulted).
// ------------------------------------------------------------------------------------------------
#ifndef G2PC_BK_EMIT_RMAX
#define G2PC_BK_EMIT_RMAX 16      // build-time A/B switch (k_bk_sort_emit): 8 = buckets beyond 512 items through LDS, half the VGPRs
#endif
constexpr int BK_UNROLL = 8;      // keys a lane of k_bk_hist / k_bk_scatter requests before it uses the first
constexpr int BK_MIN = 1024, BK_AVG = 256, BK_CAP_SMALL = 1024, BK_CAP_LARGE = 4096, BK_T = 256, BK_MAXCHUNKS = 512, BK_TAILBLOCKS = 256;
BucketPlan bucket_plan(long n) {
    BucketPlan p;
    p.nbk = BK_MIN;
    while (p.nbk < (uint32_t)BK_MAX && (long)p.nbk * BK_AVG < n) p.nbk <<= 1;
#ifndef G2PC_BK_KPB
#define G2PC_BK_KPB 8192
#endif
    long kpb = G2PC_BK_KPB;                         // keys per chunk: at most BK_MAXCHUNKS chunks (the table is nbk x nchunks)
    while (kpb * BK_MAXCHUNKS < n) kpb += 1024;
    // room of one bucket in the in-LDS sort: 4x the mean while that fits the small footprint (8 KB per wave, which
    // squeezes in beside the blend waves of the other cameras), else the large one
    p.cap = (n + p.nbk - 1) / p.nbk * 4 <= BK_CAP_SMALL ? BK_CAP_SMALL : BK_CAP_LARGE;
    p.kpb = (uint32_t)kpb;
    p.nchunks = (uint32_t)cdiv(n, kpb);
    long nm = cdiv(n, BK_T * 16);
    p.nminmax = (uint32_t)(nm > BK_PARTIALS ? BK_PARTIALS : nm);
    return p;
}
struct BucketMap {                                  // bucket = (key - kmin) * nbk / (span + 1), as a 32.32 fixed-point multiply
    uint32_t kmin, nbk; unsigned long long mul;
    __device__ __forceinline__ void set(uint32_t neg_kmin, uint32_t kmax, uint32_t nbk_) {
        kmin = ~neg_kmin; nbk = nbk_;
        const unsigned long long span1 = kmax >= kmin ? (unsigned long long)(kmax - kmin) + 1ull : 1ull;
        mul = ((unsigned long long)nbk << 32) / span1;
    }
    __device__ __forceinline__ unsigned of(uint32_t key) const {
        if (key == 0xFFFFFFFFu) return nbk;
        return (unsigned)(((unsigned long long)(key - kmin) * mul) >> 32);
    }
};
// block-wide reduction of the minmax partials (every thread returns the same map)
__device__ __forceinline__ BucketMap bucket_map_from_partials(const BucketHdr* __restrict__ h, uint32_t nminmax, uint32_t* lds2) {
    uint32_t a = 0, b = 0;
    for (uint32_t i = threadIdx.x; i < nminmax; i += blockDim.x) { a = umax_(a, h->partial[2 * i]); b = umax_(b, h->partial[2 * i + 1]); }
    a = wave_max_u32(a); b = wave_max_u32(b);
    if (threadIdx.x == 0) { lds2[0] = 0; lds2[1] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMax(&lds2[0], a); atomicMax(&lds2[1], b); }
    __syncthreads();
    BucketMap m;
    m.set(lds2[0], lds2[1], h->nbk);
    return m;
}
__global__ __launch_bounds__(BK_T) void k_bk_minmax(const uint32_t* __restrict__ keys, long n, BucketHdr* __restrict__ h, BucketPlan plan,
                                                   size_t cs) {
    __shared__ uint32_t red[2];
    keys = seg(keys, cs); h = seg(h, cs);
    uint32_t a = 0, b = 0;
    for (long i = (long)blockIdx.x * BK_T + threadIdx.x; i < n; i += (long)gridDim.x * BK_T) {
        const uint32_t k = keys[i];
        if (k != 0xFFFFFFFFu) { a = umax_(a, ~k); b = umax_(b, k); }
    }
    a = wave_max_u32(a); b = wave_max_u32(b);
    if (threadIdx.x == 0) { red[0] = 0; red[1] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMax(&red[0], a); atomicMax(&red[1], b); }
    __syncthreads();
    if (threadIdx.x == 0) {
        h->partial[2 * blockIdx.x] = red[0]; h->partial[2 * blockIdx.x + 1] = red[1];
        if (blockIdx.x == 0) { h->overflow = 0; h->nbk = plan.nbk; h->nchunks = plan.nchunks; h->cap = plan.cap; }
    }
}
__global__ __launch_bounds__(BK_T) void k_bk_hist(const uint32_t* __restrict__ keys, long n, const BucketHdr* __restrict__ h,
                                                 uint32_t* __restrict__ table, BucketPlan plan, size_t cs) {
    __shared__ uint32_t lh[BK_MAX + 1];
    __shared__ uint32_t red[2];
    keys = seg(keys, cs); h = seg(h, cs); table = seg(table, cs);
    const uint32_t nbk = plan.nbk;
    for (uint32_t i = threadIdx.x; i <= nbk; i += BK_T) lh[i] = 0;
    const BucketMap m = bucket_map_from_partials(h, plan.nminmax, red);   // (its barriers also publish the zeroed histogram)
    const long base = (long)blockIdx.x * plan.kpb;
    for (uint32_t j0 = threadIdx.x; j0 < plan.kpb; j0 += BK_T * BK_UNROLL) {      // BK_UNROLL loads in flight per lane
        uint32_t k[BK_UNROLL];
        bool ok[BK_UNROLL];
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * BK_T;
            const long i = base + j;
            ok[u] = j < plan.kpb && i < n;
            k[u] = ok[u] ? keys[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u)
            if (ok[u]) atomicAdd(&lh[m.of(k[u])], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= nbk; i += BK_T) table[(size_t)i * plan.nchunks + blockIdx.x] = lh[i];
}
__device__ __forceinline__ uint32_t rect_area(uint32_t rc) {       // tiles of ix0 | ix1 << 8 | iy0 << 16 | iy1 << 24 (inclusive)
    return (((rc >> 8) & 255u) - (rc & 255u) + 1u) * ((rc >> 24) - ((rc >> 16) & 255u) + 1u);
}
// Fused emission (BucketEmit): the same histogram with the keys' WEIGHTS summed per bucket beside the counts -- one 64-bit LDS
// add per key (count << 32 | weight: a chunk holds at most 2^16 keys and its weights sum to less than 2^32, the capacity
// test in k_bk_scan sees to the rest) --, the weight sums written row-wise (wtable[chunk][bucket], coalesced; they are only
// ever summed over the chunks).  reversed: the keys are in reversed index order, position j carries Gaussian n - 1 - j.  NBK sizes the LDS
// table (8 bytes per bucket: 32 KB at the 4 096 buckets of a 1 M-key sort, what the plain histogram takes).
template <int NBK>
__global__ __launch_bounds__(BK_T) void k_bk_hist_w(const uint32_t* __restrict__ keys, long n, const BucketHdr* __restrict__ h,
                                                   uint32_t* __restrict__ table, uint32_t* __restrict__ wtable,
                                                   const uint32_t* __restrict__ weight, const uint32_t* __restrict__ rect,
                                                   BucketPlan plan, size_t cs, bool reversed) {
    __shared__ unsigned long long lh[NBK + 1];
    __shared__ uint32_t red[2];
    // the <BK_MAX> instance (1 048 576 < n <= 2 097 152 keys) holds 64 KB + 16 B of LDS: a gfx950 size (160 KB per CU; two blocks
    // per CU either way, 3 x 64 KB would not fit) -- this library is built for gfx950 only
    static_assert(sizeof(unsigned long long) * (NBK + 1) + 8 <= 160 * 1024 / 2, "k_bk_hist_w: two blocks per CU must fit gfx950's LDS");
    keys = seg(keys, cs); h = seg(h, cs); table = seg(table, cs); wtable = seg(wtable, cs); weight = seg(weight, cs); rect = seg(rect, cs);
    const uint32_t nbk = plan.nbk;
    for (uint32_t i = threadIdx.x; i <= nbk; i += BK_T) lh[i] = 0ull;
    const BucketMap m = bucket_map_from_partials(h, plan.nminmax, red);   // (its barriers also publish the zeroed histogram)
    const long base = (long)blockIdx.x * plan.kpb;
    for (uint32_t j0 = threadIdx.x; j0 < plan.kpb; j0 += BK_T * BK_UNROLL) {      // BK_UNROLL loads in flight per lane
        uint32_t k[BK_UNROLL], wt[BK_UNROLL];
        bool ok[BK_UNROLL];
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * BK_T;
            const long i = base + j;
            ok[u] = j < plan.kpb && i < n;
            k[u] = ok[u] ? keys[i] : 0u;
            // weight == nullptr: the rect's area (every tile of the rectangle takes an instance; keys 0xFFFFFFFF carry rect 0 = one
            // tile, but land in the tail bucket, whose weight sum nobody reads)
            wt[u] = ok[u] ? (weight ? weight[reversed ? n - 1 - i : i] : rect_area(rect[reversed ? n - 1 - i : i])) : 0u;
        }
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u)
            if (ok[u]) atomicAdd(&lh[m.of(k[u])], (1ull << 32) | (unsigned long long)wt[u]);
    }
    __syncthreads();
    uint32_t* wrow = wtable + (size_t)blockIdx.x * (nbk + 1);
    for (uint32_t i = threadIdx.x; i <= nbk; i += BK_T) {
        const unsigned long long v = lh[i];
        table[(size_t)i * plan.nchunks + blockIdx.x] = (uint32_t)(v >> 32);
        wrow[i] = (uint32_t)v;
    }
}
// one wave per bucket (4 per block): table row -> exclusive prefix over the chunks, row total -> count[bucket]
// wtable != nullptr (fused emission): the bucket's weight sum over the chunks -> wstart[bucket] (scanned by k_bk_scan)
__global__ __launch_bounds__(BK_T) void k_bk_colscan(BucketHdr* __restrict__ h, uint32_t* __restrict__ table, BucketPlan plan,
                                                    size_t cs, const uint32_t* __restrict__ wtable) {
    h = seg(h, cs); table = seg(table, cs); wtable = seg(wtable, cs);
    const uint32_t b = blockIdx.x * (BK_T / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b > plan.nbk) return;
    uint32_t* row = table + (size_t)b * plan.nchunks;
    uint32_t carry = 0, wsum = 0;
    for (uint32_t j0 = 0; j0 < plan.nchunks; j0 += 64) {
        const uint32_t j = j0 + lane;
        const uint32_t v = j < plan.nchunks ? row[j] : 0u;
        if (wtable && j < plan.nchunks) wsum += wtable[(size_t)j * (plan.nbk + 1) + b];
        const uint32_t incl = wave_incl_scan_u32(v);
        if (j < plan.nchunks) row[j] = carry + incl - v;
        carry += (uint32_t)__shfl((int)incl, 63);
    }
    if (wtable) wsum = wave_sum(wsum);
    if (lane == 0) { h->count[b] = carry; if (wtable) h->wstart[b] = wsum; }
}
// emit (fused emission): the weight sums are scanned as well (wstart), and the block settles what k_resolve_count settles on the
// unfused path: l_eff = the instance count if it fits the capacity and no bucket overflowed, else 0 (camera skipped as a
// whole); the pinned counts go to the host through their device mapping.
// ONE block of 256 threads with a handful of registers: beside the blends of the other streams (5 waves x 96 VGPRs allocated
// on every SIMD) a 1 024-thread block needs four free wave slots on all four SIMDs of ONE CU at the same moment -- the first
// fused version (68 VGPRs) waited 40 - 120 us for that, every camera batch, in the middle of the head chain.
constexpr int BKS_T = 256;
__global__ __launch_bounds__(BKS_T) void k_bk_scan(BucketHdr* __restrict__ h, size_t cs, bool emit, uint32_t capacity,
                                                  uint32_t* __restrict__ l_eff, uint32_t* __restrict__ count_host) {
    __shared__ uint32_t wsum[BKS_T / 64], wsum2[BKS_T / 64], s_worst;
    __shared__ unsigned long long s_total64;      // the instance count in 64 bits: a 32-bit total that wrapped must not pass the capacity test
    h = seg(h, cs); l_eff = seg(l_eff, cs);
    const unsigned t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t nbk = h->nbk, per = nbk / BKS_T;                   // nbk is a multiple of 1024: `per` consecutive buckets per thread
    uint32_t mine = 0, worst = 0, mine_w = 0;
    if (t == 0) { s_worst = 0u; s_total64 = 0ull; }
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = h->count[t * per + k];
        mine += c;
        worst = umax_(worst, c);
        if (emit) mine_w += h->wstart[t * per + k];
    }
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint32_t incl_w = emit ? wave_incl_scan_u32(mine_w) : 0u;
    __syncthreads();                                              // (s_total64 zeroed)
    if (emit) {
        unsigned long long m64 = 0ull;
        for (uint32_t k = 0; k < per; ++k) m64 += h->wstart[t * per + k];
        m64 = wave_sum(m64);
        if (lane == 0) atomicAdd(&s_total64, m64);
    }
    if (lane == 63) { wsum[w] = incl; wsum2[w] = incl_w; }
    __syncthreads();
    uint32_t woff = 0, total = 0, woff_w = 0, total_w = 0;
    for (int k = 0; k < BKS_T / 64; ++k) {
        const uint32_t v = wsum[k], v2 = wsum2[k];
        if (k < (int)w) { woff += v; woff_w += v2; }
        total += v; total_w += v2;
    }
    uint32_t run = woff + incl - mine, run_w = woff_w + incl_w - mine_w;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = h->count[t * per + k];
        h->start[t * per + k] = run; run += c;
        if (emit) { const uint32_t wv = h->wstart[t * per + k]; h->wstart[t * per + k] = run_w; run_w += wv; }
    }
    if (worst > h->cap) { atomicMax(&h->overflow, worst); atomicMax(&s_worst, worst); }
    if (t == 0) { h->start[nbk] = total; h->start[nbk + 1] = total + h->count[nbk]; }   // the tail bucket
    if (emit) {
        __syncthreads();
        if (t == 0) {
            const uint32_t unsorted = s_worst;                    // a bucket beyond its room: the camera is not sorted
            h->wstart[nbk] = total_w;
            const unsigned long long total64 = s_total64;
            l_eff[0] = (total64 <= (unsigned long long)capacity && !unsorted) ? total_w : 0u;
            if (count_host) {
                count_host += 4 * blockIdx.y;
                count_host[0] = total64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : total_w;      // (saturated: "does not fit", whatever the capacity)
                count_host[1] = unsorted; count_host[2] = 0u;                          // ([2] is raised by k_tile_gate later)
            }
        }
    }
}
__global__ __launch_bounds__(BK_T) void k_bk_scatter(const uint32_t* __restrict__ keys, long n, const BucketHdr* __restrict__ h,
                                                    const uint32_t* __restrict__ table, unsigned long long* __restrict__ items,
                                                    BucketPlan plan, size_t cs) {
    __shared__ uint32_t cur[BK_MAX + 1];
    __shared__ uint32_t red[2];
    keys = seg(keys, cs); h = seg(h, cs); table = seg(table, cs); items = seg(items, cs);
    const uint32_t nbk = plan.nbk;
    for (uint32_t i = threadIdx.x; i <= nbk; i += BK_T) cur[i] = h->start[i] + table[(size_t)i * plan.nchunks + blockIdx.x];
    const BucketMap m = bucket_map_from_partials(h, plan.nminmax, red);   // (its barriers also publish the cursors)
    const long base = (long)blockIdx.x * plan.kpb;
    for (uint32_t j0 = threadIdx.x; j0 < plan.kpb; j0 += BK_T * BK_UNROLL) {
        uint32_t k[BK_UNROLL];
        bool ok[BK_UNROLL];
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * BK_T;
            const long i = base + j;
            ok[u] = j < plan.kpb && i < n;
            k[u] = ok[u] ? keys[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < BK_UNROLL; ++u) {
            if (!ok[u]) continue;
            const long i = base + j0 + (uint32_t)u * BK_T;
            const uint32_t pos = atomicAdd(&cur[m.of(k[u])], 1u);
            items[pos] = ((unsigned long long)k[u] << 32) | (unsigned long long)(uint32_t)i;
        }
    }
}
// Bitonic sort of 64 * R 64-bit items held R per lane (item e = r * 64 + lane): steps with a stride >= 64 exchange
// registers of the same lane, the others one __shfl_xor per register -- no LDS, no barriers, everything unrolled.
template <int R>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&v)[R], unsigned lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            if (jj >= 64) {
                const int jr = jj >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & jr) == 0) {
                        const bool up = ((r << 6) & k) == 0;
                        const unsigned long long a = v[r], b = v[r | jr];
                        const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
                        v[r] = up ? lo : hi;
                        v[r | jr] = up ? hi : lo;
                    }
                }
            } else {
                const bool lower = (lane & (unsigned)jj) == 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool up = k < 64 ? ((lane & (unsigned)k) == 0) : (((r << 6) & k) == 0);
                    const unsigned long long x = v[r];
                    const unsigned long long y = __shfl_xor(x, jj);
                    const unsigned long long lo = x < y ? x : y, hi = x < y ? y : x;
                    v[r] = (lower == up) ? lo : hi;
                }
            }
        }
    }
}
// value of the item at input position pos: vals[pos], or the position itself (rev = 0), or rev - 1 - pos (keys fed in reversed
// index order: no index array is read or written at all)
__device__ __forceinline__ uint32_t bk_value(const uint32_t* __restrict__ vals, uint32_t pos, uint32_t rev) {
    return vals ? vals[pos] : (rev ? rev - 1u - pos : pos);
}
template <int R>
__device__ __forceinline__ void bucket_sort_in_registers(const unsigned long long* __restrict__ items, uint32_t s0, uint32_t cnt,
                                                         const uint32_t* __restrict__ vals, uint32_t* __restrict__ vals_out,
                                                         uint32_t* __restrict__ keys_out, unsigned lane, uint32_t rev) {
    unsigned long long v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t e = (uint32_t)r * 64u + lane; v[r] = e < cnt ? items[s0 + e] : ~0ull; }
    wave_bitonic_sort<R>(v, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t e = (uint32_t)r * 64u + lane;
        if (e < cnt) {
            vals_out[s0 + e] = bk_value(vals, (uint32_t)v[r], rev);
            if (keys_out) keys_out[s0 + e] = (uint32_t)(v[r] >> 32);
        }
    }
}

// one WAVE per bucket (a 64-thread block finds a place on a CU whose wave slots and LDS the blend of another camera is
// holding).  Buckets of up to 1024 items are sorted in REGISTERS (wave_bitonic_sort; the LDS version of this kernel spent
// 60-100 us per camera under load in 36 dependent LDS round trips per bucket); larger rooms keep the LDS network.
template <int CAP>
__global__ __launch_bounds__(64) void k_bk_sort(const BucketHdr* __restrict__ h, const unsigned long long* __restrict__ items,
                                                const uint32_t* __restrict__ vals, uint32_t* __restrict__ vals_out,
                                                uint32_t* __restrict__ keys_out, size_t cs, uint32_t rev) {
    __shared__ unsigned long long s_it[CAP > 1024 ? CAP : 1];
    h = seg(h, cs); items = seg(items, cs); vals = seg(vals, cs); vals_out = seg(vals_out, cs); keys_out = seg(keys_out, cs);
    const unsigned lane = threadIdx.x, nbk = h->nbk;
    // blocks nbk .. nbk + BK_TAILBLOCKS - 1 share the tail bucket (tens of thousands of off-screen keys: one wave copying
    // them alone, two dependent loads per trip, took longer than the rest of the sort)
    const unsigned b = blockIdx.x < nbk ? blockIdx.x : nbk;
    const unsigned part = blockIdx.x < nbk ? 0u : blockIdx.x - nbk, parts = blockIdx.x < nbk ? 1u : (unsigned)BK_TAILBLOCKS;
    const uint32_t s0 = h->start[b], cnt = h->start[b + 1] - s0;
    if (cnt == 0) return;
    if (b == nbk || cnt > (uint32_t)CAP) {                            // tail bucket (or an overflowing one): copy, unsorted
        for (uint32_t j = part * 64 + lane; j < cnt; j += parts * 64) {
            const unsigned long long it = items[s0 + j];
            vals_out[s0 + j] = bk_value(vals, (uint32_t)it, rev);
            if (keys_out) keys_out[s0 + j] = (uint32_t)(it >> 32);
        }
        return;
    }
    if (CAP <= 1024) {                                                // wave-uniform dispatch on the bucket's size
        if (cnt <= 64) bucket_sort_in_registers<1>(items, s0, cnt, vals, vals_out, keys_out, lane, rev);
        else if (cnt <= 128) bucket_sort_in_registers<2>(items, s0, cnt, vals, vals_out, keys_out, lane, rev);
        else if (cnt <= 256) bucket_sort_in_registers<4>(items, s0, cnt, vals, vals_out, keys_out, lane, rev);
        else if (cnt <= 512) bucket_sort_in_registers<8>(items, s0, cnt, vals, vals_out, keys_out, lane, rev);
        else bucket_sort_in_registers<16>(items, s0, cnt, vals, vals_out, keys_out, lane, rev);
        return;
    }
    uint32_t np = 2;
    while (np < cnt) np <<= 1;
    for (uint32_t j = lane; j < np; j += 64) s_it[j] = j < cnt ? items[s0 + j] : ~0ull;
    wave_sync();
    // bitonic network: pair p of step jj is (i, i + jj) with i = 2 jj (p / jj) + p % jj
    const uint32_t half = np >> 1;
    for (uint32_t k = 2; k <= np; k <<= 1)
        for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint32_t p = lane; p < half; p += 64) {
                const uint32_t i = ((p & ~(jj - 1)) << 1) | (p & (jj - 1)), l = i + jj;
                const unsigned long long x = s_it[i], y = s_it[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { s_it[i] = y; s_it[l] = x; }
            }
            wave_sync();
        }
    for (uint32_t j = lane; j < cnt; j += 64) {
        const unsigned long long it = s_it[j];
        vals_out[s0 + j] = bk_value(vals, (uint32_t)it, rev);
        if (keys_out) keys_out[s0 + j] = (uint32_t)(it >> 32);
    }
}

// Fused emission: bucket b sorted in registers as above, then -- in sorted order -- the exclusive scan of its Gaussians' weights
// on top of wstart[b] and the (tile, Gaussian) instances themselves (what k_duplicate writes from offsets[] on the unfused
// path: the same words at the same places).  The rect gathers of all R registers are requested together.
__device__ __forceinline__ void emit_instances(const BucketEmit& em, const uint8_t* __restrict__ alive, bool valid, uint32_t g,
                                               uint32_t rc, uint32_t& run) {
    const int ix0 = rc & 255, ix1 = (rc >> 8) & 255, iy0 = (rc >> 16) & 255, iy1 = rc >> 24;
    // the weight the histogram summed: the rect's area, or (child pass) the children that exist for this camera
    uint32_t wgt = 0u;
    if (valid) wgt = alive ? em.weight[g] : (uint32_t)((ix1 - ix0 + 1) * (iy1 - iy0 + 1));
    const uint32_t incl = wave_incl_scan_u32(wgt);
    uint32_t off = run + incl - wgt;
    run += (uint32_t)__shfl((int)incl, 63);
    if (!valid || wgt == 0u) return;
    for (int iy = iy0; iy <= iy1; ++iy)
        for (int ix = ix0; ix <= ix1; ++ix) {
            if (alive && !child_exists(em.tile_parent, alive, iy * em.nx + ix)) continue;
            if (em.gshift) {
                em.inst_tile[off] = ((uint32_t)(iy * em.nx + ix) << em.gshift) | g;
            } else {
                em.inst_tile[off] = (uint32_t)(iy * em.nx + ix);
                em.inst_g[off] = g;
            }
            ++off;
        }
}
template <int R>
__device__ __forceinline__ void bucket_sort_emit_in_registers(const unsigned long long* __restrict__ items, uint32_t s0, uint32_t cnt,
                                                              unsigned lane, uint32_t rev, uint32_t base, const BucketEmit& em,
                                                              const uint8_t* __restrict__ alive) {
    unsigned long long v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t e = (uint32_t)r * 64u + lane; v[r] = e < cnt ? items[s0 + e] : ~0ull; }
    wave_bitonic_sort<R>(v, lane);
    uint32_t g[R], rc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool valid = (uint32_t)r * 64u + lane < cnt;
        g[r] = valid ? bk_value(nullptr, (uint32_t)v[r], rev) : 0u;
        rc[r] = valid ? em.rect[g[r]] : 0u;
    }
    uint32_t run = base;
#pragma unroll
    for (int r = 0; r < R; ++r) emit_instances(em, alive, (uint32_t)r * 64u + lane < cnt, g[r], rc[r], run);
}
// RMAX: buckets of up to 64 * RMAX items are sorted in registers, larger ones (up to the 1 024 of the room) through an 8 KB LDS
// network.  RMAX = 16 is the whole room in registers at 86 VGPRs; RMAX = 8 halves the registers of the kernel -- beside the
// blends of the other streams (5 waves x 96 VGPRs allocated per SIMD) what a head wave needs decides when it can start.
template <int RMAX>
__global__ __launch_bounds__(64) void k_bk_sort_emit(const BucketHdr* __restrict__ h, const unsigned long long* __restrict__ items,
                                                     size_t cs, uint32_t rev, BucketEmit em) {
    __shared__ unsigned long long s_it[RMAX < 16 ? 1024 : 1];
    h = seg(h, cs); items = seg(items, cs);
    em.weight = seg(em.weight, cs); em.rect = seg(em.rect, cs); em.inst_tile = seg(em.inst_tile, cs); em.inst_g = seg(em.inst_g, cs);
    em.l_eff = seg(em.l_eff, cs);
    if (*em.l_eff == 0u) return;                  // nothing to emit, more than fits, or a bucket beyond its room: camera skipped
    const uint8_t* alive = nullptr;
    if (em.tile_parent && em.jobs) {
        const G2pcCameraJob* jb = em.jobs + blockIdx.y;
        alive = (const uint8_t*)(((unsigned long long)jb->alive_hi << 32) | jb->alive_lo);
    }
    const unsigned lane = threadIdx.x, b = blockIdx.x;         // (the tail bucket -- keys 0xFFFFFFFF, weight 0 -- emits nothing)
    const uint32_t s0 = h->start[b], cnt = h->start[b + 1] - s0, base = h->wstart[b];
    if (cnt == 0) return;
    if (cnt <= 64) bucket_sort_emit_in_registers<1>(items, s0, cnt, lane, rev, base, em, alive);
    else if (cnt <= 128) bucket_sort_emit_in_registers<2>(items, s0, cnt, lane, rev, base, em, alive);
    else if (cnt <= 256) bucket_sort_emit_in_registers<4>(items, s0, cnt, lane, rev, base, em, alive);
    else if (cnt <= 512) bucket_sort_emit_in_registers<8>(items, s0, cnt, lane, rev, base, em, alive);
    else if (RMAX >= 16) bucket_sort_emit_in_registers<(RMAX >= 16 ? 16 : 1)>(items, s0, cnt, lane, rev, base, em, alive);
    else {
        // (rare: a bucket of more than twice the mean) bitonic network in LDS: pair p of step jj is (i, i + jj), i = 2 jj (p / jj) + p % jj
        uint32_t np = 2;
        while (np < cnt) np <<= 1;
        for (uint32_t j = lane; j < np; j += 64) s_it[j] = j < cnt ? items[s0 + j] : ~0ull;
        wave_sync();
        const uint32_t half = np >> 1;
        for (uint32_t k = 2; k <= np; k <<= 1)
            for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
                for (uint32_t p = lane; p < half; p += 64) {
                    const uint32_t i = ((p & ~(jj - 1)) << 1) | (p & (jj - 1)), l = i + jj;
                    const unsigned long long x = s_it[i], y = s_it[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { s_it[i] = y; s_it[l] = x; }
                }
                wave_sync();
            }
        uint32_t run = base;
        for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
            const bool valid = e0 + lane < cnt;
            const uint32_t g = valid ? bk_value(nullptr, (uint32_t)s_it[e0 + lane], rev) : 0u;
            const uint32_t rc = valid ? em.rect[g] : 0u;
            emit_instances(em, alive, valid, g, rc, run);
        }
    }
}

// the per-bucket sort is one wave per bucket: it pays while buckets stay small (the 8 KB footprint)
bool bucket_sort_pays(long n) { return n > 0 && bucket_plan(n).cap == (uint32_t)BK_CAP_SMALL; }
bool bucket_emit_supported(long n) { return bucket_sort_pays(n) && n < (1l << 31); }

size_t bucket_sort_workspace(long n) {
    const BucketPlan p = bucket_plan(n > 0 ? n : 1);
    return align_up(sizeof(BucketHdr)) + align_up((size_t)(n > 0 ? n : 1) * 8) + 2 * align_up((size_t)(p.nbk + 1) * p.nchunks * 4) + 1024;
}

// vals_out[p] = vals[r_p] (and keys_out[p] = keys[r_p] when given) for the positions r sorted by (keys[r], r) ascending,
// keys 0xFFFFFFFF last.  *overflow_flag (device u32, optional) receives the size of the largest bucket when one exceeds
// the room of a bucket (BucketPlan::cap) -- the output is then a permutation in bucket order only and the caller must sort again with sort_pairs_u32.
int bucket_sort_u32(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, uint32_t* keys_out, long n, void* ws,
                    size_t ws_bytes, uint32_t** overflow_flag, hipStream_t s, Batch b, bool minmax_done, bool reversed,
                    const BucketEmit* emit) {
    if (n <= 0) return G2PC_OK;
    const BucketPlan plan = bucket_plan(n);
    const unsigned by = (unsigned)b.n;
    const uint32_t rev = (!vals && reversed) ? (uint32_t)n : 0u;
    Arena ar(ws, ws_bytes);
    BucketHdr* h = ar.get<BucketHdr>(1);
    unsigned long long* items = ar.get<unsigned long long>((size_t)n);
    uint32_t* table = ar.get<uint32_t>((size_t)(plan.nbk + 1) * plan.nchunks);
    uint32_t* wtable = ar.get<uint32_t>((size_t)(plan.nbk + 1) * plan.nchunks);
    if (!ar.ok()) { set_error("bucket_sort", "workspace too small"); return G2PC_ERR_WORKSPACE; }
    if (emit && (vals || plan.cap != (uint32_t)BK_CAP_SMALL || !emit->rect || !emit->inst_tile || !emit->l_eff ||
                 (emit->tile_parent && !emit->weight))) {
        set_error("bucket_sort", "fused emission needs position values (vals == NULL), register-sized buckets and its arrays");
        return G2PC_ERR_ARG;
    }
    if (!minmax_done) hipLaunchKernelGGL(k_bk_minmax, dim3(plan.nminmax, by), dim3(BK_T), 0, s, keys, n, h, plan, b.cs);
    if (emit) {
        if (plan.nbk <= 4096)
            hipLaunchKernelGGL(k_bk_hist_w<4096>, dim3(plan.nchunks, by), dim3(BK_T), 0, s, keys, n, (const BucketHdr*)h, table, wtable, emit->weight, emit->rect, plan, b.cs, rev != 0u);
        else
            hipLaunchKernelGGL(k_bk_hist_w<BK_MAX>, dim3(plan.nchunks, by), dim3(BK_T), 0, s, keys, n, (const BucketHdr*)h, table, wtable, emit->weight, emit->rect, plan, b.cs, rev != 0u);
    } else {
        hipLaunchKernelGGL(k_bk_hist, dim3(plan.nchunks, by), dim3(BK_T), 0, s, keys, n, (const BucketHdr*)h, table, plan, b.cs);
    }
    hipLaunchKernelGGL(k_bk_colscan, dim3(cdiv(plan.nbk + 1, BK_T / 64), by), dim3(BK_T), 0, s, h, table, plan, b.cs,
                       emit ? (const uint32_t*)wtable : (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_bk_scan, dim3(1, by), dim3(BKS_T), 0, s, h, b.cs, emit != nullptr, emit ? emit->capacity : 0u,
                       emit ? emit->l_eff : (uint32_t*)nullptr, emit ? emit->count_host : (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_bk_scatter, dim3(plan.nchunks, by), dim3(BK_T), 0, s, keys, n, (const BucketHdr*)h, (const uint32_t*)table, items, plan, b.cs);
    if (emit) {
        hipLaunchKernelGGL(k_bk_sort_emit<G2PC_BK_EMIT_RMAX>, dim3(plan.nbk, by), dim3(64), 0, s, (const BucketHdr*)h, (const unsigned long long*)items, b.cs, rev, *emit);
        if (overflow_flag) *overflow_flag = &h->overflow;
        return check_launch("bucket_sort");
    }
    if (plan.cap == (uint32_t)BK_CAP_SMALL)
        hipLaunchKernelGGL(k_bk_sort<BK_CAP_SMALL>, dim3(plan.nbk + BK_TAILBLOCKS, by), dim3(64), 0, s, (const BucketHdr*)h,
                           (const unsigned long long*)items, vals, vals_out, keys_out, b.cs, rev);
    else
        hipLaunchKernelGGL(k_bk_sort<BK_CAP_LARGE>, dim3(plan.nbk + BK_TAILBLOCKS, by), dim3(64), 0, s, (const BucketHdr*)h,
                           (const unsigned long long*)items, vals, vals_out, keys_out, b.cs, rev);
    if (overflow_flag) *overflow_flag = &h->overflow;
    return check_launch("bucket_sort");
}

// self-test of the DPP reductions against the shuffle based ones (tests only)
__global__ __launch_bounds__(64) void k_selftest_wave_reduce(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
    uint32_t a = wave_max_u32_dpp(v), b = wave_min_u32_dpp(v), c = wave_max_u32(v), d = wave_min_u32(v);
    if (threadIdx.x == 0) { out[4 * blockIdx.x] = a; out[4 * blockIdx.x + 1] = b; out[4 * blockIdx.x + 2] = c; out[4 * blockIdx.x + 3] = d; }
}

}  // namespace g2pc

// ---- C ABI ----------------------------------------------------------------------------------------
extern "C" {
const char* g2pc_last_error(void) { return g2pc::g_err.c_str(); }
int g2pc_abi_version(void) { return G2PC_ABI_VERSION; }

#ifdef G2PC_EXPERIMENTS
#pragma GCC visibility push(default)
/* EXPERIMENTS ONLY: digit width (8 or 11 bits) for sorts of more than 8 bits -- 10: one 10-bit pass for fields of 9 - 10 bits --;
 * inputs up to small_input_keys use 4 keys per thread */
int g2pc_set_sort_tuning(int wide_digit_bits, int64_t small_input_keys) {
    if (wide_digit_bits != 8 && wide_digit_bits != 10 && wide_digit_bits != 11) return G2PC_ERR_ARG;
    g2pc::g_radix_wide_bits = wide_digit_bits;
    g2pc::g_radix_small_n = small_input_keys;
    return G2PC_OK;
}
#pragma GCC visibility pop
#endif
#define G2PC_HIP_CALL(expr, where)                                                  \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) { g2pc::set_error(where, hipGetErrorString(e_)); return G2PC_ERR_LAUNCH; } \
    } while (0)

int g2pc_graph_capture_begin(void* stream) {
    G2PC_REQUIRE(stream, G2PC_ERR_ARG, "capture needs a non-default stream");
    G2PC_HIP_CALL(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "g2pc_graph_capture_begin");
    return G2PC_OK;
}
int g2pc_graph_capture_end(void* stream, void** graph_exec) {
    G2PC_REQUIRE(stream && graph_exec, G2PC_ERR_ARG, "bad arguments");
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    G2PC_HIP_CALL(hipStreamEndCapture((hipStream_t)stream, &graph), "g2pc_graph_capture_end");
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    G2PC_HIP_CALL(e, "g2pc_graph_capture_end (instantiate)");
    *graph_exec = (void*)exec;
    return G2PC_OK;
}
int g2pc_graph_launch(void* graph_exec, void* stream) {
    G2PC_REQUIRE(graph_exec, G2PC_ERR_ARG, "null graph");
    G2PC_HIP_CALL(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "g2pc_graph_launch");
    return G2PC_OK;
}
int g2pc_graph_destroy(void* graph_exec) {
    if (graph_exec) G2PC_HIP_CALL(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "g2pc_graph_destroy");
    return G2PC_OK;
}
int g2pc_selftest_wave_reduce(const uint32_t* in, uint32_t* out, int64_t waves, void* stream) {
    hipLaunchKernelGGL(g2pc::k_selftest_wave_reduce, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, in, out);
    return g2pc::check_launch("selftest");
}
size_t g2pc_scan_workspace(int64_t n) { return g2pc::scan_workspace(n); }
int g2pc_scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, void* ws, size_t ws_bytes, void* stream) {
    return g2pc::scan_exclusive_u32(in, out, n, ws, ws_bytes, (hipStream_t)stream, nullptr);
}
size_t g2pc_sort_workspace(int64_t n) { return g2pc::sort_workspace(n); }
size_t g2pc_bucket_sort_workspace(int64_t n) { return g2pc::bucket_sort_workspace(n); }
/* Stable ascending sort of (key, value) u32 pairs for keys that are bit patterns of positive floats spread over their
 * range (one camera's depths): range-normalised bucket pass + in-LDS bitonic sort per bucket, five launches.  Keys
 * 0xFFFFFFFF end up last.  *overflow (device u32, zeroed here) != 0 afterwards means a bucket was too large and the
 * result is NOT sorted: sort again with g2pc_sort_pairs_u32. */
int g2pc_bucket_sort_u32(const uint32_t* keys, const uint32_t* vals, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
                         uint32_t* overflow, void* ws, size_t ws_bytes, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(keys && vals_out && ws, G2PC_ERR_ARG, "null pointer");      // vals == NULL: the values are the input positions
    uint32_t* flag = nullptr;
    int rc = bucket_sort_u32(keys, vals, vals_out, keys_out, (long)n, ws, ws_bytes, &flag, (hipStream_t)stream);
    if (rc) return rc;
    if (overflow && hipMemcpyAsync(overflow, flag, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        set_error("bucket_sort", "copy failed");
        return G2PC_ERR_LAUNCH;
    }
    return G2PC_OK;
}
int g2pc_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                        uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                        size_t ws_bytes, void* stream) {
    return g2pc::sort_pairs_u32(keys_in, vals_in, keys_out, vals_out, keys_tmp, vals_tmp, n, bit_lo, bit_hi, ws,
                                ws_bytes, (hipStream_t)stream);
}
}
