// Internal helpers shared by the g2pc HIP translation units (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../../include/g2pc.h"

namespace g2pc {

constexpr int kWave = 64;

// ---- error plumbing (thread-local message, negative int status) -------------------------------
void set_error(const char* where, const char* what);
int check_launch(const char* where);

#define G2PC_REQUIRE(cond, code, msg)                      \
    do {                                                   \
        if (!(cond)) {                                     \
            ::g2pc::set_error(__func__, msg);              \
            return code;                                   \
        }                                                  \
    } while (0)

static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
struct Arena {
    char* base;
    size_t off, cap;
    Arena(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
    template <typename T> T* get(size_t n) {
        size_t o = align_up(off);
        off = o + n * sizeof(T);
        return (T*)(base + o);
    }
    bool ok() const { return off <= cap; }
};

// ---- camera batching -----------------------------------------------------------------------------
// A batched launch runs the same kernel for `batch` cameras at once: grid.y = batch, and everything a camera owns lives in
// ONE contiguous arena, the arenas `cs` bytes apart -- so every per-camera pointer of a kernel moves by the same
// blockIdx.y * cs bytes.  batch = 1 (grid.y = 1) is the single-camera launch, whatever cs.
template <typename T> __device__ __forceinline__ T* seg(T* p, size_t cs) {
    return p ? (T*)((char*)p + (size_t)blockIdx.y * cs) : p;
}
// The blends put the camera in blockIdx.x instead (workgroups are dispatched x fastest): the chunks of all cameras of the
// batch start in the layout's long-walks-first order TOGETHER, so the launch has one tail, not one per camera.
template <typename T> __device__ __forceinline__ T* seg_at(T* p, size_t cs, unsigned cam) {
    return p ? (T*)((char*)p + (size_t)cam * cs) : p;
}
struct Batch { int n = 1; size_t cs = 0; };          // host side: number of cameras, arena stride in bytes

// ---- wave-level helpers -------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned l = lane_id();
    return l == 0 ? 0ull : (~0ull >> (64 - l));
}
// compiler + LDS ordering point between the lanes of ONE wave (no s_barrier)
__device__ __forceinline__ void wave_sync();

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { unsigned o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { unsigned o = __shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
    unsigned l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned o = __shfl_up(v, d); if (l >= (unsigned)d) v += o; }
    return v;
}

// ---- device-side primitives implemented in prims.hip -------------------------------------------
// exclusive scan of n u32 values; out has n+1 entries (out[n] = total).  in may alias out.
size_t scan_workspace(long n);
// gather != nullptr: scans in[gather[i]] (in and out must then be distinct)
int scan_exclusive_u32(const uint32_t* in, uint32_t* out, long n, void* ws, size_t ws_bytes, hipStream_t s,
                       const uint32_t* gather = nullptr, Batch b = Batch());
// `rows` independent scans of n values each (row r: in + r*n -> out + r*(n+1)) in two launches; n <= 2M per row
size_t scan_rows_workspace(long n, int rows);
int scan_exclusive_rows_u32(const uint32_t* in, uint32_t* out, long n, int rows, void* ws, size_t ws_bytes, hipStream_t s);
// stable LSD radix sort of (key,value) u32 pairs on bits [bit_lo, bit_hi).  Result ends in
// keys_out/vals_out (ping-pong buffers keys_tmp/vals_tmp are scratch of n entries each).
size_t sort_workspace(long n);
int sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                   uint32_t* keys_tmp, uint32_t* vals_tmp, long n, int bit_lo, int bit_hi, void* ws,
                   size_t ws_bytes, hipStream_t s, const uint32_t* n_dev = nullptr, Batch b = Batch());

// bucket sort of (key, input position) pairs for range-spread float-bit keys (see prims.hip); *overflow_flag points into
// the workspace afterwards (device u32: non-zero = a bucket overflowed, sort again with sort_pairs_u32)
constexpr int BK_MAX = 8192, BK_PARTIALS = 256;
struct BucketHdr {
    uint32_t overflow, nbk, nchunks, cap;
    uint32_t partial[2 * BK_PARTIALS];              // per minmax slot: max(~key), max(key)
    uint32_t count[BK_MAX + 2], start[BK_MAX + 2];
    uint32_t wstart[BK_MAX + 2];                    // fused emission (BucketEmit): exclusive scan of the buckets' weight sums
};
struct BucketPlan { uint32_t nbk, kpb, nchunks, nminmax, cap; };
// Fused emission (the rasteriser's captured camera path): every key carries a weight -- the tiles its Gaussian touches -- and
// the sort's last kernel, one wave per bucket, turns its sorted bucket straight into (tile, Gaussian) instances at
// [exclusive scan of the weights in sorted order).  The weight sums per bucket ride in the histogram pass (one 64-bit LDS
// add per key: count << 32 | weight), their scan and the capacity test in k_bk_scan: no separate scan over the n weights,
// no duplication kernel, no count kernel (four launches and ~25 MB of traffic per camera less).  The values are the input
// positions (vals == NULL; n - 1 - position when the keys are fed in reversed index order); weights / rects are indexed by the value.
struct BucketEmit {
    const uint32_t* weight;       // [n] instances per Gaussian (0 for keys 0xFFFFFFFF); NULL: the area of the rect (no child pass)
    const uint32_t* rect;         // [n] ix0 | ix1 << 8 | iy0 << 16 | iy1 << 24 tile-interval ranges
    uint32_t* inst_tile;          // out: tile << gshift | Gaussian (gshift > 0), else the tile id ...
    uint32_t* inst_g;             // ... with the Gaussian here
    int gshift, nx;
    uint32_t capacity;            // room of the instance arrays
    uint32_t* l_eff;              // out (device): the instance count, or 0 when it exceeds the capacity / a bucket overflowed
    uint32_t* count_host;         // out (pinned, optional): [camera][instances, unsorted, 0, -]
    const int32_t* tile_parent;   // child pass of a camera: only the children of split nodes take instances ...
    const G2pcCameraJob* jobs;    // ... per the camera's `alive` bytes (weight counted them the same way)
};
BucketPlan bucket_plan(long n);
size_t bucket_sort_workspace(long n);
bool bucket_sort_pays(long n);          // measured on MI355X: 90 vs 101 us (radix) at 1 M keys, 533 vs 278 us at 5 M
// the header bucket_sort_u32 keeps at the start of its workspace: a producer of the keys may fill partial[] itself
// (zeroed header + atomicMax of (~key, key) into slot block % plan.nminmax, see bucket_hdr_init / bucket_minmax_note) and
// pass minmax_done = true, which saves the pass over the keys that finds their range
inline BucketHdr* bucket_sort_header(void* ws) { return (BucketHdr*)ws; }
__device__ __forceinline__ void bucket_hdr_init(BucketHdr* h, const BucketPlan& plan, unsigned tid, unsigned nthreads) {
    for (unsigned i = tid; i < 2u * BK_PARTIALS; i += nthreads) h->partial[i] = 0u;
    if (tid == 0) { h->overflow = 0; h->nbk = plan.nbk; h->nchunks = plan.nchunks; h->cap = plan.cap; }
}
// vals == nullptr: the values are the input positions r themselves, or n - 1 - r when `reversed` (keys fed in reversed index order)
int bucket_sort_u32(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, uint32_t* keys_out, long n, void* ws,
                    size_t ws_bytes, uint32_t** overflow_flag, hipStream_t s, Batch b = Batch(), bool minmax_done = false,
                    bool reversed = false, const BucketEmit* emit = nullptr);
bool bucket_emit_supported(long n);     // the fused emission exists for the register-sorted buckets (cap 1024) only

// Philox4x32-10 keyed standard normals (see oracle/np_philox.py for the definition)
struct Normal3 { float x, y, z; };
__device__ __forceinline__ Normal3 keyed_normal3(unsigned seed_lo, unsigned seed_hi, unsigned gid_lo,
                                                 unsigned gid_hi, unsigned attempt, unsigned k);

// closed-form eigenvalues of a symmetric 3x3 (fp64); ascending order e[0] <= e[1] <= e[2]
__device__ __forceinline__ void sym3_eigvals(double a00, double a01, double a02, double a11, double a12,
                                             double a22, double e[3]);

}  // namespace g2pc

#include "g2pc_device.inl"
