// TEST INFRASTRUCTURE ONLY -- a tiny single-threaded (fiber based) host emulation of the HIP
// device model, used to exercise the *logic* of csrc/*.hip on the GPU-less authoring container.
//
// The product sources contain no emulation switches: the emulation build simply puts this
// directory first on the include path so that <hip/hip_runtime.h> resolves here, and compiles
// the same .hip files with g++ (see tests/hipemu/build_emu.sh).  Nothing in the product package
// ever loads the resulting libg2pc_emu.so; only tests/test_emu_*.py do.
//
// Model: one OS thread; every GPU thread of a block is a ucontext fiber; blocks run one after
// another (so `__shared__` can be a plain function-local static).  __syncthreads() and the wave
// intrinsics yield until the whole block / wave has arrived.  Wave size is 64.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define G2PC_PIN(x) ((void)0)
#define G2PC_PIN_S(x) ((void)0)
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDeviceToHost 2
#define hipMemcpyHostToDevice 1
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
// ---- stream capture / graphs: while capturing, queued operations are recorded as closures instead of executed ----
typedef void* hipEvent_t;
namespace hipemu {
struct Graph { std::vector<std::function<void()>> ops; };
inline Graph*& capturing() { static Graph* g = nullptr; return g; }
template <typename F> inline void submit(F f) { if (capturing()) capturing()->ops.push_back(f); else f(); }
}  // namespace hipemu
typedef hipemu::Graph* hipGraph_t;
typedef hipemu::Graph* hipGraphExec_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
#define hipStreamCaptureModeThreadLocal 1
#define hipEventRecordDefault 0u
#define hipEventRecordExternal 1u
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { hipemu::submit([=]() { memset(p, v, n); }); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { hipemu::submit([=]() { memcpy(d, s, n); }); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = nullptr; return 0; }   // (no CUs to mask here)
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { hipemu::capturing() = new hipemu::Graph(); return 0; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = hipemu::capturing(); hipemu::capturing() = nullptr; return 0; }
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipemu::capturing() ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return 0; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new hipemu::Graph(*g); return 0; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return 0; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; return 0; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t) { for (auto& op : g->ops) op(); return 0; }

namespace hipemu {
// Fiber switch.  glibc's swapcontext saves and restores the signal mask with two system calls per switch, and a wave
// collective of 64 lanes costs ~250 switches: on x86-64 a six-register hand-written switch makes the whole CPU test suite
// several times faster.  Other hosts keep ucontext.
#if defined(__x86_64__)
struct Ctx { void* sp = nullptr; };
static __attribute__((naked, noinline)) void ctx_switch(Ctx* /*from: rdi*/, Ctx* /*to: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t pushq %rbx\n\t pushq %r12\n\t pushq %r13\n\t pushq %r14\n\t pushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq (%rsi), %rsp\n\t"
        "popq %r15\n\t popq %r14\n\t popq %r13\n\t popq %r12\n\t popq %rbx\n\t popq %rbp\n\t"
        "ret\n\t");
}
inline void ctx_make(Ctx& c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of `entry` (it never returns): rsp % 16 == 8 at its first instruction
    *--sp = (void*)entry;            // popped by ctx_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = nullptr;     // rbp rbx r12 r13 r14 r15
    c.sp = (void*)sp;
}
#else
struct Ctx { ucontext_t uc; };
inline void ctx_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
inline void ctx_make(Ctx& c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = size;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif
struct Fiber {
    Ctx ctx;
    std::vector<char> stack;
    bool done = false;
};
struct State {
    dim3 grid, block;
    unsigned nthreads = 0;
    uint3_emu tid{}, bid{};
    unsigned cur = 0;          // linear thread id of the running fiber
    std::vector<Fiber> fibers;
    Ctx sched;
    // block barrier
    unsigned blk_arrived = 0, blk_alive = 0;
    unsigned long blk_gen = 0;
    // wave barriers
    unsigned wave_arrived[16] = {0}, wave_alive[16] = {0};
    unsigned long wave_gen[16] = {0};
    uint64_t slot[16][64];
    std::function<void()> body;
};
inline State& S() { static State s; return s; }
inline void yield() { State& s = S(); ctx_switch(&s.fibers[s.cur].ctx, &s.sched); }
inline void block_barrier() {
    State& s = S();
    unsigned long g = s.blk_gen;
    if (++s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; ++s.blk_gen; return; }
    while (s.blk_gen == g) yield();
}
inline void wave_barrier() {
    State& s = S();
    unsigned w = s.cur >> 6;
    unsigned long g = s.wave_gen[w];
    if (++s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; ++s.wave_gen[w]; return; }
    while (s.wave_gen[w] == g) yield();
}
inline void fiber_exit() {
    State& s = S();
    unsigned w = s.cur >> 6;
    s.fibers[s.cur].done = true;
    --s.blk_alive;
    --s.wave_alive[w];
    if (s.blk_alive && s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; ++s.blk_gen; }
    if (s.wave_alive[w] && s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; ++s.wave_gen[w]; }
}
inline void trampoline() {
    S().body();
    fiber_exit();
    State& s = S();
    ctx_switch(&s.fibers[s.cur].ctx, &s.sched);
}
inline void set_ids(unsigned t) {
    State& s = S();
    s.cur = t;
    s.tid.x = t % s.block.x;
    s.tid.y = (t / s.block.x) % s.block.y;
    s.tid.z = t / (s.block.x * s.block.y);
}
inline void run_block(unsigned bx, unsigned by, unsigned bz) {
    State& s = S();
    s.bid = {bx, by, bz};
    s.blk_arrived = 0; s.blk_alive = s.nthreads;
    for (unsigned w = 0; w < 16; ++w) {
        s.wave_arrived[w] = 0;
        unsigned lo = w * 64, hi = std::min(lo + 64, s.nthreads);
        s.wave_alive[w] = hi > lo ? hi - lo : 0;
    }
    for (unsigned t = 0; t < s.nthreads; ++t) {
        Fiber& f = s.fibers[t];
        f.done = false;
        ctx_make(f.ctx, f.stack.data(), f.stack.size(), trampoline);
    }
    unsigned remaining = s.nthreads;
    while (remaining) {
        remaining = 0;
        for (unsigned t = 0; t < s.nthreads; ++t) {
            if (s.fibers[t].done) continue;
            set_ids(t);
            ctx_switch(&s.sched, &s.fibers[t].ctx);
            if (!s.fibers[t].done) ++remaining;
        }
    }
}
template <typename K, typename... A>
inline void launch_now(K kernel, dim3 grid, dim3 block, A... args);
template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, A... args) {
    submit([=]() { launch_now(kernel, grid, block, args...); });
}
template <typename K, typename... A>
inline void launch_now(K kernel, dim3 grid, dim3 block, A... args) {
    State& s = S();
    s.grid = grid; s.block = block;
    s.nthreads = block.x * block.y * block.z;
    if (s.nthreads > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if (s.fibers.size() < s.nthreads) {
        size_t old = s.fibers.size();
        s.fibers.resize(s.nthreads);
        for (size_t i = old; i < s.nthreads; ++i) s.fibers[i].stack.resize(256 * 1024);
    }
    s.body = [=]() { kernel(args...); };
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) run_block(bx, by, bz);
}
// all alive lanes of the calling wave publish v; the returned table is valid until the next exchange
inline uint64_t alive_mask() {
    State& s = S();
    unsigned w = s.cur >> 6;
    uint64_t m = 0;
    for (unsigned l = 0; l < 64; ++l) {
        unsigned t = w * 64 + l;
        if (t < s.nthreads && !s.fibers[t].done) m |= (1ull << l);
    }
    return m;
}
inline const uint64_t* exchange(uint64_t v, uint64_t* active_mask = nullptr) {
    State& s = S();
    unsigned w = s.cur >> 6, l = s.cur & 63;
    wave_barrier();              // previous readers are done
    s.slot[w][l] = v;
    // the participants, taken while all of them are still inside the collective: a lane released early from the second
    // barrier may run to the end of the kernel before the others look (readfirstlane then picked the wrong lane)
    if (active_mask) *active_mask = alive_mask();
    wave_barrier();
    return s.slot[w];
}
inline int block_reduce(int v, int op) {   // op 0: or, 1: sum
    static int acc;
    block_barrier();
    acc = 0;
    block_barrier();
    if (op == 0) acc |= (v != 0); else acc += v;
    block_barrier();
    int r = acc;
    block_barrier();
    return r;
}
}  // namespace hipemu

#define threadIdx (hipemu::S().tid)
#define blockIdx (hipemu::S().bid)
#define blockDim (hipemu::S().block)
#define gridDim (hipemu::S().grid)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave intrinsics (all active lanes of the wave must call them together) ----
template <typename T> static inline uint64_t emu_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> static inline T emu_unbits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    unsigned l = hipemu::S().cur & 63;
    const uint64_t* t = hipemu::exchange(emu_bits(v));
    int base = (l / width) * width;
    return emu_unbits<T>(t[base + (src % width + width) % width]);
}
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) {
    unsigned l = hipemu::S().cur & 63;
    const uint64_t* t = hipemu::exchange(emu_bits(v));
    (void)width;
    return emu_unbits<T>(t[(l ^ m) & 63]);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    unsigned l = hipemu::S().cur & 63;
    const uint64_t* t = hipemu::exchange(emu_bits(v));
    (void)width;
    return l >= d ? emu_unbits<T>(t[l - d]) : v;
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    unsigned l = hipemu::S().cur & 63;
    const uint64_t* t = hipemu::exchange(emu_bits(v));
    (void)width;
    return l + d < 64 ? emu_unbits<T>(t[l + d]) : v;
}
static inline unsigned long long __ballot(int pred) {
    uint64_t act = 0;
    const uint64_t* t = hipemu::exchange(pred ? 1 : 0, &act);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (((act >> i) & 1) && t[i]) m |= (1ull << i);
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    uint64_t act = 0;
    const uint64_t* t = hipemu::exchange(pred ? 1 : 0, &act);
    for (int i = 0; i < 64; ++i) if (((act >> i) & 1) && !t[i]) return 0;
    return 1;
}
static inline int __syncthreads_or(int pred) { return hipemu::block_reduce(pred != 0, 0); }
static inline int __syncthreads_and(int pred) { return !hipemu::block_reduce(pred == 0, 0); }
static inline int __syncthreads_count(int pred) { return hipemu::block_reduce(pred != 0, 1); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline unsigned __lane_id() { return hipemu::S().cur & 63; }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
    unsigned l = hipemu::S().cur & 63;
    unsigned m = l >= 32 ? mask : (mask & ((1u << l) - 1));
    return add + __builtin_popcount(m);
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
    unsigned l = hipemu::S().cur & 63;
    unsigned m = l <= 32 ? 0u : (mask & ((1u << (l - 32)) - 1));
    return add + __builtin_popcount(m);
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
    uint64_t act = 0;
    const uint64_t* t = hipemu::exchange((uint64_t)(uint32_t)v, &act);
    return (int)(uint32_t)t[__builtin_ctzll(act)];
}

// ---- atomics (single threaded emulation: plain RMW) ----
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// scoped atomics (single threaded emulation: plain accesses)
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <typename T, typename V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = (T)v; }

// ---- math ----
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline void sincospif(float x, float* s, float* c) {
    double a = M_PI * (double)x; *s = (float)sin(a); *c = (float)cos(a);
}
static inline float cospif(float x) { return (float)cos(M_PI * (double)x); }
static inline float sinpif(float x) { return (float)sin(M_PI * (double)x); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
using std::max;
using std::min;

// amdgcn builtins used by g2pc_device.inl
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }

// DPP / readlane emulation (only the controls g2pc uses: quad_perm, row_shr:n, row_bcast:15, row_bcast:31)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    unsigned l = hipemu::S().cur & 63;
    uint64_t act = 0;
    const uint64_t* t = hipemu::exchange((uint64_t)(uint32_t)src, &act);
    unsigned row = l >> 4, bank = (l & 15) >> 2;
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    int j = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) j = (int)((l & ~3u) | ((unsigned)(ctrl >> (2 * (l & 3))) & 3u));      // quad_perm
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { int n = ctrl - 0x110; if ((int)(l & 15) >= n) j = (int)l - n; }
    else if (ctrl == 0x142) { if (row >= 1) j = (int)(16 * (row - 1) + 15); }
    else if (ctrl == 0x143) { if (row >= 2) j = 31; }
    else { fprintf(stderr, "hipemu: unsupported dpp ctrl 0x%x\n", ctrl); abort(); }
    if (j < 0 || !((act >> j) & 1)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)t[j];
}
static inline int __builtin_amdgcn_readlane(int v, int lane) {
    const uint64_t* t = hipemu::exchange((uint64_t)(uint32_t)v);
    return (int)(uint32_t)t[lane & 63];
}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }
static inline unsigned long long wall_clock64() { return 0ull; }
