// Device inline implementations for g2pc_internal.h
#pragma once

// Pins a value in a VGPR at this program point: stops LLVM from sinking independent work (e.g. exp chains) past
// later branches, i.e. keeps hand-scheduled instruction-level parallelism.
#ifndef G2PC_PIN
#define G2PC_PIN(x) asm volatile("" : "+v"(x))
#endif

// Constant address space (LLVM addrspace 4): memory that no thread writes while the kernel runs.  A load through such a
// pointer with a wave-uniform address is selected as a SCALAR load (s_load_dword*: scalar cache, result in SGPRs) instead of
// a vector load that fetches the same bytes for 64 lanes.  (The CPU build of these sources ignores it.)
#if defined(__clang__)
#define G2PC_CONSTANT __attribute__((address_space(4)))
typedef float g2pc_f4v __attribute__((ext_vector_type(4)));     // plain vector types: loadable through any address space
typedef float g2pc_f2v __attribute__((ext_vector_type(2)));
#else
#define G2PC_CONSTANT
typedef float g2pc_f4v __attribute__((vector_size(16)));
typedef float g2pc_f2v __attribute__((vector_size(8)));
#endif

namespace g2pc {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// diagnostics: where a wave runs.  HW_ID (hwreg 4): wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13];
// XCC_ID (hwreg 20): the XCD.  s_getreg_b32 immediate = (size - 1) << 11 | offset << 6 | register.
__device__ __forceinline__ uint32_t g2pc_hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ uint32_t g2pc_xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }

// ---- wave64 reductions on the DPP crossbar (no LDS traffic, unlike ds_bpermute based __shfl) ----------------
// row_shr:1,2,4,8 give every lane 15 of a 16-lane row the row result; row_bcast:15 / row_bcast:31 fold the four
// rows into lane 63; v_readlane broadcasts it.  `identity` fills lanes that have no source.
#define G2PC_DPP_STEP(OP, CTRL, ROWMASK)                                                              \
    {                                                                                                 \
        unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROWMASK, 0xF, false); \
        v = OP(v, o);                                                                                 \
    }
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
    const unsigned identity = 0u;
    G2PC_DPP_STEP(umax_, 0x111, 0xF)   // row_shr:1
    G2PC_DPP_STEP(umax_, 0x112, 0xF)   // row_shr:2
    G2PC_DPP_STEP(umax_, 0x114, 0xF)   // row_shr:4
    G2PC_DPP_STEP(umax_, 0x118, 0xF)   // row_shr:8
    G2PC_DPP_STEP(umax_, 0x142, 0xA)   // row_bcast:15 -> rows 1,3
    G2PC_DPP_STEP(umax_, 0x143, 0xC)   // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v) {
    const unsigned identity = 0xFFFFFFFFu;
    G2PC_DPP_STEP(umin_, 0x111, 0xF)
    G2PC_DPP_STEP(umin_, 0x112, 0xF)
    G2PC_DPP_STEP(umin_, 0x114, 0xF)
    G2PC_DPP_STEP(umin_, 0x118, 0xF)
    G2PC_DPP_STEP(umin_, 0x142, 0xA)
    G2PC_DPP_STEP(umin_, 0x143, 0xC)
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#undef G2PC_DPP_STEP

// ---- packed f32 pairs: v_pk_add/mul/fma_f32 retire two IEEE f32 operations per lane per issue slot (the 157 TF
// vector peak of the part is quoted on them); element-wise results are bit-identical to the scalar instructions.
#if defined(__clang__)
typedef float pk2 __attribute__((ext_vector_type(2)));
#else
typedef float pk2 __attribute__((vector_size(8)));      // tests/hipemu builds this file with g++
#endif
__device__ __forceinline__ pk2 pk_make(float a, float b) { pk2 v = {a, b}; return v; }
__device__ __forceinline__ pk2 pk_splat(float a) { pk2 v = {a, a}; return v; }
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) {
#if defined(__clang__)
    return __builtin_elementwise_fma(a, b, c);
#else
    pk2 r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    return r;
#endif
}

// ---- Philox4x32-10 ------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                             unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // 32x32 -> 64 products: one v_mad_u64_u32 each instead of a v_mul_hi_u32 + v_mul_lo_u32 pair (both quarter rate)
    const unsigned long long p0 = (unsigned long long)M0 * (unsigned long long)c0;
    const unsigned long long p1 = (unsigned long long)M1 * (unsigned long long)c2;
    unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// eps(seed, gid, attempt, k): counter = (k, attempt, gid_lo, gid_hi), key = (seed_lo, seed_hi);
// u = ((x >> 9) + 0.5) * 2^-23;  z0 = r0 cos(2 pi u1), z1 = r0 sin(2 pi u1), z2 = r1 cos(2 pi u3)
// with r0 = sqrt(-2 ln u0), r1 = sqrt(-2 ln u2).
__device__ __forceinline__ Normal3 keyed_normal3(unsigned seed_lo, unsigned seed_hi, unsigned gid_lo,
                                                 unsigned gid_hi, unsigned attempt, unsigned k) {
    unsigned c0 = k, c1 = attempt, c2 = gid_lo, c3 = gid_hi;
    unsigned k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float scale = 1.1920928955078125e-07f;  // 2^-23
    float u0 = ((float)(c0 >> 9) + 0.5f) * scale;
    float u1 = ((float)(c1 >> 9) + 0.5f) * scale;
    float u2 = ((float)(c2 >> 9) + 0.5f) * scale;
    float u3 = ((float)(c3 >> 9) + 0.5f) * scale;
    float r0 = sqrtf(-2.0f * logf(u0));
    float r1 = sqrtf(-2.0f * logf(u2));
    float s0, cs0;
    sincospif(2.0f * u1, &s0, &cs0);
    float cs1 = cospif(2.0f * u3);
    Normal3 o;
    o.x = r0 * cs0;
    o.y = r0 * s0;
    o.z = r1 * cs1;
    return o;
}

// ---- exp(x) in double from IEEE operations only, rounded ONCE to f32 ----------------------------------------------------
// k = rint(x log2 e), r = x - k ln2 (two-part ln2: exact products), exp(r) by its Taylor series to r^14 (|r| <= 0.347:
// truncation 1e-19) in Horner form with explicit fma, scaled by 2^k.  The double result is within ~2e-16 of exp(x), so its
// rounding to float IS the correctly rounded f32 exponential except where exp(x) lies within 2e-16 of a rounding boundary
// (~1e-8 of all arguments) -- and, unlike a library exp (the device's double exp landed 1.6 % of its results on the other
// side of an f32 rounding boundary: 7.5 % of the covariance rows differed from the host build's, 2.8 % now, the same rows on
// the MI355X and in the CPU emulator build), it is the SAME function on every target: only +, *, fma, rint.
__device__ __forceinline__ float exp_cr(float xf) {
    const double x = (double)xf;
    if (!(x > -104.0)) return x != x ? xf : 0.0f;                 // below the smallest denormal (NaN passes through)
    if (x > 88.8) return __builtin_inff();
    const double kd = rint(x * 1.4426950408889634074);
    double r = fma(-kd, 6.93147180369123816490e-01, x);
    r = fma(-kd, 1.90821492927058770002e-10, r);
    double p = 1.1470745597729725e-11;                            // 1/14!
    p = fma(p, r, 1.6059043836821613e-10);                        // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                          // 1/12!
    p = fma(p, r, 2.505210838544172e-08);                         // 1/11!
    p = fma(p, r, 2.755731922398589e-07);                         // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                        // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                          // 1/8!
    p = fma(p, r, 0.0001984126984126984);                         // 1/7!
    p = fma(p, r, 0.001388888888888889);                          // 1/6!
    p = fma(p, r, 0.008333333333333333);                          // 1/5!
    p = fma(p, r, 0.041666666666666664);                          // 1/4!
    p = fma(p, r, 0.16666666666666666);                           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long k = (long long)kd;                            // |k| <= 151: 2^k is a normal double
    union { long long i; double d; } two_k;
    two_k.i = (k + 1023) << 52;
    return (float)(p * two_k.d);
}

// ---- symmetric 3x3 eigenvalues (trigonometric closed form, fp64) --------------------------------
__device__ __forceinline__ void sym3_eigvals(double a00, double a01, double a02, double a11, double a12,
                                             double a22, double e[3]) {
    double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    double q = (a00 + a11 + a22) / 3.0;
    double d0 = a00 - q, d1 = a11 - q, d2 = a22 - q;
    double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
    if (!(p2 > 0.0)) { e[0] = e[1] = e[2] = q; return; }
    double p = sqrt(p2 / 6.0);
    double ip = 1.0 / p;
    double b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = a01 * ip, b02 = a02 * ip, b12 = a12 * ip;
    double det = b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02);
    double r = 0.5 * det;
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    double phi = acos(r) / 3.0;
    double hi = q + 2.0 * p * cos(phi);
    double lo = q + 2.0 * p * cos(phi + 2.0943951023931954923);  // + 2 pi / 3
    double mid = 3.0 * q - hi - lo;
    e[0] = lo; e[1] = mid; e[2] = hi;
}

// Child level of another layout (G2pcTileLayout.tile_parent, G2PC_TILE_PARENTS entries per tile, -1 = none): does this tile exist
// for the camera whose first pass left `alive` (one byte per parent tile: "split")?  A tile can be the child of several parents:
// a child reaches one pixel beyond an odd-sized parent, and the neighbouring parent may have the very same rectangle among its own.
__device__ __forceinline__ bool child_exists(const int32_t* __restrict__ tile_parent, const uint8_t* __restrict__ alive, int t) {
    const int32_t* p = tile_parent + (size_t)G2PC_TILE_PARENTS * t;
    const int a = p[0], b = p[1], c = p[2], d = p[3];           // (one 16-byte load)
    return (a >= 0 && alive[a]) || (b >= 0 && alive[b]) || (c >= 0 && alive[c]) || (d >= 0 && alive[d]);
}

}  // namespace g2pc
