// field.hpp — Goldilocks / BabyBear base fields and their extension towers
// (reference: src/field.rs:36-109).  Canonical representatives everywhere:
// every value handed between kernels is < P, so results are bit-identical to
// arkworks' `into_bigint()` of the same field element.
//
//   Goldilocks  p = 2^64 - 2^32 + 1, generator 7           (field.rs:43-47)
//               Fp2 = Fp[u]/(u^2 - 7)                       (field.rs:50-62)
//   BabyBear    p = 2013265921, "generator" 440564289       (field.rs:72-76, quirk Q8)
//               Fp2 = Fp[u]/(u^2 - 11)                      (field.rs:78-91)
//               Fp4 = Fp2[v]/(v^2 - (2013265910 + u))       (field.rs:93-109)
#pragma once
#include "rt.hpp"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct GL {
  typedef u64 T;
  static constexpr int ID = 0;
  static constexpr u64 P = 0xFFFFFFFF00000001ULL;
  static constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
  static constexpr u64 GENERATOR = 7;
  static constexpr int TWO_ADICITY = 32;
  static constexpr int EXT = 2;  // StarkField::Extension = GoldilocksFp2 (field.rs:38-41)
  static constexpr u64 NR2 = 7;
  static constexpr int MAX_DIGITS = 20;
  // Compare + select formulation: the SHORTEST dependency chains (3-4 instructions deep), which is what the latency-bound kernels of
  // the path (single-workgroup scans and reductions, the FRI tail) want.  The throughput-bound NTT tiles use GLT below.
  static MS_HD u32 hi(u64 x) { return (u32)(x >> 32); }
  static MS_HD u32 lo(u64 x) { return (u32)x; }
  static MS_HD u64 mk(u32 l, u32 h) { return ((u64)h << 32) | l; }
  static MS_HD T add(T a, T b) {
    const T s = a + b;
    const T t = s - P;           // also the wrapped case: s + 2^64 - p == s - p (mod 2^64)
    return (s < a || s >= P) ? t : s;
  }
  static MS_HD T sub(T a, T b) {
    T d = a - b;
    if (a < b) d -= EPS;         // wrapped: - 2^64 == - EPS (mod p)
    return d;
  }
  static MS_HD T neg(T a) { return a ? P - a : 0; }
  // 128-bit product folded with 2^64 == 2^32 - 1 and 2^96 == -1 (mod p)
  static MS_HD T reduce128(u64 lo_, u64 hi_) {
    const u64 hi_hi = hi_ >> 32, hi_lo = hi_ & EPS;
    u64 t0 = lo_ - hi_hi;
    t0 -= (lo_ < hi_hi) ? EPS : 0;         // borrow: - 2^64 == - EPS
    const u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * EPS
    const u64 r = t0 + t1;
    const u64 r2 = r + EPS;                // r - p (mod 2^64): the fix for a carry AND for p <= r < 2^64
    return (r < t1 || r >= P) ? r2 : r;
  }
  // 128-bit product as four 32x32+64 multiply-adds (v_mad_u64_u32): a*b and umul64hi separately cost two more multiplies
  static MS_HD T mul(T a, T b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);            // < 2^64: (2^32-1)^2 + 2^32 - 1
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    const u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return reduce128((p10 << 32) | (u32)p00, p11);
  }
  static MS_HD T from_u64(u64 v) { return v; }
  static MS_HD u64 to_u64(T v) { return v; }
  // twiddle tables hold to_tw(w); mul_tw(a, to_tw(w)) == a * w.  Nothing to gain for Goldilocks: identity.
  static MS_HD T to_tw(T w) { return w; }
  static MS_HD T tw_of(T w) { return w; }
  static MS_HD T mul_tw(T a, T w_tab) { return mul(a, w_tab); }
  static MS_HD T from_tw(T w_tab) { return w_tab; }
};

// Goldilocks, THROUGHPUT formulation (used by the NTT tiles): the same canonical results, decided on the sign bits of the high words
// with v_bitop3_b32 instead of 64-bit compares and selects.  Measured on MI355X (tools/ntt_lab.hip): 5-13 % fewer issue cycles per
// butterfly stage in a register-resident radix-16/32 network (11.7 instead of 12.6 VALU instructions per element-stage, no SGPR
// round trips), but one instruction deeper per operation - slower in the latency-bound single-workgroup kernels, which keep GL.
struct GLT : GL {
  static MS_HD u64 sel(u32 m, u64 t, u64 s) { return mk(ms_bitop3<0xCA>(m, lo(t), lo(s)), ms_bitop3<0xCA>(m, hi(t), hi(s))); }  // m ? t : s (m = 0 / ~0)
  static MS_HD T add(T a, T b) {
    const u64 s = a + b, t = s + EPS;                              // t = s - P (mod 2^64)
    const u32 c1 = ms_bitop3<0xD4>(hi(a), hi(b), hi(s));           // bit 31: carry out of a + b = maj(a, b, ~s)
    const u32 m = ms_sar31(ms_bitop3<0xF4>(c1, hi(s), hi(t)));     // ... or s >= P, i.e. carry out of s + EPS = s & ~t
    return sel(m, t, s);
  }
  static MS_HD T sub(T a, T b) {
    const u64 d = a - b;
    const u32 bo = ms_bitop3<0x8E>(hi(a), hi(b), hi(d));           // bit 31: borrow out of a - b
    return d + ms_pin64(mk(bo >> 31, ms_sar31(bo)));               // + P on a borrow
  }
  // A + h * 2^64 for h < 2^31  (2^64 == EPS): one multiply-add, the wrap / >= P decision on sign bits
  static MS_HD T fold_small(u64 A, u32 h) {
    const u64 r = (u64)h * 0xFFFFFFFFu + A, t = r + EPS;
    // h*EPS < 2^63: the sum wrapped iff A's top bit is set and r's is clear (then r < 2^63 and r + EPS is the answer);
    // otherwise r >= P iff r + EPS wraps (r's top bit set, t's clear)
    return sel(ms_sar31(ms_bitop3<0x74>(hi(A), hi(r), hi(t))), t, r);
  }
  // 128-bit value folded with 2^64 == 2^32 - 1 and 2^96 == -1 (mod p):  lo - hi_hi + hi_lo * EPS
  static MS_HD T reduce128(u64 lo_, u64 hi_) {
    const u32 h0 = lo(hi_), h1 = hi(hi_);
    const u64 A = lo_ - h1;                                        // wrapped iff lo < h1 < 2^32: lo's top bit clear, A's set
    const u32 b1 = ms_bitop3<0x0C>(hi(lo_), hi(A), 0u);
    const u64 U = (u64)h0 * 0xFFFFFFFFu;
    const u64 r = A + U, t = r + EPS;
    const u32 c1 = ms_bitop3<0xD4>(hi(A), hi(U), hi(r));           // carry out of A + U
    const u32 c2 = ms_bitop3<0x30>(hi(r), hi(t), 0u);              // r >= P (when nothing wrapped)
    // true value = r + (c1 - b1) * 2^64: +EPS if only the carry (or, with neither, r >= P), -EPS == +P if only the borrow
    const u32 mm = ms_bitop3<0x30>(b1, c1, 0u);
    const u64 rm = r + ms_pin64(mk(mm >> 31, ms_sar31(mm)));
    return sel(ms_sar31(ms_bitop3<0x0E>(b1, c1, c2)), t, rm);
  }
  // 128-bit product as four 32x32+64 multiply-adds (v_mad_u64_u32): a*b and umul64hi separately cost two more multiplies
  static MS_HD T mul(T a, T b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);            // < 2^64: (2^32-1)^2 + 2^32 - 1
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    const u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return reduce128((p10 << 32) | (u32)p00, p11);
  }
  // z * 2^32 and z * 2^64 for canonical z = z0 + z1 * 2^32
  static MS_HD T mul_x32(T z) {                                     // z0 * 2^32 + z1 * EPS  (< 2^65)
    const u64 L = (u64)lo(z) << 32, U = (u64)hi(z) * 0xFFFFFFFFu;
    const u64 r = L + U, t = r + EPS;
    const u32 c1 = ms_bitop3<0xD4>(hi(L), hi(U), hi(r));
    return sel(ms_sar31(ms_bitop3<0xF4>(c1, hi(r), hi(t))), t, r);
  }
  static MS_HD T mul_x64(T z) {                                     // z0 * EPS - z1  (z0 >= 1 keeps it >= 0; both < P)
    const u64 U = (u64)lo(z) * 0xFFFFFFFFu;
    const u64 r = U - hi(z);
    const u32 bo = ms_bitop3<0x0C>(hi(U), hi(r), 0u);              // wrapped iff U < z1 < 2^32
    return r + ms_pin64(mk(bo >> 31, ms_sar31(bo)));
  }
  static MS_HD T mul_tw(T a, T w_tab) { return mul(a, w_tab); }
};

// Goldilocks, EXEC-MASKED formulation (inline asm, gfx950 only; every other build falls back to GL's formulas): the conditional
// corrections of add / sub / fold / reduce run as ONE 64-bit add under an exec mask set from the compare's SGPR result instead of
// compare + select pairs: 4 VALU instructions per add or sub (7 / 6 in GL / GLT), 19 per general multiplication (27), SALU does the
// mask plumbing.  Measured (tools/ntt_lab.hip): 7.8 VALU instructions and 28 issue cycles per element-stage of a radix-32 butterfly
// network against 11.7 / 38 (GLT) - provided the SIMD holds >= 4 waves; at 2 waves per SIMD the exec writes are not hidden and it
// loses to GLT (42 vs 39).  The NTT tiles choose the class by their occupancy.
// Hazards handled inside the blocks: a VALU read of an SGPR/VCC written by the VALU instruction before it needs 2 wait states on
// gfx950 (s_nop 1, or an SALU copy in between); SALU reads of VALU-written SGPRs and VALU after an SALU exec write are interlocked.
struct GLM : GL {
#if defined(__HIP_DEVICE_COMPILE__)
  static __device__ __forceinline__ T add(T a, T b) {
    u64 s, sv;
    asm("v_lshl_add_u64 %0, %2, 0, %3\n\t"
        "v_cmp_lt_u64 vcc, %0, %2\n\t"            // carry out of a + b
        "v_cmp_lt_u64 %1, %4, %0\n\t"             // s > P - 1
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %5\n\t"        // s - P == s + EPS (mod 2^64)
        "s_mov_b64 exec, %1"
        : "=&v"(s), "=&s"(sv) : "v"(a), "v"(b), "s"(P - 1), "s"(EPS) : "vcc", "scc");
    return s;
  }
  // r03: the borrow comes out of v_subb_co_u32 as an SGPR pair - no v_cmp_lt_u64: 3 VALU instructions (4).  Two statements: the first
  // leaves exec alone, so whatever the compiler schedules between them runs under the caller's mask.
  static __device__ __forceinline__ T sub(T a, T b) {
    u32 d0, d1; u64 bm, sv;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"                            // VALU wrote vcc, the next VALU reads it as carry-in: 2 wait states on gfx950
        "v_subb_co_u32 %1, %2, %4, %6, vcc"        // borrow out -> SGPR pair
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(lo(a)), "v"(hi(a)), "v"(lo(b)), "v"(hi(b)) : "vcc");
    u64 d = mk(d0, d1);
    asm("s_and_saveexec_b64 %1, %2\n\t"
        "v_lshl_add_u64 %0, %0, 0, %3\n\t"        // + P
        "s_mov_b64 exec, %1"
        : "+v"(d), "=&s"(sv) : "s"(bm), "s"(P) : "scc");
    return d;
  }
  // A + h * EPS, canonical, for any u64 A and h < 2^32 (the true sum is < 2^65 - 2^33: at most one wrap, and a wrapped sum + EPS < P)
  static __device__ __forceinline__ T fold(u64 A, u32 h) {
    u64 sv;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"   // carry out in vcc
        "v_cmp_lt_u64 %1, %3, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(A), "=&s"(sv) : "v"(h), "s"(P - 1), "s"(EPS) : "vcc", "scc");
    return A;
  }
  static __device__ __forceinline__ T fold_small(u64 A, u32 h) { return fold(A, h); }
  static __device__ __forceinline__ T mul_x32(T z) { return fold(z << 32, hi(z)); }      // z0 * 2^32 + z1 * EPS
  static __device__ __forceinline__ T mul_x64(T z) {                                      // z0 * EPS - z1
    const u64 U = (u64)lo(z) * 0xFFFFFFFFu;
    u32 d0, d1; u64 bm, sv;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, %2, 0, %4, vcc"
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(lo(U)), "v"(hi(U)), "v"(hi(z)) : "vcc");
    u64 r = mk(d0, d1);
    asm("s_and_saveexec_b64 %1, %2\n\t"
        "v_lshl_add_u64 %0, %0, 0, %3\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(r), "=&s"(sv) : "s"(bm), "s"(P) : "scc");
    return r;
  }
  static __device__ __forceinline__ T mul(T a, T b) {
    u64 T0, M, T1, c;
    asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1"        // M = a0*b1 + a1*b0, its carry in an SGPR pair
        : "=&v"(T0), "=&v"(M), "=&v"(T1), "=&s"(c) : "v"(lo(a)), "v"(hi(a)), "v"(lo(b)), "v"(hi(b)) : "vcc");
    // 128-bit (lo, hi) = T0 + M * 2^32 + T1 * 2^64 + c * 2^96;  r = lo - hi_hi, then + hi_lo * EPS
    u32 L1, H0, H1, R0, R1; u64 bm;
#define MS_NOP1 "s_nop 1\n\t"
#define MS_NOP0 "s_nop 0\n\t"
    asm(MS_NOP1                                     // c was written by the VALU instruction that ended the previous statement
        "v_addc_co_u32 %2, vcc, %9, 0, %11\n\t"   // H1 = hi(T1) + c   (no wrap: the product is < 2^128)
        "v_add_co_u32 %0, vcc, %6, %7\n\t"        // L1 = hi(T0) + lo(M)
        MS_NOP1
        "v_addc_co_u32 %1, vcc, %10, %8, vcc\n\t" // H0 = lo(T1) + hi(M) + carry
        MS_NOP1
        "v_addc_co_u32 %2, vcc, %2, 0, vcc\n\t"   // H1 += carry
        MS_NOP0
        "v_sub_co_u32 %3, vcc, %12, %2\n\t"       // r = lo - H1
        MS_NOP1
        "v_subbrev_co_u32 %4, vcc, 0, %0, vcc\n\t"
        "s_mov_b64 %5, vcc"                       // borrow mask
        : "=&v"(L1), "=&v"(H0), "=&v"(H1), "=&v"(R0), "=&v"(R1), "=&s"(bm)
        : "v"(hi(T0)), "v"(lo(M)), "v"(hi(M)), "v"(hi(T1)), "v"(lo(T1)), "s"(c), "v"(lo(T0)) : "vcc");
    u64 R = mk(R0, R1), sv;
    asm("s_and_saveexec_b64 %1, %3\n\t"           // borrow: r += P  (== - EPS)
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"   // + hi_lo * EPS, carry in vcc
        "v_cmp_lt_u64 %1, %5, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %6\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(R), "=&s"(sv) : "v"(H0), "s"(bm), "s"(P), "s"(P - 1), "s"(EPS) : "vcc", "scc");
    return R;
  }
  static __device__ __forceinline__ T mul_tw(T a, T w_tab) { return mul(a, w_tab); }
#else
  static MS_HD T fold_small(u64 A, u32 h) { return GLT::fold_small(A, h); }
  static MS_HD T mul_x32(T z) { return GLT::mul_x32(z); }
  static MS_HD T mul_x64(T z) { return GLT::mul_x64(z); }
#endif
};

struct BB {
  typedef u32 T;
  static constexpr int ID = 1;
  static constexpr u64 P = 2013265921ULL;
  static constexpr u64 GENERATOR = 440564289ULL;
  static constexpr int TWO_ADICITY = 27;
  static constexpr int EXT = 4;  // StarkField::Extension = BabyBearFp4 (field.rs:67-70)
  static constexpr u64 NR2 = 11;
  static constexpr int MAX_DIGITS = 10;
  // a + b < 2^32 (p < 2^31): the reduced value is the smaller of s and s - p as unsigned numbers (s < p: s - p wraps to something huge); same for a - b
  static MS_HD T add(T a, T b) { const u32 s = a + b, t = s - (u32)P; return s < t ? s : t; }
  static MS_HD T sub(T a, T b) { const u32 d = a - b, t = d + (u32)P; return d < t ? d : t; }
  static MS_HD T neg(T a) { return a ? (u32)P - a : 0; }
  // Montgomery reduction (R = 2^32): x * 2^-32 mod p for x < p * 2^32;  MU = -p^-1 mod 2^32
  static constexpr u32 MU = 0x77FFFFFFu;
  static constexpr u32 R2 = 1172168163u;  // 2^64 mod p
  static MS_HD T redc(u64 x) {
    const u32 m = (u32)x * MU;
    const u32 t = (u32)((x + (u64)m * (u64)P) >> 32);  // x + m*p < 2^63 + 2^63; t < 2p
    // r05: the final correction as in add / sub - the smaller of t and t - p as unsigned numbers (t < p: t - p wraps above t) = v_sub + v_min_u32.  The
    // `t >= p ? t - p : t` it replaces compiled to v_cmp_lt_u64 + v_add + v_cndmask, ~12 issue cycles against ~6 (tools/isa_census.py on the NTT pass kernels:
    // one in twelve VALU instructions of a BabyBear pass was that compare; tools/valu_rate.hip: compares and v_cndmask issue at half the rate of adds)
    const u32 r = t - (u32)P;
    return r < t ? r : t;
  }
  // canonical in, canonical out: (a*b/R) * R^2 / R  — two 32x32 products + two reductions instead of a 64-bit modulo
  static MS_HD T mul(T a, T b) { return redc((u64)redc((u64)a * (u64)b) * (u64)R2); }
  static MS_HD T from_u64(u64 v) { return (T)v; }
  static MS_HD u64 to_u64(T v) { return v; }
  // twiddle tables hold the Montgomery form w * 2^32 mod p: a canonical value times a table entry is ONE product and ONE
  // reduction (a * wR / R = a * w), and the product of two table entries is again in table form
  static MS_HD T to_tw(T w) { return (T)((((u64)w) << 32) % P); }
  static MS_HD T tw_of(T w) { return redc((u64)w * (u64)R2); }   // == to_tw(w) without the 64-bit modulo (device code: run-time values)
  static MS_HD T mul_tw(T a, T w_tab) { return redc((u64)a * (u64)w_tab); }
  static MS_HD T from_tw(T w_tab) { return redc((u64)w_tab); }
};

template <class F> MS_HD typename F::T f_pow(typename F::T a, u64 e) {
  typename F::T r = F::from_u64(1);
  while (e) { if (e & 1) r = F::mul(r, a); a = F::mul(a, a); e >>= 1; }
  return r;
}
template <class F> MS_HD typename F::T f_inv(typename F::T a) { return f_pow<F>(a, F::P - 2); }

// Goldilocks x^(p-2) with the 72-multiplication addition chain for p - 2 = 2^64 - 2^32 - 1 (63 squarings + 9 products)
MS_HD u64 gl_inv_chain(u64 x) {
  auto sqn = [](u64 v, int n) { for (int i = 0; i < n; i++) v = GL::mul(v, v); return v; };
  const u64 t2 = GL::mul(GL::mul(x, x), x);            // x^(2^2 - 1)
  const u64 t3 = GL::mul(GL::mul(t2, t2), x);          // x^(2^3 - 1)
  const u64 t6 = GL::mul(sqn(t3, 3), t3);
  const u64 t12 = GL::mul(sqn(t6, 6), t6);
  const u64 t24 = GL::mul(sqn(t12, 12), t12);
  const u64 t30 = GL::mul(sqn(t24, 6), t6);
  const u64 t31 = GL::mul(GL::mul(t30, t30), x);       // x^(2^31 - 1)
  const u64 t63 = GL::mul(sqn(t31, 32), t31);          // x^(2^63 - 2^32 + 2^31 - 1)
  return GL::mul(GL::mul(t63, t63), x);                // x^(2^64 - 2^32 - 1)
}

// [ark-mem] Radix2EvaluationDomain::new(n).group_gen: GENERATOR^((p-1)/2^s) squared (s - log2 n) times
template <class F> inline typename F::T f_root_of_unity(int log_n) {
  typename F::T w = f_pow<F>(F::from_u64(F::GENERATOR), (F::P - 1) >> F::TWO_ADICITY);
  for (int i = log_n; i < F::TWO_ADICITY; i++) w = F::mul(w, w);
  return w;
}

// ---------------------------------------------------------------------------
// Extension elements: E base limbs in registers.  E == 1 is the base field.
// ---------------------------------------------------------------------------
template <class F, int E> struct Ext { typename F::T c[E]; };

template <class F, int E> MS_HD Ext<F, E> e_zero() { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = 0; return r; }
template <class F, int E> MS_HD Ext<F, E> e_from_base(typename F::T b) { Ext<F, E> r = e_zero<F, E>(); r.c[0] = b; return r; }
template <class F, int E> MS_HD Ext<F, E> e_one() { return e_from_base<F, E>(F::from_u64(1)); }
template <class F, int E> MS_HD Ext<F, E> e_add(const Ext<F, E>& a, const Ext<F, E>& b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::add(a.c[i], b.c[i]); return r; }
template <class F, int E> MS_HD Ext<F, E> e_sub(const Ext<F, E>& a, const Ext<F, E>& b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::sub(a.c[i], b.c[i]); return r; }
template <class F, int E> MS_HD Ext<F, E> e_mul_base(const Ext<F, E>& a, typename F::T b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::mul(a.c[i], b); return r; }
template <class F, int E> MS_HD bool e_is_zero(const Ext<F, E>& a) { bool z = true; for (int i = 0; i < E; i++) z = z && (a.c[i] == 0); return z; }
template <class F, int E> MS_HD bool e_eq(const Ext<F, E>& a, const Ext<F, E>& b) { bool z = true; for (int i = 0; i < E; i++) z = z && (a.c[i] == b.c[i]); return z; }

template <class F> MS_HD Ext<F, 1> e_mul(const Ext<F, 1>& a, const Ext<F, 1>& b) { Ext<F, 1> r; r.c[0] = F::mul(a.c[0], b.c[0]); return r; }
template <class F> MS_HD Ext<F, 2> e_mul(const Ext<F, 2>& a, const Ext<F, 2>& b) {
  // Karatsuba: 3 base multiplications + one by the small non-residue
  typename F::T v0 = F::mul(a.c[0], b.c[0]);
  typename F::T v1 = F::mul(a.c[1], b.c[1]);
  typename F::T s = F::mul(F::add(a.c[0], a.c[1]), F::add(b.c[0], b.c[1]));
  Ext<F, 2> r;
  r.c[0] = F::add(v0, F::mul_tw(v1, F::to_tw(F::from_u64(F::NR2))));   // the constant in table form (folded at compile time): one product + one reduction for BabyBear
  r.c[1] = F::sub(F::sub(s, v0), v1);
  return r;
}
// multiply an Fp2 element by the quartic non-residue (2013265910 + u) = (u - 11) (BabyBear)
template <class F> MS_HD Ext<F, 2> e_mul_nr4(const Ext<F, 2>& a) {
  // (a0 + a1 u)(n0 + u) = (a0 n0 + NR2 a1) + (a0 + a1 n0) u ,  n0 = 2013265910 = -11 mod p
  const typename F::T n0 = F::to_tw(F::from_u64(2013265910ULL % F::P)), nr = F::to_tw(F::from_u64(F::NR2));   // table form, compile-time constants
  Ext<F, 2> r;
  r.c[0] = F::add(F::mul_tw(a.c[0], n0), F::mul_tw(a.c[1], nr));
  r.c[1] = F::add(a.c[0], F::mul_tw(a.c[1], n0));
  return r;
}
template <class F> MS_HD Ext<F, 4> e_mul(const Ext<F, 4>& a, const Ext<F, 4>& b) {
  Ext<F, 2> a0{{a.c[0], a.c[1]}}, a1{{a.c[2], a.c[3]}}, b0{{b.c[0], b.c[1]}}, b1{{b.c[2], b.c[3]}};
  Ext<F, 2> v0 = e_mul<F>(a0, b0), v1 = e_mul<F>(a1, b1);
  Ext<F, 2> s = e_mul<F>(e_add<F, 2>(a0, a1), e_add<F, 2>(b0, b1));
  Ext<F, 2> r0 = e_add<F, 2>(v0, e_mul_nr4<F>(v1));
  Ext<F, 2> r1 = e_sub<F, 2>(e_sub<F, 2>(s, v0), v1);
  Ext<F, 4> r; r.c[0] = r0.c[0]; r.c[1] = r0.c[1]; r.c[2] = r1.c[0]; r.c[3] = r1.c[1];
  return r;
}
// a * b with b's limbs in TABLE form (e_to_tw): every base product is F::mul_tw - one product and one reduction for BabyBear instead of two (Goldilocks: the same
// code as e_mul).  Linear in a, so a in table form gives the product in table form.  For loop-invariant multiplicands (the point of a Horner scan, the step of a power).
template <class F, int E> MS_HD Ext<F, E> e_to_tw(const Ext<F, E>& b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::tw_of(b.c[i]); return r; }
template <class F> MS_HD Ext<F, 1> e_mul_tw(const Ext<F, 1>& a, const Ext<F, 1>& bt) { Ext<F, 1> r; r.c[0] = F::mul_tw(a.c[0], bt.c[0]); return r; }
template <class F> MS_HD Ext<F, 2> e_mul_tw(const Ext<F, 2>& a, const Ext<F, 2>& bt) {
  typename F::T v0 = F::mul_tw(a.c[0], bt.c[0]);
  typename F::T v1 = F::mul_tw(a.c[1], bt.c[1]);
  typename F::T s = F::mul_tw(F::add(a.c[0], a.c[1]), F::add(bt.c[0], bt.c[1]));
  Ext<F, 2> r;
  r.c[0] = F::add(v0, F::mul_tw(v1, F::to_tw(F::from_u64(F::NR2))));
  r.c[1] = F::sub(F::sub(s, v0), v1);
  return r;
}
template <class F> MS_HD Ext<F, 4> e_mul_tw(const Ext<F, 4>& a, const Ext<F, 4>& bt) {
  Ext<F, 2> a0{{a.c[0], a.c[1]}}, a1{{a.c[2], a.c[3]}}, b0{{bt.c[0], bt.c[1]}}, b1{{bt.c[2], bt.c[3]}};
  Ext<F, 2> v0 = e_mul_tw<F>(a0, b0), v1 = e_mul_tw<F>(a1, b1);
  Ext<F, 2> s = e_mul_tw<F>(e_add<F, 2>(a0, a1), e_add<F, 2>(b0, b1));
  Ext<F, 2> r0 = e_add<F, 2>(v0, e_mul_nr4<F>(v1));
  Ext<F, 2> r1 = e_sub<F, 2>(e_sub<F, 2>(s, v0), v1);
  Ext<F, 4> r; r.c[0] = r0.c[0]; r.c[1] = r0.c[1]; r.c[2] = r1.c[0]; r.c[3] = r1.c[1];
  return r;
}
template <class F, int E> MS_HD Ext<F, E> e_pow(Ext<F, E> a, u64 e) {
  Ext<F, E> r = e_one<F, E>();
  while (e) { if (e & 1) r = e_mul<F>(r, a); a = e_mul<F>(a, a); e >>= 1; }
  return r;
}

// host-side inverses (only used for small per-call scalars)
template <class F> inline Ext<F, 1> e_inv(const Ext<F, 1>& a) { Ext<F, 1> r; r.c[0] = f_inv<F>(a.c[0]); return r; }
template <class F> inline Ext<F, 2> e_inv(const Ext<F, 2>& a) {
  typename F::T n = F::sub(F::mul(a.c[0], a.c[0]), F::mul(F::from_u64(F::NR2), F::mul(a.c[1], a.c[1])));
  typename F::T ni = f_inv<F>(n);
  Ext<F, 2> r; r.c[0] = F::mul(a.c[0], ni); r.c[1] = F::mul(F::neg(a.c[1]), ni);
  return r;
}

template <class F> inline Ext<F, 4> e_inv(const Ext<F, 4>& a) {   // (a0 + a1 v)^-1 = (a0 - a1 v) / (a0^2 - (u - 11) a1^2), v^2 = u - 11
  const Ext<F, 2> a0{{a.c[0], a.c[1]}}, a1{{a.c[2], a.c[3]}};
  const Ext<F, 2> di = e_inv<F>(e_sub<F, 2>(e_mul<F>(a0, a0), e_mul_nr4<F>(e_mul<F>(a1, a1))));
  const Ext<F, 2> r0 = e_mul<F>(a0, di), r1 = e_mul<F>(a1, di);
  Ext<F, 4> r; r.c[0] = r0.c[0]; r.c[1] = r0.c[1]; r.c[2] = F::neg(r1.c[0]); r.c[3] = F::neg(r1.c[1]);
  return r;
}

// SoA view of a vector of extension elements: limb k of element j at p[k*limb_stride + off + j*stride]
template <class F, int E> struct ExtView {
  typename F::T* p;
  size_t limb_stride;
  MS_HD Ext<F, E> load(size_t j) const { Ext<F, E> r; for (int k = 0; k < E; k++) r.c[k] = p[k * limb_stride + j]; return r; }
  MS_HD void store(size_t j, const Ext<F, E>& v) const { for (int k = 0; k < E; k++) p[k * limb_stride + j] = v.c[k]; }
};
