// ntt_lab.hip — arithmetic micro-benchmark behind the NTT pass design (MI355X): radix-2^B DIF butterfly networks on register-resident
// Goldilocks values, no memory traffic, for several formulations of the modular add / sub / shift-multiply.  Reports ns per
// butterfly-stage-element and VALU instructions are read off the ISA (hipcc -S).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../mini-stark_amd/csrc/ntt.hpp"

// ---- variant 0: the r01 formulation (compare + select) = field.hpp GL
struct GL0 {
  static MS_HD u64 add(u64 a, u64 b) { return GL::add(a, b); }
  static MS_HD u64 sub(u64 a, u64 b) { return GL::sub(a, b); }
  template <int S> static MS_HD u64 mul_pow2(u64 x) { return msntt::gl_mul_pow2_v1<S>(x); }
};
// ---- variant 1: the r02 formulation (sign-bit decisions via v_bitop3)
struct GL1 {
  static MS_HD u64 add(u64 a, u64 b) { return GLT::add(a, b); }
  static MS_HD u64 sub(u64 a, u64 b) { return GLT::sub(a, b); }
  template <int S> static MS_HD u64 mul_pow2(u64 x) { return msntt::gl_mul_pow2<S>(x); }
};
// ---- variant 2: LAZY (NOT exact: single-fix add/sub on [0, 2^64), for cost exploration only)
struct GL2 {
  static MS_HD u64 add(u64 a, u64 b) { const u64 s = a + b; const u32 c = ms_bitop3<0xD4>(GL::hi(a), GL::hi(b), GL::hi(s)); return s + ms_pin64(GL::mk(ms_sar31(c), 0u)); }
  static MS_HD u64 sub(u64 a, u64 b) { const u64 d = a - b; const u32 bo = ms_bitop3<0x8E>(GL::hi(a), GL::hi(b), GL::hi(d)); return d + ms_pin64(GL::mk(bo >> 31, ms_sar31(bo))); }
  template <int S> static MS_HD u64 mul_pow2(u64 x) { return msntt::gl_mul_pow2<S>(x); }
};

// ---- variant 3: canonical, corrections under an EXEC mask instead of selects (inline asm; SALU does the mask plumbing)
struct GL3 {
  static __device__ __forceinline__ u64 add(u64 a, u64 b) {
    u64 s, sv;
    asm("v_lshl_add_u64 %0, %2, 0, %3\n\t"
        "v_cmp_lt_u64 vcc, %0, %2\n\t"
        "v_cmp_lt_u64 %1, %4, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %5\n\t"
        "s_mov_b64 exec, %1"
        : "=&v"(s), "=&s"(sv) : "v"(a), "v"(b), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
    return s;
  }
  static __device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u64 d = a - b, sv;
    asm("v_cmp_lt_u64 vcc, %2, %3\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(d), "=&s"(sv) : "v"(a), "v"(b), "s"(GL::P) : "vcc", "scc");
    return d;
  }
  // A + h * EPS, canonical (A any u64, h < 2^32 with A + h*EPS < 2^65 - ...: one wrap at most)
  static __device__ __forceinline__ u64 fold(u64 A, u32 h) {
    u64 sv;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"
        "v_cmp_lt_u64 %1, %3, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(A), "=&s"(sv) : "v"(h), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
    return A;
  }
  static __device__ __forceinline__ u64 mul_x32(u64 z) { return fold(z << 32, GL::hi(z)); }
  static __device__ __forceinline__ u64 mul_x64(u64 z) {       // z0 * EPS - z1
    u64 U = (u64)GL::lo(z) * 0xFFFFFFFFu;
    u64 r = U - GL::hi(z), sv;
    asm("v_cmp_lt_u64 vcc, %2, %3\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(r), "=&s"(sv) : "v"(U), "v"((u64)GL::hi(z)), "s"(GL::P) : "vcc", "scc");
    return r;
  }
  template <int S> static __device__ __forceinline__ u64 mul_pow2(u64 x) {
    if constexpr (S == 0) return x;
    else if constexpr (S < 32) return fold(x << S, GL::hi(x) >> (32 - S));
    else if constexpr (S < 64) return mul_x32(mul_pow2<S - 32>(x));
    else return mul_x64(mul_pow2<S - 64>(x));
  }
  static __device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u64 T0, M, T1, c;
    asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1"
        : "=&v"(T0), "=&v"(M), "=&v"(T1), "=&s"(c) : "v"(GL::lo(a)), "v"(GL::hi(a)), "v"(GL::lo(b)), "v"(GL::hi(b)) : "vcc");
    // 128-bit (lo, hi) = T0 + M * 2^32 + T1 * 2^64 + c * 2^96;  r = lo - hi_hi (+ hi_lo * EPS below)
    u32 L1, H0, H1, R0, R1; u64 bm;
    asm("s_nop 1\n\t"
        "v_addc_co_u32 %2, vcc, %9, 0, %11\n\t"          // H1 = hi(T1) + c   (no wrap: the product is < 2^128)
        "v_add_co_u32 %0, vcc, %6, %7\n\t"               // L1 = hi(T0) + lo(M)
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %10, %8, vcc\n\t"        // H0 = lo(T1) + hi(M) + carry
        "s_nop 1\n\t"
        "v_addc_co_u32 %2, vcc, %2, 0, vcc\n\t"          // H1 += carry
        "v_sub_co_u32 %3, vcc, %12, %2\n\t"              // r = lo - H1
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %4, vcc, 0, %0, vcc\n\t"
        "s_mov_b64 %5, vcc"
        : "=&v"(L1), "=&v"(H0), "=&v"(H1), "=&v"(R0), "=&v"(R1), "=&s"(bm)
        : "v"(GL::hi(T0)), "v"(GL::lo(M)), "v"(GL::hi(M)), "v"(GL::hi(T1)), "v"(GL::lo(T1)), "s"(c), "v"(GL::lo(T0)) : "vcc");
    u64 R = GL::mk(R0, R1), sv;
    asm("s_and_saveexec_b64 %1, %3\n\t"                  // borrow: r += P  (== - EPS)
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"          // + hi_lo * EPS, carry in vcc
        "v_cmp_lt_u64 %1, %5, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %6\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(R), "=&s"(sv) : "v"(H0), "s"(bm), "s"(GL::P), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
    return R;
  }
};

// ---- variant 5 (r03): variant 3 with the subtraction's borrow taken from v_subb_co_u32's carry-out (an SGPR pair) instead of a
// v_cmp_lt_u64: 3 VALU instructions per sub (4), also inside mul_x64.  Two asm statements: the first leaves exec alone.
struct GL5 : GL3 {
  static __device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u32 d0, d1; u64 bm, sv;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, %2, %4, %6, vcc"
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(GL::lo(a)), "v"(GL::hi(a)), "v"(GL::lo(b)), "v"(GL::hi(b)) : "vcc");
    u64 d = GL::mk(d0, d1);
    asm("s_and_saveexec_b64 %1, %2\n\t"
        "v_lshl_add_u64 %0, %0, 0, %3\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(d), "=&s"(sv) : "s"(bm), "s"(GL::P) : "scc");
    return d;
  }
  static __device__ __forceinline__ u64 mul_x64(u64 z) {       // z0 * EPS - z1
    const u64 U = (u64)GL::lo(z) * 0xFFFFFFFFu;
    u32 d0, d1; u64 bm, sv;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, %2, 0, %4, vcc"
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(GL::lo(U)), "v"(GL::hi(U)), "v"(GL::hi(z)) : "vcc");
    u64 r = GL::mk(d0, d1);
    asm("s_and_saveexec_b64 %1, %2\n\t"
        "v_lshl_add_u64 %0, %0, 0, %3\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(r), "=&s"(sv) : "s"(bm), "s"(GL::P) : "scc");
    return r;
  }
  template <int S> static __device__ __forceinline__ u64 mul_pow2(u64 x) {
    if constexpr (S == 0) return x;
    else if constexpr (S < 32) return fold(x << S, GL::hi(x) >> (32 - S));
    else if constexpr (S < 64) return mul_x32(mul_pow2<S - 32>(x));
    else return mul_x64(mul_pow2<S - 64>(x));
  }
};
// ---- variant 6 (r03): variant 5 under the assumption that EVERY lane of the wave is live (exec == -1 on entry): the masks go straight
// into exec (s_or_b64 exec / s_mov_b64 exec) and exec is restored with a constant - one scalar instruction less per operation and no
// dependency of the restore on a saved copy.  Only valid in straight-line, wave-uniform code.
struct GL6 {
  static __device__ __forceinline__ u64 add(u64 a, u64 b) {
    u64 s, sv;
    asm("v_lshl_add_u64 %0, %2, 0, %3\n\t"
        "v_cmp_lt_u64 vcc, %0, %2\n\t"
        "v_cmp_lt_u64 %1, %4, %0\n\t"
        "s_or_b64 exec, vcc, %1\n\t"
        "v_lshl_add_u64 %0, %0, 0, %5\n\t"
        "s_mov_b64 exec, -1"
        : "=&v"(s), "=&s"(sv) : "v"(a), "v"(b), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
    return s;
  }
  static __device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u32 d0, d1; u64 bm;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, %2, %4, %6, vcc"
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(GL::lo(a)), "v"(GL::hi(a)), "v"(GL::lo(b)), "v"(GL::hi(b)) : "vcc");
    u64 d = GL::mk(d0, d1);
    asm("s_mov_b64 exec, %1\n\t"
        "v_lshl_add_u64 %0, %0, 0, %2\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(d) : "s"(bm), "s"(GL::P));
    return d;
  }
  static __device__ __forceinline__ u64 fold(u64 A, u32 h) {
    u64 sv;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"
        "v_cmp_lt_u64 %1, %3, %0\n\t"
        "s_or_b64 exec, vcc, %1\n\t"
        "v_lshl_add_u64 %0, %0, 0, %4\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(A), "=&s"(sv) : "v"(h), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
    return A;
  }
  static __device__ __forceinline__ u64 mul_x32(u64 z) { return fold(z << 32, GL::hi(z)); }
  static __device__ __forceinline__ u64 mul_x64(u64 z) {
    const u64 U = (u64)GL::lo(z) * 0xFFFFFFFFu;
    u32 d0, d1; u64 bm;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, %2, 0, %4, vcc"
        : "=&v"(d0), "=&v"(d1), "=&s"(bm) : "v"(GL::lo(U)), "v"(GL::hi(U)), "v"(GL::hi(z)) : "vcc");
    u64 r = GL::mk(d0, d1);
    asm("s_mov_b64 exec, %1\n\t"
        "v_lshl_add_u64 %0, %0, 0, %2\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(r) : "s"(bm), "s"(GL::P));
    return r;
  }
  template <int S> static __device__ __forceinline__ u64 mul_pow2(u64 x) {
    if constexpr (S == 0) return x;
    else if constexpr (S < 32) return fold(x << S, GL::hi(x) >> (32 - S));
    else if constexpr (S < 64) return mul_x32(mul_pow2<S - 32>(x));
    else return mul_x64(mul_pow2<S - 64>(x));
  }
  static __device__ __forceinline__ u64 mul(u64 a, u64 b) { return GL3::mul(a, b); }
};

// ---- variant 4: variant 3 with TWO independent operations per asm block (instruction-level parallelism inside one wave:
// the SGPR round trips of one chain hide behind the other's VALU work), for tiles whose LDS footprint leaves only 2 waves per SIMD
struct GL4 : GL3 {
  // (s0, d0) = (a0 + b0, a0 - b0), (s1, d1) = (a1 + b1, a1 - b1); d0 / d1 come in as the wrapped differences
  static __device__ __forceinline__ void bfly2(u64 a0, u64 b0, u64 a1, u64 b1, u64& s0, u64& d0, u64& s1, u64& d1) {
    d0 = a0 - b0; d1 = a1 - b1;
    u64 sv, m0, m1, m2, m3;
    asm("v_lshl_add_u64 %0, %9, 0, %10\n\t"
        "v_lshl_add_u64 %1, %11, 0, %12\n\t"
        "v_cmp_lt_u64 %5, %9, %10\n\t"            // borrow 0
        "v_cmp_lt_u64 %6, %11, %12\n\t"           // borrow 1
        "v_cmp_lt_u64 vcc, %0, %9\n\t"            // carry 0
        "v_cmp_lt_u64 %7, %13, %0\n\t"            // s0 > P-1
        "v_cmp_lt_u64 %8, %1, %11\n\t"            // carry 1
        "s_or_b64 %7, %7, vcc\n\t"
        "v_cmp_lt_u64 vcc, %13, %1\n\t"           // s1 > P-1
        "s_and_saveexec_b64 %4, %5\n\t"
        "v_lshl_add_u64 %2, %2, 0, %15\n\t"       // d0 += P
        "s_and_b64 exec, %4, %6\n\t"
        "v_lshl_add_u64 %3, %3, 0, %15\n\t"       // d1 += P
        "s_and_b64 exec, %4, %7\n\t"
        "s_or_b64 %8, %8, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %14\n\t"       // s0 += EPS
        "s_and_b64 exec, %4, %8\n\t"
        "v_lshl_add_u64 %1, %1, 0, %14\n\t"       // s1 += EPS
        "s_mov_b64 exec, %4"
        : "=&v"(s0), "=&v"(s1), "+v"(d0), "+v"(d1), "=&s"(sv), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(GL::P - 1), "s"(GL::EPS), "s"(GL::P) : "vcc", "scc");
  }
  // two folds A + h * EPS
  static __device__ __forceinline__ void fold2(u64& A0, u32 h0, u64& A1, u32 h1) {
    u64 sv, m0, m1;
    asm("v_mad_u64_u32 %0, %3, %5, -1, %0\n\t"
        "v_mad_u64_u32 %1, %4, %6, -1, %1\n\t"
        "v_cmp_lt_u64 vcc, %7, %0\n\t"
        "s_or_b64 %3, %3, vcc\n\t"
        "v_cmp_lt_u64 vcc, %7, %1\n\t"
        "s_and_saveexec_b64 %2, %3\n\t"
        "v_lshl_add_u64 %0, %0, 0, %8\n\t"
        "s_or_b64 %4, %4, vcc\n\t"
        "s_and_b64 exec, %2, %4\n\t"
        "v_lshl_add_u64 %1, %1, 0, %8\n\t"
        "s_mov_b64 exec, %2"
        : "+v"(A0), "+v"(A1), "=&s"(sv), "=&s"(m0), "=&s"(m1) : "v"(h0), "v"(h1), "s"(GL::P - 1), "s"(GL::EPS) : "vcc", "scc");
  }
};
// shift-multiply pre-steps shared by the paired path: x * 2^S = fold-chain; stage k of value v: returns (A, h) of the next fold, or applies x64
template <int S> struct ShiftPlan {
  static constexpr int T = S % 32, Q = S / 32;   // 2^S = 2^T * (2^32)^Q
};
template <int S0, int S1> __device__ __forceinline__ void mul_pow2_pair(u64& x0, u64& x1) {
  // both go through the same kind of steps only when their (T != 0, Q) classes agree; otherwise fall back to the unpaired chain
  constexpr int T0 = S0 % 32, Q0 = S0 / 32, T1 = S1 % 32, Q1 = S1 / 32;
  if constexpr ((T0 != 0) && (T1 != 0)) {
    u64 A0 = x0 << T0, A1 = x1 << T1;
    GL4::fold2(A0, GL::hi(x0) >> (32 - T0), A1, GL::hi(x1) >> (32 - T1));
    x0 = A0; x1 = A1;
  } else { x0 = GL3::mul_pow2<T0>(x0); x1 = GL3::mul_pow2<T1>(x1); }
  if constexpr (Q0 >= 1 && Q1 >= 1 && Q0 == 1 && Q1 == 1) {
    u64 A0 = x0 << 32, A1 = x1 << 32;
    GL4::fold2(A0, GL::hi(x0), A1, GL::hi(x1));
    x0 = A0; x1 = A1;
  } else {
    if constexpr (Q0 == 1) x0 = GL3::mul_x32(x0); else if constexpr (Q0 == 2) x0 = GL3::mul_x64(x0);
    if constexpr (Q1 == 1) x1 = GL3::mul_x32(x1); else if constexpr (Q1 == 2) x1 = GL3::mul_x64(x1);
  }
}
template <int LOG2H2, int J> struct TwExp { static constexpr int E = (39 * (64 >> LOG2H2) * J) % 192; static constexpr bool NEG = E >= 96; static constexpr int S = NEG ? E - 96 : E; };
// paired radix-2^B DIF: butterflies two at a time
template <int B, int S, int BLK, int J> struct Stage2 {
  static __device__ __forceinline__ void run(u64 (&x)[1 << B]) {
    constexpr int h = 1 << S;
    if constexpr (h >= 2) {
      const u64 a0 = x[BLK + J], b0 = x[BLK + J + h], a1 = x[BLK + J + 1], b1 = x[BLK + J + 1 + h];
      typedef TwExp<S + 1, J> E0; typedef TwExp<S + 1, J + 1> E1;
      u64 s0, d0, s1, d1;
      // a negative twiddle multiplies (b - a): swap the operands of that difference
      if constexpr (!E0::NEG && !E1::NEG) GL4::bfly2(a0, b0, a1, b1, s0, d0, s1, d1);
      else {  // rare enough: unpaired differences, paired sums via the general block with swapped roles is not possible -> plain ops
        s0 = GL3::add(a0, b0); s1 = GL3::add(a1, b1);
        d0 = E0::NEG ? GL3::sub(b0, a0) : GL3::sub(a0, b0); d1 = E1::NEG ? GL3::sub(b1, a1) : GL3::sub(a1, b1);
      }
      mul_pow2_pair<(J == 0 ? 0 : E0::S), E1::S>(d0, d1);
      x[BLK + J] = s0; x[BLK + J + h] = d0; x[BLK + J + 1] = s1; x[BLK + J + 1 + h] = d1;
      if constexpr (J + 2 < h) Stage2<B, S, BLK, J + 2>::run(x);
      else if constexpr (BLK + 2 * h < (1 << B)) Stage2<B, S, BLK + 2 * h, 0>::run(x);
      else Stage2<B, S - 1, 0, 0>::run(x);
    } else {  // last stage: blocks of 2, paired two blocks at a time
      u64 s0, d0, s1, d1;
      GL4::bfly2(x[BLK], x[BLK + 1], x[BLK + 2], x[BLK + 3], s0, d0, s1, d1);
      x[BLK] = s0; x[BLK + 1] = d0; x[BLK + 2] = s1; x[BLK + 3] = d1;
      if constexpr (BLK + 4 < (1 << B)) Stage2<B, S, BLK + 4, 0>::run(x);
    }
  }
};

template <class A, int LOG2H2, int J> __device__ __forceinline__ u64 tw(u64 a, u64 b) {
  constexpr int EXP = (39 * (64 >> LOG2H2) * J) % 192;
  if constexpr (EXP >= 96) return A::template mul_pow2<EXP - 96>(A::sub(b, a));
  else return A::template mul_pow2<EXP>(A::sub(a, b));
}
template <class A, int B, int S, int BLK, int J> struct Stage {
  static __device__ __forceinline__ void run(u64 (&x)[1 << B]) {
    constexpr int h = 1 << S;
    u64 a = x[BLK + J], b = x[BLK + J + h];
    x[BLK + J] = A::add(a, b);
    if constexpr (J == 0) x[BLK + J + h] = A::sub(a, b); else x[BLK + J + h] = tw<A, S + 1, J>(a, b);
    if constexpr (J + 1 < h) Stage<A, B, S, BLK, J + 1>::run(x);
    else if constexpr (BLK + 2 * h < (1 << B)) Stage<A, B, S, BLK + 2 * h, 0>::run(x);
    else if constexpr (S > 0) Stage<A, B, S - 1, 0, 0>::run(x);
  }
};
template <class A, int B, int MODE> __global__ void __launch_bounds__(256) lab(u64* out, const u64* in, int iters) {
  u64 x[1 << B];
  const int tid = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < (1 << B); i++) x[i] = in[(tid * 7 + i * 13) & 4095];
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 4) Stage2<B, B - 1, 0, 0>::run(x);                               // paired masked-asm radix-2^B DIF
    if (MODE == 0) Stage<A, B, B - 1, 0, 0>::run(x);                                      // full radix-2^B DIF (adds, subs, shift twiddles)
    if (MODE == 1) { for (int i = 0; i < (1 << B); i += 2) { u64 a = x[i], b = x[i + 1]; x[i] = A::add(a, b); x[i + 1] = A::sub(a, b); } }  // add/sub only
    if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < (1 << B); i++) x[i] = GLT::mul(x[i], x[(i + 1) & ((1 << B) - 1)]); }              // general multiplies
    if constexpr (MODE == 3) { _Pragma("unroll") for (int i = 0; i < (1 << B); i++) x[i] = A::mul(x[i], x[(i + 1) & ((1 << B) - 1)]); }
  }
  u64 s = 0;
  for (int i = 0; i < (1 << B); i++) s ^= x[i];
  out[tid] = s;
}
static int g_lds = 0;   // dynamic LDS per workgroup: limits the workgroups resident per CU (occupancy experiments)
template <class A, int B, int MODE> void run(const char* name, u64* d_out, u64* d_in, int blocks, double ops_per_iter_per_thread) {
  const int iters = 200;
  if (g_lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(&lab<A, B, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((lab<A, B, MODE>), dim3(blocks), dim3(256), g_lds, 0, d_out, d_in, 2);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((lab<A, B, MODE>), dim3(blocks), dim3(256), g_lds, 0, d_out, d_in, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)blocks * 256 * iters * ops_per_iter_per_thread;
  // lane-cycles per op at 2.4 GHz with all 256 CUs x 4 SIMDs x 32 lanes busy... report lane-ns instead: chip lanes = 256*4*64 wave-lanes per "cycle slot"
  const double lane_cycles = ms * 1e-3 * 2.4e9 * 256 * 4 * 64 / total;
  printf("%-44s blocks %5d  %8.3f ms  %8.1f wave64-lane-cycles per op (2.4 GHz nominal)\n", name, blocks, ms, lane_cycles);
}
template <class A> __global__ void check_k(u64* out, const u64* in, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 a = in[i], b = in[(i * 7 + 3) % n];
  u64* o = out + (size_t)i * 12;
  o[0] = A::add(a, b); o[1] = A::sub(a, b); o[2] = A::mul(a, b);
  o[3] = A::template mul_pow2<3>(a); o[4] = A::template mul_pow2<31>(a); o[5] = A::template mul_pow2<32>(a); o[6] = A::template mul_pow2<45>(a);
  o[7] = A::template mul_pow2<63>(a); o[8] = A::template mul_pow2<64>(a); o[9] = A::template mul_pow2<78>(a); o[10] = A::template mul_pow2<95>(a); o[11] = A::template mul_pow2<12>(b);
}
struct GLr { static MS_HD u64 add(u64 a, u64 b) { return GLT::add(a, b); } static MS_HD u64 sub(u64 a, u64 b) { return GLT::sub(a, b); } static MS_HD u64 mul(u64 a, u64 b) { return GLT::mul(a, b); }
  template <int S> static MS_HD u64 mul_pow2(u64 x) { return msntt::gl_mul_pow2<S>(x); } };
static u64 href_mulmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL::P); }
template <class A> int check(const char* name) {
  const int n = 1 << 16;
  std::vector<u64> h(n);
  u64 s = 0x243F6A8885A308D3ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s % GL::P; }
  const u64 edge[] = {0, 1, 2, GL::P - 1, GL::P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEull, 0x00000001FFFFFFFFull};
  for (int i = 0; i < n; i++) if ((i % 5) == 0) h[i] = edge[(i / 5) % 13];
  for (int i = 0; i < 13 * 13; i++) { h[1000 + 2 * i] = edge[i / 13]; }
  u64 *d_in, *d_out; hipMalloc(&d_in, n * 8); hipMalloc(&d_out, (size_t)n * 12 * 8);
  hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((check_k<A>), dim3(n / 256), dim3(256), 0, 0, d_out, d_in, n);
  std::vector<u64> o((size_t)n * 12);
  hipError_t e = hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  const int sh[9] = {3, 31, 32, 45, 63, 64, 78, 95, 12};
  for (int i = 0; i < n && bad < 10; i++) {
    const u64 a = h[i], b = h[(i * 7 + 3) % n];
    u64 want[12]; want[0] = (u64)(((unsigned __int128)a + b) % GL::P); want[1] = (u64)(((unsigned __int128)a + GL::P - b) % GL::P); want[2] = href_mulmod(a, b);
    for (int k = 0; k < 9; k++) { u64 pw = 1; for (int t = 0; t < sh[k]; t++) pw = href_mulmod(pw, 2); want[3 + k] = href_mulmod(k == 8 ? b : a, pw); }
    for (int k = 0; k < 12; k++) if (o[(size_t)i * 12 + k] != want[k]) { printf("  %s MISMATCH op %d a=%llx b=%llx got %llx want %llx\n", name, k, (unsigned long long)a, (unsigned long long)b, (unsigned long long)o[(size_t)i * 12 + k], (unsigned long long)want[k]); bad++; }
  }
  printf("check %-28s %s (hip status %d)\n", name, bad ? "FAILED" : "ok: add sub mul and 9 shift-multiplies exact on 65536 pairs incl. edge values", (int)e);
  hipFree(d_in); hipFree(d_out);
  return bad;
}
__global__ void check_pair_k(u64* out, const u64* in) {
  u64 x[32], y[32];
  for (int i = 0; i < 32; i++) x[i] = y[i] = in[(threadIdx.x * 32 + i) & 65535];
  Stage<GL1, 5, 4, 0, 0>::run(x);
  Stage2<5, 4, 0, 0>::run(y);
  u64 bad = 0;
  for (int i = 0; i < 32; i++) bad |= (x[i] ^ y[i]);
  out[threadIdx.x] = bad;
}
int main() {
  {
    std::vector<u64> h(65536); u64 s = 12345;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % GL::P; }
    for (int i = 0; i < 65536; i += 7) h[i] = (i % 3) ? GL::P - 1 - (i % 5) : (u64)(i % 4);
    u64 *di, *dout; hipMalloc(&di, 65536 * 8); hipMalloc(&dout, 256 * 8);
    hipMemcpy(di, h.data(), 65536 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_pair_k, dim3(1), dim3(256), 0, 0, dout, di);
    u64 r[256]; hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    u64 bad = 0; for (int i = 0; i < 256; i++) bad |= r[i];
    printf("check paired radix-32 network == sign-bit network: %s\n", bad ? "FAILED" : "ok");
  }

  check<GLr>("r02 sign-bit GLT (field.hpp)");
  check<GL3>("masked asm");
  check<GL5>("masked asm, borrow from subb (r03)");
  check<GL6>("masked asm, full-exec form (r03)");

  u64 *d_in, *d_out;
  std::vector<u64> h(4096);
  u64 s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % GL::P; }
  hipMalloc(&d_in, 4096 * 8); hipMalloc(&d_out, 8192 * 256 * 8);
  hipMemcpy(d_in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  for (int lds : {0, 40 << 10, 72 << 10, 140 << 10}) {   // 8 (register-limited), 4, 2, 1 workgroups of 4 waves per CU
    const int blocks = 4096; g_lds = lds;
    printf("==== dynamic LDS %d KiB per workgroup\n", lds >> 10);
    run<GL0, 5, 0>("r01 radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL1, 5, 0>("r02 radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL3, 5, 0>("masked-asm radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL3, 4, 0>("masked-asm radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL5, 5, 0>("r03 sub3 radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL5, 4, 0>("r03 sub3 radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL5, 3, 0>("r03 sub3 radix-8 DIF", d_out, d_in, blocks, 8 * 3);
    run<GL6, 5, 0>("r03 full-exec radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL6, 4, 0>("r03 full-exec radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL6, 3, 0>("r03 full-exec radix-8 DIF", d_out, d_in, blocks, 8 * 3);
    run<GL3, 3, 0>("masked-asm radix-8 DIF", d_out, d_in, blocks, 8 * 3);
    run<GL4, 5, 4>("paired masked-asm radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL4, 4, 4>("paired masked-asm radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL3, 4, 3>("masked-asm general multiply (16 values)", d_out, d_in, blocks, 16);
  }
  g_lds = 0;
  for (int blocks : {4096}) {   // 1, 2, 4, 8 workgroups (4, 8, 16, 32 waves) per CU if registers allow
    printf("---- %d workgroups of 256 threads\n", blocks);
    // op = one element through one butterfly stage
    run<GL0, 4, 1>("r01 add/sub only (16 values)", d_out, d_in, blocks, 16);
    run<GL1, 4, 1>("r02 add/sub only (16 values)", d_out, d_in, blocks, 16);
    run<GL2, 4, 1>("lazy add/sub only (16 values, inexact)", d_out, d_in, blocks, 16);
    run<GL0, 4, 0>("r01 radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL1, 4, 0>("r02 radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL2, 4, 0>("lazy radix-16 DIF (inexact)", d_out, d_in, blocks, 16 * 4);
    run<GL3, 4, 0>("masked-asm radix-16 DIF", d_out, d_in, blocks, 16 * 4);
    run<GL3, 5, 0>("masked-asm radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL0, 5, 0>("r01 radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL1, 5, 0>("r02 radix-32 DIF", d_out, d_in, blocks, 32 * 5);
    run<GL1, 4, 2>("general multiply (16 values)", d_out, d_in, blocks, 16);
    run<GL3, 4, 3>("masked-asm general multiply (16 values)", d_out, d_in, blocks, 16);
  }
  return 0;
}
