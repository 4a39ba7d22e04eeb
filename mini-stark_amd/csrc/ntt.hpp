// ntt.hpp — radix-2 NTT / INTT / zero-padded coset evaluation over the base
// field, natural order in and out, as 1–4 HBM passes of LDS-resident tiles.
//
// Replaces ark-poly's Radix2EvaluationDomain::{ifft, fft} at the reference call
// sites  src/air.rs:154 (INTT of trace columns), src/starks.rs:89 (coset LDE)
// and src/fri.rs:350 (FRI codeword; extension limbs are independent base
// transforms because the domain points are base-field elements).
//
// Decomposition (self-sorting, no bit-reversal pass).  n = r_0 r_1 r_2 ... r_P.
// After pass p the array holds A_p[k_rest * R_p + i_done]: `i_done` = the low
// output digits already produced (R_p = r_0..r_p values), `k_rest` = the input
// index digits not yet transformed.  Pass p+1 transforms the TOP digit of
// k_rest (stride n / r) with an r-point DFT, multiplies by the inter-pass
// twiddle and stores to  A_{p+1}[k_low * R_p * r + i_done + R_p * i_new].
// The twiddle of an element is  w_n^(k_low * (its output index so far)).
//
// Zero padding (blowup): when only n_in <= n / r_0 inputs are non-zero
// (r_0 = 2, 4 or 8) the first pass is VIRTUAL — an r_0-point DFT of
// (x, 0, .., 0) is r_0 copies of x — and is folded into the load of the first
// real pass:  A_1[k*r_0 + i_1] = x[k] * w_n^(i_1 k).  A blowup-8 LDE of 2^20
// coefficients is then two real passes (2^10 x 2^10), not three.
//
// A workgroup owns a tile of r rows x C consecutive "columns" f = k_low*R_p +
// i_done, so global accesses are runs of C consecutive elements (C = 16: 128 B
// for Goldilocks) or, when R_p < C, whole contiguous blocks of R_p * r elements
// (transposed store).  Algorithmic HBM traffic: 2 * n * sizeof(T) per real pass.
//
// Inside a tile the r-point DFT runs as register sub-rounds of 2^b points
// (b <= 4) with LDS exchanges between them: DIF, digits taken from the top,
// results left digit-reversed in LDS and un-reversed by the store phase.
// Goldilocks: every root of unity of order <= 64 is a power of two
// (w_64 = 2^39 for the reference's generator), so the twiddles INSIDE a
// sub-round are shifts + one reduction, not multiplications.
// Tile rows are padded by one element (bank spreading).
#pragma once
#include "field.hpp"

namespace msntt {

constexpr int TILE_LOG_C = 4;       // 16 columns per tile
constexpr int MAX_LOG_R = 10;       // tile rows <= 1024
constexpr int MAX_LOG_PAD = 3;      // zero-padding factor folded into the virtual first pass <= 8
constexpr int MAX_LOG_RHO = 2;      // ... times a real radix <= 4 over the non-zero input blocks

// sub-round digit sizes for a tile of 2^K rows (top digit first)
MS_HD constexpr int subround_count(int K) { return K <= 4 ? 1 : (K <= 8 ? 2 : 3); }
MS_HD constexpr int subround_bits(int K, int s) {
  // K<=4: {K}; 5:{3,2} 6:{3,3} 7:{4,3} 8:{4,4}; 9:{3,3,3} 10:{4,3,3}
  return K <= 4 ? K : (K <= 8 ? (s == 0 ? (K + 1) / 2 : K - (K + 1) / 2) : (K == 9 ? 3 : (s == 0 ? 4 : 3)));
}
MS_HD constexpr int subround_slo(int K, int s) {  // lowest row bit of sub-round s
  int done = 0;
  for (int t = 0; t <= s; t++) done += subround_bits(K, t);
  return K - done;
}

template <class F> struct PassParams {
  typedef typename F::T T;
  const T* src; T* dst;
  size_t src_bstride, dst_bstride;  // elements between consecutive batch entries (blockIdx.y)
  size_t n_in;                      // valid input elements (zero padded); first real pass only
  // every table holds F::to_tw(w) (Goldilocks: w itself; BabyBear: Montgomery form, one reduction per twiddle multiply)
  const T* tw_lo; const T* tw_hi;   // w_n^j = tw_lo[j & lo_mask] * tw_hi[j >> lo_bits]
  const T* w_r;                     // w_r^j, j < r
  const T* vtw;                     // virtual pass: w_(r0*r)^j, j < r0*r
  const T* w0;                      // virtual pass: w_(r0)^j, j < r0
  T scale; u32 do_scale;            // table form of the factor multiplied into the output of the last pass (do_scale = 0: none)
  u32 log_n, log_r, log_Rp, log_C, lo_bits;
  u32 log_r0 /* >0: this pass loads through a virtual r0-point pass */, log_rho /* of whose inputs 2^log_rho blocks are non-zero */;
  u32 last;
  u32 nbatch;                       // batch entries of the launch (cooperative kernels walk batch x tiles themselves)
};

// ---- Goldilocks shift twiddles ------------------------------------------------
// x * 2^S mod p for a compile-time S in [0, 96)
// round-1 formulation (compare + select reduce128), used by the round-1 tiles (arithmetic class GL)
template <int S> MS_HD u64 gl_mul_pow2_v1(u64 x) {
  static_assert(S >= 0 && S < 96, "shift out of range");
  if constexpr (S == 0) return x;
  else if constexpr (S < 64) return GL::reduce128(x << S, x >> (64 - S));
  else {  // x*2^S == x*2^(S-32) - x*2^(S-64)  (2^64 == 2^32 - 1)
    constexpr int A = S - 32, B = S - 64;  // 32 <= A < 64, 0 <= B < 32
    const u64 alo = x << A, ahi = x >> (64 - A);
    const u64 blo = x << B, bhi = (B == 0) ? 0 : (x >> ((64 - B) & 63));
    const u64 lo = alo - blo;
    const u64 hi = ahi - bhi - (alo < blo ? 1 : 0);
    return GL::reduce128(lo, hi);
  }
}
// throughput formulation (arithmetic class GLT): every step canonical and branch-free on sign bits
template <int S, class A = GLT> MS_HD u64 gl_mul_pow2(u64 x) {
  static_assert(S >= 0 && S < 96, "shift out of range");
  if constexpr (S == 0) return x;
  else if constexpr (S < 32) return A::fold_small(ms_pin64(x << S), GL::hi(x) >> (32 - S));  // the S bits shifted out times 2^64 == EPS (pinned: one v_lshlrev_b64, its high word reused)
  else if constexpr (S < 64) return A::mul_x32(gl_mul_pow2<S - 32, A>(x));                    // two steps: both stay canonical
  else return A::mul_x64(gl_mul_pow2<S - 64, A>(x));
}
// (a - b) * w_(2^LOG2H2)^J for the reference's roots: w_64 = 2^39 (forward), 2^153 (inverse)
template <class F, bool INV, int LOG2H2, int J> struct TwMul;
template <bool INV, int LOG2H2, int J> struct TwMul<GL, INV, LOG2H2, J> {
  static constexpr int EXP = ((INV ? 153 : 39) * (64 >> LOG2H2) * J) % 192;
  static MS_HD u64 diff_mul(u64 a, u64 b, const u64* w_r, int log_r) {
    (void)w_r; (void)log_r;
    if constexpr (EXP >= 96) return gl_mul_pow2_v1<EXP - 96>(GL::sub(b, a));  // 2^96 == -1
    else return gl_mul_pow2_v1<EXP>(GL::sub(a, b));
  }
};
template <bool INV, int LOG2H2, int J> struct TwMul<GLT, INV, LOG2H2, J> {
  static constexpr int EXP = ((INV ? 153 : 39) * (64 >> LOG2H2) * J) % 192;
  static MS_HD u64 diff_mul(u64 a, u64 b, const u64* w_r, int log_r) {
    (void)w_r; (void)log_r;
    if constexpr (EXP >= 96) return gl_mul_pow2<EXP - 96>(GLT::sub(b, a));
    else return gl_mul_pow2<EXP>(GLT::sub(a, b));
  }
};
template <bool INV, int LOG2H2, int J> struct TwMul<GLM, INV, LOG2H2, J> {
  static constexpr int EXP = ((INV ? 153 : 39) * (64 >> LOG2H2) * J) % 192;
  static MS_HD u64 diff_mul(u64 a, u64 b, const u64* w_r, int log_r) {
    (void)w_r; (void)log_r;
    if constexpr (EXP >= 96) return gl_mul_pow2<EXP - 96, GLM>(GLM::sub(b, a));
    else return gl_mul_pow2<EXP, GLM>(GLM::sub(a, b));
  }
};
template <bool INV, int LOG2H2, int J> struct TwMul<BB, INV, LOG2H2, J> {
  static MS_HD u32 diff_mul(u32 a, u32 b, const u32* w_r, int log_r) { return BB::mul_tw(BB::sub(a, b), w_r[(size_t)J << (log_r - LOG2H2)]); }
};
// arithmetic class of the round-2 tiles: the throughput formulation where the field has one
template <class F> struct NttArith { typedef F type; };
#ifndef MS_NTT_ARITH
#define MS_NTT_ARITH 1   // 0: GL (compare + select)  1: GLT (sign bits)  2: GLM (exec-masked asm; wants >= 4 waves per SIMD)
#endif
#if MS_NTT_ARITH == 1
template <> struct NttArith<GL> { typedef GLT type; };
#elif MS_NTT_ARITH == 2
template <> struct NttArith<GL> { typedef GLM type; };
#endif

template <class F, bool INV, int B, int S, int BLK, int J> struct DifStage {
  // butterflies of stage S (distance 2^S) inside a 2^B-point DIF, unrolled at compile time
  static MS_DEV void run(typename F::T (&x)[1 << B], const typename F::T* w_r, int log_r) {
    constexpr int h = 1 << S;
    typename F::T a = x[BLK + J], b = x[BLK + J + h];
    x[BLK + J] = F::add(a, b);
    if constexpr (J == 0) x[BLK + J + h] = F::sub(a, b);
    else x[BLK + J + h] = TwMul<F, INV, S + 1, J>::diff_mul(a, b, w_r, log_r);
    if constexpr (J + 1 < h) DifStage<F, INV, B, S, BLK, J + 1>::run(x, w_r, log_r);
    else if constexpr (BLK + 2 * h < (1 << B)) DifStage<F, INV, B, S, BLK + 2 * h, 0>::run(x, w_r, log_r);
    else if constexpr (S > 0) DifStage<F, INV, B, S - 1, 0, 0>::run(x, w_r, log_r);
  }
};
// in-register DIF of 2^B points; output left in bit-reversed register order
template <class F, bool INV, int B> MS_DEV void dif_regs(typename F::T (&x)[1 << B], const typename F::T* w_r, int log_r) {
  DifStage<F, INV, B, B - 1, 0, 0>::run(x, w_r, log_r);
}
MS_HD constexpr int bitrev(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }

// LDS row -> output digit index (and back) for the digit-reversed tile
MS_HD constexpr int row_to_inew(int row, int K) {
  int S = subround_count(K), done = 0, inew = 0;
  for (int s = 0; s < S; s++) {
    int b = subround_bits(K, s);
    int e = (row >> (K - done - b)) & ((1 << b) - 1);
    inew |= e << done;
    done += b;
  }
  return inew;
}
MS_HD constexpr int inew_to_row(int inew, int K) {
  int S = subround_count(K), done = 0, row = 0;
  for (int s = 0; s < S; s++) {
    int b = subround_bits(K, s);
    int e = (inew >> done) & ((1 << b) - 1);
    row |= e << (K - done - b);
    done += b;
  }
  return row;
}

template <class F, bool INV, int TH> struct PassKernel {
  typedef typename F::T T;
  typedef PassParams<F> Params;
  static constexpr int THREADS = TH;
#ifdef MS_NTT_MINWAVES
  static constexpr int MIN_WAVES = MS_NTT_MINWAVES;
#endif

  static MS_HD int nphases(const Params& p) { return 2 + subround_count((int)p.log_r); }
  // LDS: tile r*(C+1) (rows padded by one element: bank spreading) | w_r table r
  static MS_HD size_t lds_bytes(int log_r, int log_C, int, bool) {
    const size_t r = (size_t)1 << log_r, C = (size_t)1 << log_C;
    return (r * (C + (C > 1 ? 1 : 0)) + r) * sizeof(T);
  }
  static MS_HD int tix(int row, int cidx, int log_C) { return row * ((1 << log_C) + (log_C ? 1 : 0)) + cidx; }

  template <int B>
  static MS_DEV void subround(const Params& p, int tid, T* tile, const T* w, int s_lo) {
    const int K = (int)p.log_r, C = 1 << p.log_C, lc = (int)p.log_C;
    const int q = 1 << s_lo;                      // row distance between the 2^B points
    const int items = (1 << (K - B)) << p.log_C;  // work items in the tile
    for (int it = tid; it < items; it += TH) {
      const int cidx = it & (C - 1);
      const int g = it >> p.log_C;           // (hi, lo) packed
      const int lo = g & (q - 1), hi = g >> s_lo;
      const int row0 = (hi << (s_lo + B)) + lo;
      T x[1 << B];
#pragma unroll
      for (int t = 0; t < (1 << B); t++) x[t] = tile[tix(row0 + t * q, cidx, lc)];
      dif_regs<F, INV, B>(x, w, K);
      // x[bitrev(e)] = y[e]; twiddle w_{q 2^B}^(e*lo) = w_r[e * lo * r / (q 2^B)]
#pragma unroll
      for (int e = 0; e < (1 << B); e++) {
        T v = x[bitrev(e, B)];
        if (e != 0 && lo != 0) v = F::mul_tw(v, w[((size_t)(e * lo)) << (K - s_lo - B)]);
        tile[tix(row0 + e * q, cidx, lc)] = v;
      }
    }
  }

  static MS_DEV T tw_global(const Params& p, size_t e) {
    T tw = p.tw_lo[e & (((size_t)1 << p.lo_bits) - 1)];
    const size_t eh = e >> p.lo_bits;
    if (eh) tw = F::mul_tw(tw, p.tw_hi[eh]);
    return tw;
  }

  static MS_DEV void phase(int ph, const Params& p, int bx, int by, int tid, int, unsigned char* lds) {
    const int K = (int)p.log_r, r = 1 << K, C = 1 << p.log_C, lc = (int)p.log_C;
    T* tile = reinterpret_cast<T*>(lds);
    T* w = tile + (size_t)r * (C + (C > 1 ? 1 : 0));
    const size_t n = (size_t)1 << p.log_n;
    const size_t col_stride = n >> K;  // n / r : distance between tile rows in the source
    const size_t f0 = (size_t)bx << p.log_C;
    const int S = subround_count(K);
    const bool transposed = (int)p.log_Rp < lc;       // output of each k_low is one contiguous block
    if (ph == 0) {
      const T* src = p.src + (size_t)by * p.src_bstride;
      if (p.log_r0 == 0) {
        for (int idx = tid; idx < r * C; idx += TH) {
          const int row = idx >> p.log_C, cidx = idx & (C - 1);
          const size_t a = f0 + cidx + (size_t)row * col_stride;
          tile[tix(row, cidx, lc)] = (a < p.n_in) ? src[a] : (T)0;
        }
      } else {
        // virtual r0-point first pass over x[k2 + (n/r0) k1], of which only blocks k1 < rho are non-zero:
        //   A_1[k2*r0 + i1] = w_n^(i1 k2) * sum_{k1 < rho} w_r0^(i1 k1) x[k2 + (n/r0) k1].
        // The k_low part of w_n^(i1 k2) is merged into the store twiddle, the k_top part is w_(r0 r)^(i1 k_top).
        const int r0m = (1 << p.log_r0) - 1, rho = 1 << p.log_rho;
        const size_t nprime = n >> (p.log_r0 + K), blk = n >> p.log_r0;
        for (int idx = tid; idx < r * C; idx += TH) {
          const int row = idx >> p.log_C, cidx = idx & (C - 1);
          const size_t f = f0 + cidx;
          const int i1 = (int)(f & r0m);
          const size_t k = (f >> p.log_r0) + nprime * (size_t)row;
          T v = (k < p.n_in) ? src[k] : (T)0;
          for (int k1 = 1; k1 < rho; k1++) {
            const size_t kk = k + blk * (size_t)k1;
            T x = (kk < p.n_in) ? src[kk] : (T)0;
            const int e = (i1 * k1) & r0m;
            if (e) x = F::mul_tw(x, p.w0[e]);
            v = F::add(v, x);
          }
          if (i1) v = F::mul_tw(v, p.vtw[(size_t)i1 * row]);
          tile[tix(row, cidx, lc)] = v;
        }
      }
      for (int j = tid; j < r; j += TH) w[j] = p.w_r[j];
      return;
    }
    if (ph <= S) {
      const int s = ph - 1;
      int done = 0;
      for (int t = 0; t < s; t++) done += subround_bits(K, t);
      const int b = subround_bits(K, s);
      const int s_lo = K - done - b;
      switch (b) {
        case 1: subround<1>(p, tid, tile, w, s_lo); break;
        case 2: subround<2>(p, tid, tile, w, s_lo); break;
        case 3: subround<3>(p, tid, tile, w, s_lo); break;
        default: subround<4>(p, tid, tile, w, s_lo); break;
      }
      return;
    }
    // store phase
    T* dst = p.dst + (size_t)by * p.dst_bstride;
    const bool do_scale = p.do_scale != 0;
    const int Rp_m = (1 << p.log_Rp) - 1;
    for (int idx = tid; idx < r * C; idx += TH) {
      int row, cidx, inew, kk, i_done;
      if (transposed) {  // (i_done, i_new) fastest: each k_low writes Rp*r contiguous elements
        kk = idx >> (K + p.log_Rp);
        const int j = idx & ((r << p.log_Rp) - 1);
        i_done = j & Rp_m; inew = j >> p.log_Rp;
        cidx = (kk << p.log_Rp) + i_done; row = inew_to_row(inew, K);
      } else {
        row = idx >> p.log_C; cidx = idx & (C - 1); inew = row_to_inew(row, K);
        kk = 0; i_done = (int)((f0 + cidx) & Rp_m);
      }
      T v = tile[tix(row, cidx, lc)];
      const size_t k_low = (f0 + cidx) >> p.log_Rp;
      if (!p.last && k_low) {  // w_n^(k_low * (output index so far)); the i_done term belongs to the virtual pass
        const size_t e = k_low * (((size_t)inew << p.log_Rp) + (p.log_r0 ? (size_t)i_done : 0));
        if (e) v = F::mul_tw(v, tw_global(p, e));
      }
      if (do_scale) v = F::mul_tw(v, p.scale);
      const size_t out = ((k_low << p.log_Rp) << K) + ((size_t)(f0 + cidx) & (size_t)Rp_m) + ((size_t)inew << p.log_Rp);
      dst[out] = v;
    }
  }
};


// ---------------------------------------------------------------------------------------------
// Compile-time specialised pass: K (log2 tile rows), 16 columns, TH threads, no virtual pass.
// Same arithmetic and data movement as PassKernel, but every tile index, LDS address and output
// offset is "per-thread base + compile-time constant": the pass is integer-VALU-issue bound on
// MI355X (~4.4 cycles per wave-instruction), so the index arithmetic of the generic kernel
// (~40% of its instructions) is what this removes.  Used for the large transforms.
template <class F, bool INV, int K, int TH> struct PassKernelK {
  typedef typename F::T T;
  typedef PassParams<F> Params;
  static constexpr int THREADS = TH;
  static constexpr int C = 16, LC = 4, R = 1 << K, CP = C + 1;
  static constexpr int RPT = TH / C;          // tile rows covered by one sweep of the workgroup
  static constexpr int LRPT = (TH == 256) ? 4 : 5;
  static constexpr int NIT = R / RPT;         // sweeps over the tile in the load / store phases
  static constexpr int S = subround_count(K);
  static_assert(K >= 5 && K <= 9 && (TH == 256 || TH == 512) && R >= RPT, "unsupported tile");

  static MS_HD int nphases(const Params&) { return 2 + S; }
  static MS_HD size_t lds_bytes() { return ((size_t)R * CP + 2 * R) * sizeof(T); }
  static MS_HD bool applicable(const Params& p) {
    return p.log_r == K && p.log_C == LC && p.log_r0 == 0 && (p.log_Rp == 0 || p.log_Rp >= LC);
  }

  static MS_DEV T tw_global(const Params& p, size_t e) {
    T tw = p.tw_lo[e & (((size_t)1 << p.lo_bits) - 1)];
    const size_t eh = e >> p.lo_bits;
    if (eh) tw = F::mul_tw(tw, p.tw_hi[eh]);
    return tw;
  }
  // row bits of a sub-round work item g: B zero bits inserted at position SLO
  template <int SLO, int B> static MS_HD constexpr int place(int g) { return ((g >> SLO) << (SLO + B)) | (g & ((1 << SLO) - 1)); }

  template <int SR, int J> static MS_DEV void subround_items(int tid, T* tile, const T* w) {
    constexpr int B = subround_bits(K, SR), SLO = subround_slo(K, SR), Q = 1 << SLO;
    constexpr int GROUPS = R >> B;                 // work items per column
    constexpr int NJ = (GROUPS + RPT - 1) / RPT;   // items per thread
    constexpr int SH = K - SLO - B;                // twiddle index shift
    const int cidx = tid & (C - 1), gb = tid >> LC;
    if (GROUPS >= RPT || gb < GROUPS) {
      constexpr int GJ = J * RPT;                  // compile-time part of the item index (disjoint bits from gb)
      constexpr int ROW_J = place<SLO, B>(GJ), LO_J = GJ & (Q - 1);
      T* base = tile + place<SLO, B>(gb) * CP + cidx;
      T x[1 << B];
#pragma unroll
      for (int t = 0; t < (1 << B); t++) x[t] = base[(ROW_J + t * Q) * CP];
      dif_regs<F, INV, B>(x, w, K);
      if constexpr (SLO == 0) {
#pragma unroll
        for (int e = 0; e < (1 << B); e++) base[(ROW_J + e * Q) * CP] = x[bitrev(e, B)];
      } else {
        const int a = (gb & (Q - 1)) << SH;        // runtime part of lo << SH
#pragma unroll
        for (int e = 0; e < (1 << B); e++) {
          T v = x[bitrev(e, B)];
          if (e != 0) v = F::mul_tw(v, w[e * a + ((e * LO_J) << SH)]);  // w_{Q 2^B}^(e*lo); lo == 0 multiplies by w[0] = 1
          base[(ROW_J + e * Q) * CP] = v;
        }
      }
    }
    if constexpr (J + 1 < NJ) subround_items<SR, J + 1>(tid, tile, w);
  }

  template <int IT> static MS_DEV void load_rows(const T* src, size_t cs, T* trow) {
    trow[IT * RPT * CP] = src[(size_t)(IT * RPT) * cs];
    if constexpr (IT + 1 < NIT) load_rows<IT + 1>(src, cs, trow);
  }
  template <int IT> static MS_DEV void store_rows(const Params& p, const T* trow, T* out, const T* twr, bool tw, bool do_scale) {
    T v = trow[IT * RPT * CP];
    if (tw) v = F::mul_tw(v, twr[IT * RPT]);   // one twiddle per tile ROW, shared by its 16 columns
    if (do_scale) v = F::mul_tw(v, p.scale);
    out[(size_t)row_to_inew(IT * RPT, K) << p.log_Rp] = v;
    if constexpr (IT + 1 < NIT) store_rows<IT + 1>(p, trow, out, twr, tw, do_scale);
  }

  static MS_DEV void phase(int ph, const Params& p, int bx, int by, int tid, int, unsigned char* lds) {
    T* tile = reinterpret_cast<T*>(lds);
    T* w = tile + (size_t)R * CP;
    T* twr = w + R;                                // [R] store twiddle of every tile row: w_n^(k_low * Rp * i_new(row))
    const size_t n = (size_t)1 << p.log_n, cs = n >> K, f0 = (size_t)bx << LC;
    const int cidx = tid & (C - 1), rb = tid >> LC;
    if (ph == 0) {
      const T* src = p.src + (size_t)by * p.src_bstride + f0 + cidx + (size_t)rb * cs;
      T* trow = tile + rb * CP + cidx;
      if (p.n_in >= n) load_rows<0>(src, cs, trow);
      else {
        for (int it = 0; it < NIT; it++) {
          const size_t a = f0 + cidx + (size_t)(rb + it * RPT) * cs;
          trow[it * RPT * CP] = (a < p.n_in) ? src[(size_t)(it * RPT) * cs] : (T)0;
        }
      }
      for (int j = tid; j < R; j += TH) w[j] = p.w_r[j];
      if (!p.last && p.log_Rp) {
        const size_t k_low = f0 >> p.log_Rp;
        if (k_low) for (int row = tid; row < R; row += TH) twr[row] = tw_global(p, ((size_t)row_to_inew(row, K) * k_low) << p.log_Rp);
      }
      return;
    }
    if (ph <= S) {
      if (ph == 1) subround_items<0, 0>(tid, tile, w);
      if constexpr (S >= 2) { if (ph == 2) subround_items<1, 0>(tid, tile, w); }
      if constexpr (S >= 3) { if (ph == 3) subround_items<2, 0>(tid, tile, w); }
      return;
    }
    T* dst = p.dst + (size_t)by * p.dst_bstride;
    const bool do_scale = p.do_scale != 0;
    if (p.log_Rp) {
      // columns stay columns: out = k_low*Rp*r + i_done + Rp*i_new, C-element runs
      const size_t f = f0 + cidx, k_low = f0 >> p.log_Rp, i_done = f & (((size_t)1 << p.log_Rp) - 1);
      const int inew_rb = row_to_inew(rb, K);
      const bool tw = !p.last && k_low != 0;
      T* out = dst + ((k_low << p.log_Rp) << K) + i_done + ((size_t)inew_rb << p.log_Rp);
      store_rows<0>(p, tile + rb * CP + cidx, out, twr + rb, tw, do_scale);
    } else {
      // first pass: out = f*r + i_new, i_new fastest across lanes (transposed)
      for (int idx = tid; idx < R * C; idx += TH) {
        const int inew = idx & (R - 1), c2 = idx >> K;
        const size_t f = f0 + c2;
        T v = tile[inew_to_row(inew, K) * CP + c2];
        if (!p.last) { const size_t e = (size_t)inew * f; if (e) v = F::mul_tw(v, tw_global(p, e)); }
        if (do_scale) v = F::mul_tw(v, p.scale);
        dst[(f << K) + inew] = v;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// PassKernel2: the large-transform pass.  A tile of 2^K rows x 2^LC columns (GL: 2^10 x 8 = 64 KiB, two workgroups per CU) is
// transformed by NSUB register sub-rounds, so that a 2^20-point transform (and, behind the virtual radix-8 zero-padding pass, the
// 2^23-point LDE of 2^20 coefficients) is TWO HBM passes of ten butterfly stages instead of three of eight.
// These kernels are bound by the NUMBER of VALU instructions they issue (3.3-3.6 issue cycles each whatever the opcode:
// tools/ntt_lab.hip) and by how many waves hide each other's stalls:
//   NSUB = 2, 256 threads: radix-32 x radix-32 per thread, one twiddle multiplication inside the tile, compiler-scheduled arithmetic
//             (tolerates the 2 waves per SIMD a 64 KiB tile leaves)
//   NSUB = 3, 512 threads, arithmetic class GLM: radix-16 x 8 x 8, 16 waves per CU - the occupancy the exec-masked arithmetic
//             (7.8 instead of 11.7 VALU instructions per element-stage) needs; one more twiddle multiplication and LDS round trip.
// Global access is 16 bytes per lane (global_load/store_dwordx4: VEC = 2 Goldilocks / 4 BabyBear elements), a tile row is one
// 16*LPR-byte run.  LDS: element (row, c) at prow(row)*C + c with prow(row) = row ^ ((row >> BL) & 3), BL = bits of the last digit:
// the XOR spreads the four row groups a half-wave touches in the last sub-round over all banks (no padding, 16-byte alignment kept),
// and permutes rows only inside aligned groups of four, which the other access patterns (lanes over consecutive rows x columns)
// do not notice.
// MODE (compile time: each instance carries ONE load path and ONE store path; same index algebra as PassKernel above):
//   0  log_r0 == 0, log_Rp == 0   first pass of a plain transform: rows in, transposed store, twiddle w_n^(i_new * f)
//   1  log_r0 == 0, log_Rp >= LC  later pass: rows in, 16*LPR-byte output runs, one twiddle per tile row (none in the last pass)
//   2  log_r0 == LC               first pass behind the virtual zero-padding pass: the C columns of a tile are the r0 cosets of ONE
//                                 coefficient index; coefficients staged through LDS one tile ahead, the tile's output is one
//                                 contiguous block of r0 * r elements
// What was tried on these tiles and lost (cross-tile register prefetch, LDS-DMA prefetch with a counted vmcnt, the expanded
// coefficients kept in registers for the first sub-round, 16-column 1024-thread tiles, the timing-only ablation branches) lives in
// the round-2 history of this file and in profiles/HISTORY.md, not in the product kernel.
template <int K, int NSUB> struct Digits2 {
  // digit sizes, top digit first: NSUB == 2: (ceil(K/2), floor(K/2)); NSUB == 3: last = K/3, middle = (K - last)/2, top = the rest
  // (10: 4,3,3   9: 3,3,3   8: 3,3,2   7: 3,2,2)
  static constexpr int bits(int s) { return NSUB == 2 ? (s == 0 ? (K + 1) / 2 : K / 2) : (s == 2 ? K / 3 : (s == 1 ? (K - K / 3) / 2 : K - K / 3 - (K - K / 3) / 2)); }
  static constexpr int slo(int s) { int d = 0; for (int t = 0; t <= s; t++) d += bits(t); return K - d; }
};
template <class F, class A, bool INV, int K, int LC, int TH, int NSUB, int MODE> struct PassKernel2 {
  typedef typename F::T T;
  typedef PassParams<F> Params;
  typedef Digits2<K, NSUB> DG;
  typedef A Arith;
  // MODE 3 = MODE 2 (first pass behind the virtual pass) with the per-tile twiddle tables SHARED by the batch entries of a tile: a separate instance, because
  // the extra loop level costs the plain one registers (r03: 120 VGPRs without, 128 + 22 spilled dwords with the choice made at run time)
  static constexpr int PM = (MODE == 3) ? 2 : MODE;
  static constexpr bool SHARE = MODE == 3;
  static constexpr bool INVERSE = INV;
  static constexpr int THREADS = TH;
  static constexpr int R = 1 << K, C = 1 << LC;
  static constexpr int BL = DG::bits(NSUB - 1);
  static constexpr int VEC = 16 / (int)sizeof(T), LPR = C / VEC, RPS = TH / LPR, SWEEPS = R / RPS;  // lanes per row, rows per sweep
  // waves per SIMD the register allocator must leave room for: 16 waves per CU = 2 x 512 threads on 64 KiB tiles, or 4 x 256 threads on BabyBear's 32 KiB tiles behind the virtual pass
  static constexpr int MIN_WAVES = (TH >= 512 || (sizeof(T) == 4 && K == 10 && LC == 3 && NSUB == 3)) ? 4 : 2;
  static_assert(K >= 7 && K <= 10 && DG::bits(0) <= 5 && BL >= 2 && C % VEC == 0 && TH % LPR == 0 && R % RPS == 0 && (NSUB == 2 || NSUB == 3), "unsupported tile");
  static_assert(NSUB == 2 || DG::bits(1) >= 2, "middle digit");
  static_assert(PM >= 0 && PM <= 2, "PM");
  typedef T V16 __attribute__((vector_size(16)));   // one 16-byte global / LDS access

  // tile | w_r [R] | store twiddle of every tile row [R] (filled in the load phase, while the tile's global loads are in flight)
  static MS_HD size_t lds_bytes() { return ((size_t)R * C + 2 * R) * sizeof(T); }
  // r03, SHIFT1: the SECOND sub-round boundary of a three-sub-round Goldilocks tile multiplies by w_(2^(B1+BL))^(e * lo), a root of order <= 64, i.e.
  // a power of two (w_64 = 2^39 for the reference's generator) - so if every lane of a wave works on the same `lo` the multiplication is a SHIFT by a
  // wave-uniform amount (one of 2^BL compile-time variants picked by a scalar branch): 6.7 VALU instructions per element instead of 19 * 7/8.  The
  // sub-round's items are therefore mapped wave = lo, lane = (top digit, column); the lanes of a wave then read rows 2^(B1+BL) apart, which the second
  // XOR term of the row swizzle (the low two bits of the top digit) spreads over the banks again (two lanes per bank: the minimum for 512 bytes).
#ifndef MS_NTT_SHIFT1
#define MS_NTT_SHIFT1 1   // 1: plain first pass and later passes (measured r03: later pass 83 -> 79 VALU instructions per element, 239 -> 234 us per six-column launch);
                          // 2: also behind the virtual pass (126.7 -> 124.9 instructions, but 296 -> 303 us: bank conflicts of the bigger store table + scalar work); 0: off
#endif
  static constexpr bool SHIFT1 = MS_NTT_SHIFT1 != 0 && (PM != 2 || MS_NTT_SHIFT1 == 2) && F::ID == 0 && NSUB == 3 && (TH / 64) == (1 << DG::slo(1)) && ((C << DG::bits(0)) % 64) == 0 && (64 % C) == 0 &&
                                 DG::bits(0) >= 2 && DG::bits(1) >= 2 && DG::bits(1) + BL <= 6 &&
                                 (PM != 2 || ((C << BL) << DG::bits(1)) <= R / 2);   // behind the virtual pass the store table [E1][E2][c] must fit half the row-twiddle region
  static MS_HD int prow(int row) { return row ^ ((row >> BL) & 3) ^ (SHIFT1 ? ((row >> DG::slo(0)) & 3) : 0); }
  static MS_HD int tix(int row, int c) { return prow(row) * C + c; }
  static MS_HD int mode_of(const Params& p) { return p.log_r0 ? 2 : (p.log_Rp == 0 ? 0 : 1); }
  static MS_HD bool applicable(const Params& p) {
    return p.log_r == K && p.log_C == LC && p.log_rho == 0 && mode_of(p) == PM && (!SHARE || p.nbatch > 1) &&
           ((p.log_r0 == 0 && (p.log_Rp == 0 || p.log_Rp >= LC)) || (p.log_r0 == LC && p.log_Rp == LC));
  }
  static MS_DEV T tw_global(const Params& p, size_t e) {
    T tw = p.tw_lo[e & (((size_t)1 << p.lo_bits) - 1)];
    const size_t eh = e >> p.lo_bits;
    if (eh) tw = A::mul_tw(tw, p.tw_hi[eh]);
    return tw;
  }
  // output digit index <-> tile row (digit-reversed): the digit taken from the top of the row index is the LOWEST output digit
  static MS_HD int row_to_inew(int row) {
    int inew = 0, done = 0;
    for (int s = 0; s < NSUB; s++) { const int b = DG::bits(s); inew |= ((row >> (K - done - b)) & ((1 << b) - 1)) << done; done += b; }
    return inew;
  }
  static MS_HD int inew_to_row(int inew) {
    int row = 0, done = 0;
    for (int s = 0; s < NSUB; s++) { const int b = DG::bits(s); row |= ((inew >> done) & ((1 << b) - 1)) << (K - done - b); done += b; }
    return row;
  }

  // one sub-round: digit of B bits at row bits [SLO, SLO + B); item g = the other row bits (hi:lo packed), c = column
  // MG: `w` is this boundary's merged per-tile table [2^B][Q] (sub-round twiddle x the row-twiddle factor of this output digit) instead of w_r
  template <int SR, int J, bool MG = false> static MS_DEV void sub_items(int tid, T* tile, const T* w) {
    constexpr int B = DG::bits(SR), SLO = DG::slo(SR), Q = 1 << SLO;
    constexpr int ITEMS = (R >> B) * C, NJ = (ITEMS + TH - 1) / TH;
    constexpr bool LAST = (SR == NSUB - 1);
    static_assert(SLO >= BL + 2 || SLO == BL || SLO == 0, "digit position vs the row swizzle");
    const int it = tid + J * TH;
    if (ITEMS % TH == 0 || it < ITEMS) {
      const int c = it & (C - 1), g = it >> LC;
      const int lo = g & (Q - 1), R0 = ((g >> SLO) << (SLO + B)) | lo;     // row of t = 0
      // physical row of element t = R0 + t*Q under the swizzle: up to four bases selected by the compile-time (t & 3)
      T* base[4];
      static_assert(!SHIFT1 || SR != 1 || MG, "SHIFT1 tiles run their second sub-round through sub1_shift");
      if constexpr (SLO >= BL + 2 && SHIFT1) { for (int k = 0; k < 4; k++) base[k] = tile + ((R0 ^ ((R0 >> BL) & 3) ^ k) * C + c); }   // top digit: its low two bits (t & 3) are the second swizzle term
      else if constexpr (SLO >= BL + 2) { base[0] = tile + ((R0 ^ ((R0 >> BL) & 3)) * C + c); base[1] = base[2] = base[3] = base[0]; }
      else if constexpr (SLO == 0) { const int h3 = ((R0 >> BL) & 3) ^ (SHIFT1 ? ((R0 >> DG::slo(0)) & 3) : 0); for (int k = 0; k < 4; k++) base[k] = tile + ((R0 + (k ^ h3)) * C + c); }
      else { for (int k = 0; k < 4; k++) base[k] = tile + ((R0 ^ k ^ (SHIFT1 ? ((R0 >> DG::slo(0)) & 3) : 0)) * C + c); }                // SLO == BL: the swizzle term is t & 3
      T x[1 << B];
#pragma unroll
      for (int t = 0; t < (1 << B); t++) x[t] = base[t & 3][(SLO == 0 ? (t & ~3) : t * Q) * C];
      dif_regs<A, INV, B>(x, w, K);
#pragma unroll
      for (int e = 0; e < (1 << B); e++) {
        T v = x[bitrev(e, B)];
        if constexpr (!LAST) { if (e != 0) v = A::mul_tw(v, MG ? w[(e << SLO) + lo] : w[(e * lo) << (K - SLO - B)]); }   // w_{Q 2^B}^(e * lo); lo == 0 multiplies by w[0] = 1
        base[e & 3][(SLO == 0 ? (e & ~3) : e * Q) * C] = v;
      }
    }
    if constexpr (J + 1 < NJ) sub_items<SR, J + 1, MG>(tid, tile, w);
  }

  // x * w_(2^(B1+BL))^(E * LO) for compile-time E, LO: a shift (2^96 == -1: exponents >= 96 negate)
  template <int E, int LO> static MS_DEV T shift_tw(T v) {
    constexpr int EXP = ((INV ? 153 : 39) * (64 >> (DG::bits(1) + BL)) * E * LO) % 192;
    if constexpr (EXP == 0) return v;
    else if constexpr (EXP >= 96) return A::sub((T)0, gl_mul_pow2<EXP - 96, A>(v));
    else return gl_mul_pow2<EXP, A>(v);
  }
  template <int LO, int E = 1> static MS_DEV void shift_all(T (&x)[1 << DG::bits(1)]) {
    constexpr int B = DG::bits(1);
    x[bitrev(E, B)] = shift_tw<E, LO>(x[bitrev(E, B)]);
    if constexpr (E + 1 < (1 << B)) shift_all<LO, E + 1>(x);
  }
  template <int LO = 0> static MS_DEV void shift_dispatch(int lo, T (&x)[1 << DG::bits(1)]) {   // `lo` is wave-uniform: scalar branches
    if (lo == LO) { if constexpr (LO != 0) shift_all<LO>(x); return; }
    if constexpr (LO + 1 < (1 << BL)) shift_dispatch<LO + 1>(lo, x);
  }
  // SHIFT1: the second sub-round with wave = lo (row bits [0, BL)), lane = (top digit hi, column c)
  static MS_DEV void sub1_shift(int tid, T* tile) {
    constexpr int B = DG::bits(1), Q = 1 << BL, SLO0 = DG::slo(0), HPJ = 64 >> LC, NJ = (1 << DG::bits(0)) / HPJ;
    const int lo = msrt::wave_uniform(tid >> 6), lane = tid & 63, c = lane & (C - 1);
#pragma unroll
    for (int J = 0; J < NJ; J++) {
      const int hi = (lane >> LC) + J * HPJ, R0 = (hi << SLO0) | lo;
      T* base[4];
      for (int k = 0; k < 4; k++) base[k] = tile + ((R0 ^ k ^ (hi & 3)) * C + c);   // element t at row R0 + t * Q: swizzle term (t & 3) ^ (hi & 3)
      T x[1 << B];
#pragma unroll
      for (int t = 0; t < (1 << B); t++) x[t] = base[t & 3][t * Q * C];
      dif_regs<A, INV, B>(x, (const T*)nullptr, K);    // Goldilocks butterflies need no table
      shift_dispatch(lo, x);
#pragma unroll
      for (int e = 0; e < (1 << B); e++) base[e & 3][e * Q * C] = x[bitrev(e, B)];
    }
  }
  // ---- cooperative, persistent form: a workgroup walks tiles g = bx, bx + nbx, ... of the nbatch * tiles of the launch.
  // Fused tail: the last sub-round's lanes take VEC neighbouring columns of their 2^BL rows, so that what they hold after the
  // butterflies is exactly the 16-byte pieces of the store - no write-back of the last sub-round, no read-back for the store, one
  // barrier less per tile.
  static constexpr int TAIL_ITEMS = (R >> BL) * LPR, NJT = (TAIL_ITEMS + TH - 1) / TH;
  static constexpr bool FUSE_TAIL = PM != 0 && NSUB == 3 && TAIL_ITEMS % TH == 0;
  // Behind the virtual pass: the tile's R coefficients (one gather each) are fetched ONE TILE AHEAD into NS registers per thread and handed to
  // the expanding lanes through LDS (the row-twiddle region, free at that point): R gathers per tile instead of R * C broadcast loads.
  static constexpr bool STAGE = PM == 2;
  // ... and an element meets five twiddles there: the virtual pass's at the load, two sub-round boundaries, and at the store w_n^(8 k2 i_new)
  // (row) and w_n^(k2 i1) (column).  i_new = E0 + 2^B0 E1 + 2^(B0+B1) E2 (one output digit per sub-round), so the row twiddle is a product of
  // one factor per digit: the factors of E0 and E1 ride on the boundary tables (per tile: tb0[E0][lo] in the w region, tb1[E1][lo] in the
  // row-twiddle region) and the factor of the last digit is merged with the column twiddle into ts[E_last][i1] - FOUR multiplications per
  // element instead of five.  All factors are exact field elements, so the regrouping does not change a bit of the result.
  // Goldilocks only: BabyBear's butterflies read w_r inside the sub-rounds, so its w region cannot be given away.
  static constexpr bool MERGE = PM == 2 && F::ID == 0;
  static constexpr int VLOAD_UNROLL = 4;   // twiddle gathers in flight per thread in the expansion behind the virtual pass
  static MS_DEV void locate(size_t g, size_t tiles, size_t* tile, size_t* by) { *by = g / tiles; *tile = g - *by * tiles; }
  // PM 0 / 1: the tile's rows, SWEEPS 16-byte pieces per lane (fully unrolled: `rows` never leaves the register file)
  static MS_DEV void load_rows(const Params& p, size_t tile, size_t by, int tid, V16 (&rows)[SWEEPS]) {
    const size_t n = (size_t)1 << p.log_n, cs = n >> K, f0 = tile << LC;
    const int c0 = (tid % LPR) * VEC, rb = tid / LPR;
    const T* s0 = p.src + by * p.src_bstride + f0 + c0 + (size_t)rb * cs;
    if (PM == 1 || p.n_in >= n) {
#pragma unroll
      for (int i = 0; i < SWEEPS; i++) rows[i] = *reinterpret_cast<const V16*>(s0 + (size_t)(i * RPS) * cs);
    } else {
#pragma unroll
      for (int i = 0; i < SWEEPS; i++) {
        const size_t a = f0 + c0 + (size_t)(rb + i * RPS) * cs;
#pragma unroll
        for (int v = 0; v < VEC; v++) rows[i][v] = (a + v < p.n_in) ? s0[(size_t)(i * RPS) * cs + v] : (T)0;
      }
    }
  }
  static MS_DEV void run(const Params& p, int bx, int, int nbx, int tid, unsigned char* lds) {
    T* tile = reinterpret_cast<T*>(lds);
    T* w = tile + (size_t)R * C;                 // [R]: w_r (sub-round twiddles), loaded once per workgroup (MERGE: boundary table tb0, per tile)
    T* twr = w + R;                              // [R]: store twiddle of every tile row (PM 2: staged coefficients, then tb1 | ts or the row twiddles)
    const size_t tiles = ((size_t)1 << (p.log_n - K)) >> LC;
    // r03, behind the virtual pass: the per-tile twiddle tables (tb0, tb1 | ts, or the row twiddles) depend on the tile index only, so a workgroup takes ALL batch
    // entries (columns) of a tile one after the other and builds them once per tile (a slot = a tile; otherwise a slot = one (tile, column) item)
    // (measured r03, six-column LDE 2^20 -> 2^23: Goldilocks 0.560 -> 0.555 ms; BabyBear 0.333 -> 0.375 ms - its row twiddles are cheap and the hand-over through the
    //  32-byte tile rows conflicts in LDS - so Goldilocks only)
    constexpr bool share = SHARE;   // the launcher picks this instance when tiles >= grid, tiles % 8 == 0 and the launch has more than one column
    const size_t total = share ? tiles : tiles * (size_t)p.nbatch;
    const u32 nin = share ? p.nbatch : 1u;
    const int c0 = (tid % LPR) * VEC, rb = tid / LPR;
    const bool do_scale = p.do_scale != 0;
    // work items of this workgroup: g(i) for i = 0, 1, ..; XCD x = bx & 7 walks the contiguous range [x * total / 8, (x + 1) * total / 8):
    // behind the virtual pass its workgroups share 64-byte source lines through that XCD's L2; in a later pass the 64 tiles an XCD works on
    // at a time are 4 KiB runs of every row
    const bool xcd_map = PM != 0 && (nbx & 7) == 0 && (total & 7) == 0 && total >= 64;
    const size_t stride = xcd_map ? (size_t)(nbx >> 3) : (size_t)nbx, first = xcd_map ? (size_t)(bx >> 3) : (size_t)bx;
    const size_t lim = xcd_map ? (total >> 3) : total, base_g = xcd_map ? (size_t)(bx & 7) * (total >> 3) : 0;
    if (first >= lim) return;
    [[maybe_unused]] constexpr int B0 = DG::bits(0), Q0 = 1 << DG::slo(0), Q1 = 1 << DG::slo(1);
    constexpr int NTB1 = NSUB == 3 ? (R >> B0) : 0;          // entries of tb1 (none with two sub-rounds: the second one is the last)
    static_assert(!MERGE || NTB1 + (1 << BL) * C <= R, "merged tables must fit the row-twiddle region");
    [[maybe_unused]] T* tb1 = twr; [[maybe_unused]] T* ts = twr + NTB1;
    [[maybe_unused]] T* ts1 = twr + R / 2;     // SHIFT1 + MERGE: the store table [E1][E2][c] (fa | fb sit at the start of the region)
    static_assert(!(MERGE && SHIFT1) || ((1 << DG::bits(1)) + (1 << BL) * C <= R / 2 && ((C << BL) << DG::bits(1)) <= R / 2), "store table must fit the row-twiddle region");
    if (!MERGE) for (int j = tid; j < R; j += TH) w[j] = p.w_r[j];
    size_t tl, by;
    auto slot_item = [&](size_t slot, u32 bi, size_t* t_, size_t* b_) { if (share) { *t_ = slot; *b_ = bi; } else locate(slot, tiles, t_, b_); };
    slot_item(base_g + first, 0, &tl, &by);
    constexpr int NS = STAGE ? (R + TH - 1) / TH : 1;
    [[maybe_unused]] T sv[NS];
    auto stage_issue = [&](size_t tile_, size_t by_) {
      const size_t nprime = ((size_t)1 << p.log_n) >> (LC + K);
      const T* src = p.src + by_ * p.src_bstride;
#pragma unroll
      for (int j = 0; j < NS; j++) {
        const int row = tid + j * TH;
        const size_t k = tile_ + nprime * (size_t)row;     // k2 == the tile index
        sv[j] = (row < R && k < p.n_in) ? src[k] : (T)0;
      }
    };
    if constexpr (STAGE) stage_issue(tl, by);
    // per-tile twiddle tables behind the virtual pass
    [[maybe_unused]] auto build_tb0 = [&](size_t f0) {   // tb0[E0][lo] = w_r^(E0 lo) * w_n^(X E0), X = k_low * Rp (the previous tile's last reader is behind the end-of-tile barrier)
      const size_t X = (f0 >> p.log_Rp) << p.log_Rp;
      for (int idx = tid; idx < R; idx += TH) {
        const int e = idx >> DG::slo(0), lo = idx & (Q0 - 1);
        T v = p.w_r[e * lo];
        if (X && e) v = A::mul_tw(v, tw_global(p, X * (size_t)e));
        w[idx] = v;
      }
    };
    [[maybe_unused]] auto build_tables = [&](size_t f0, bool row_tw) {
      if constexpr (MERGE && SHIFT1) {
        // The second boundary is a shift here, so the row-twiddle factor of ITS digit E1 moves into the store
        // table: ts[E1][E2][c] = fa[E1] * fb[E2][c], fa[E1] = w_n^(X E1 2^B0), fb[E2][c] = w_n^(X E2 2^(B0+B1)) * w_n^(k_low c).  fa | fb (2^B1 + 2^BL C
        // entries) are built here, the 2^(B1+BL) C products behind the next barrier (one per thread and tile); ts is read in the store phase.
        const size_t k_low = f0 >> p.log_Rp, X = k_low << p.log_Rp;
        constexpr int NFA = 1 << DG::bits(1);
        for (int idx = tid; idx < NFA + (1 << BL) * C; idx += TH) {
          T v;
          if (idx < NFA) v = tw_global(p, (X * (size_t)idx) << B0);
          else {
            const int j = idx - NFA, e = j >> LC, c = j & (C - 1);           // last digit x column
            v = tw_global(p, (X * (size_t)e) << (K - BL));
            if (k_low && c) v = A::mul_tw(v, tw_global(p, k_low * (size_t)c));
          }
          twr[idx] = v;
        }
      } else if constexpr (MERGE) {   // tb1 (read in the second sub-round) and ts (read in the store phase)
        const size_t k_low = f0 >> p.log_Rp, X = k_low << p.log_Rp;
        for (int idx = tid; idx < NTB1 + (1 << BL) * C; idx += TH) {
          T v;
          if (idx < NTB1) {
            const int e = idx >> DG::slo(1), lo = idx & (Q1 - 1);
            v = p.w_r[(e * lo) << B0];                                        // w_(Q1 2^B1)^(e lo)
            if (X && e) v = A::mul_tw(v, tw_global(p, (X * (size_t)e) << B0));
            tb1[idx] = v;
          } else {
            const int j = idx - NTB1, e = j >> LC, c = j & (C - 1);           // last digit x column
            v = tw_global(p, (X * (size_t)e) << (K - BL));
            if (k_low && c) v = A::mul_tw(v, tw_global(p, k_low * (size_t)c));
            ts[j] = v;
          }
        }
      } else if constexpr (STAGE) {   // no merged tables (BabyBear): the row twiddles (read in the store phase)
        if (row_tw) {
          const size_t k_low = f0 >> p.log_Rp;
          for (int row = tid; row < R; row += TH) twr[row] = tw_global(p, ((size_t)row_to_inew(row) * k_low) << p.log_Rp);
        }
      }
    };
    for (size_t it = first; it < lim; it += stride) {
    if constexpr (SHARE) {   // the tile's tables, once for all its batch entries (the previous tile's last readers are behind its end-of-tile barrier)
      const size_t f0t = (base_g + it) << LC;
      if constexpr (MERGE) build_tb0(f0t);
      build_tables(f0t, !p.last && (f0t >> p.log_Rp) != 0);
      msrt::wg_barrier();
    }
    for (u32 bi = 0; bi < nin; bi++) {
      slot_item(base_g + it, bi, &tl, &by);
      [[maybe_unused]] const bool new_tile = bi == 0;
      const size_t f0 = tl << LC;
      const bool row_tw = PM != 0 && !p.last && (f0 >> p.log_Rp) != 0;
      // ---- load: the tile's inputs go to LDS (behind the virtual pass: times w_(r0 r)^(i1 * row)); the row twiddles of this tile
      if constexpr (PM != 2) {
        V16 rows[SWEEPS];
        load_rows(p, tl, by, tid, rows);
        if (row_tw) {
          const size_t k_low = f0 >> p.log_Rp;
          for (int row = tid; row < R; row += TH) twr[row] = tw_global(p, ((size_t)row_to_inew(row) * k_low) << p.log_Rp);   // w_n^(k_low * Rp * i_new)
        }
#pragma unroll
        for (int i = 0; i < SWEEPS; i++) *reinterpret_cast<V16*>(tile + tix(rb + i * RPS, c0)) = rows[i];
      } else {
        // virtual r0-point pass (r0 = C, only the first n/r0 inputs non-zero): A_1[k2*r0 + i1] = w_n^(i1*k) x[k], k = k2 + nprime*row.
        // The k2 part of the twiddle rides on the store twiddle; here x[k] * w_(r0 r)^(i1 * row).  One lane per tile element.
        // The staged coefficients are handed over through the tile's own column 0 (the expansion leaves that column as it is: i1 = 0 has no twiddle),
        // so that the table regions keep their content from one batch entry of a tile to the next.
#pragma unroll
        for (int j = 0; j < NS; j++) { const int row = tid + j * TH; if (row < R) tile[tix(row, 0)] = sv[j]; }
        msrt::wg_barrier();
        if constexpr (MERGE && !SHARE) build_tb0(f0);
        if constexpr (!SHIFT1 && TH % C == 0 && (((TH >> LC) >> BL) & 3) == 0) {
          // a thread's elements are (row r0 + k * TH/C, column i1): the rows' swizzle term is the same for all k, so every address is base + k * constant
          constexpr int RSTEP = TH >> LC;
          const int i1 = tid & (C - 1), r0 = tid >> LC;
          T* tp = tile + (((r0 ^ ((r0 >> BL) & 3)) << LC));
          if (i1) {
            const T* vt = p.vtw + (size_t)i1 * r0;
#pragma unroll VLOAD_UNROLL
            for (int k = 0; k < R / RSTEP; k++) tp[i1 + k * RSTEP * C] = A::mul_tw(tp[k * RSTEP * C], vt[(size_t)i1 * (RSTEP * k)]);
          }
        } else {
#pragma unroll VLOAD_UNROLL
          for (int idx = tid; idx < R * C; idx += TH) {
            const int row = idx >> LC, i1 = idx & (C - 1);
            if (i1) tile[tix(row, i1)] = A::mul_tw(tile[tix(row, 0)], p.vtw[(size_t)i1 * row]);
          }
        }
      }
      msrt::wg_barrier();
      if constexpr (!SHARE) build_tables(f0, row_tw);   // (SHARE: built once per tile, ahead of its batch entries)
      if constexpr (STAGE) {   // the next item's coefficients: in flight during the sub-rounds and the store
        if (bi + 1 < nin) stage_issue(tl, (size_t)bi + 1);
        else if (it + stride < lim) {
          size_t ntl, nby;
          slot_item(base_g + it + stride, 0, &ntl, &nby);
          stage_issue(ntl, nby);
        }
      }
      sub_items<0, 0, MERGE>(tid, tile, w);
      msrt::wg_barrier();
      if constexpr (SHIFT1) {
        if constexpr (MERGE) {   // ts[E1][E2][c] = fa[E1] * fb[E2][c] (fa | fb complete behind the barrier above; read in the store phase, behind the next one)
          if (row_tw && new_tile) {
            constexpr int NFA = 1 << DG::bits(1);
            for (int idx = tid; idx < (NFA << BL) * C; idx += TH) {
              const int e1 = idx >> (BL + LC), j = idx & (((1 << BL) * C) - 1);
              T v = twr[NFA + j];
              if (e1) v = A::mul_tw(v, twr[e1]);
              ts1[idx] = v;
            }
          }
        }
        sub1_shift(tid, tile);
      } else if constexpr (MERGE && NSUB == 3) sub_items<1, 0, true>(tid, tile, tb1); else sub_items<1, 0>(tid, tile, w);
      msrt::wg_barrier();
      T* dst = p.dst + by * p.dst_bstride;
      if constexpr (FUSE_TAIL) {
        // ---- last sub-round + store, fused: item = (row group g of 2^BL rows, column piece cq); the rows of a group are the last digit
        // E2 = 0 .. 2^BL - 1 of the output index, i_new = E0 + 2^B0 E1 + 2^(B0+B1) E2 with (E0, E1) = the digits of g
        constexpr int NE = 1 << BL, B0_ = DG::bits(0), B1_ = DG::bits(1);
        const size_t k_low = f0 >> p.log_Rp;
        T* outb = dst + ((k_low << p.log_Rp) << K) + (f0 & (((size_t)1 << p.log_Rp) - 1));
        const bool col_tw = row_tw && PM == 2 && !MERGE;
        V16 o[NJT][NE];
#pragma unroll
        for (int J = 0; J < NJT; J++) {
          const int itq = tid + J * TH, cq = (itq % LPR) * VEC, g = itq / LPR, R0 = g << BL, h3 = (g & 3) ^ (SHIFT1 ? ((g >> DG::bits(1)) & 3) : 0);   // swizzle terms of row R0
#pragma unroll
          for (int e = 0; e < NE; e++) o[J][e] = *reinterpret_cast<const V16*>(tile + (size_t)(R0 + (e & ~3) + ((e & 3) ^ h3)) * C + cq);
        }
#pragma unroll
        for (int J = 0; J < NJT; J++) {
          const int itq = tid + J * TH, cq = (itq % LPR) * VEC, g = itq / LPR, R0 = g << BL;
          const int inew0 = (g >> B1_) | ((g & ((1 << B1_) - 1)) << B0_);
          T gc[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) gc[v] = col_tw ? tw_global(p, k_low * (size_t)(cq + v)) : F::to_tw(F::from_u64(1));
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            T x[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) x[e] = o[J][e][v];
            dif_regs<A, INV, BL>(x, w, K);
#pragma unroll
            for (int e = 0; e < NE; e++) {
              T y = x[bitrev(e, BL)];
              if constexpr (MERGE && SHIFT1) { if (row_tw) y = A::mul_tw(y, ts1[((((g & ((1 << B1_) - 1)) << BL) + e) << LC) + cq + v]); }   // row factors of the last two digits x column twiddle
              else if constexpr (MERGE) { if (row_tw) y = A::mul_tw(y, ts[(e << LC) + cq + v]); }   // last digit's row factor x column twiddle
              else if (row_tw) y = A::mul_tw(y, twr[R0 + e]);
              if (col_tw && (cq + v)) y = A::mul_tw(y, gc[v]);
              if (do_scale) y = A::mul_tw(y, p.scale);
              o[J][e][v] = y;
            }
          }
#pragma unroll
          for (int e = 0; e < NE; e++) *reinterpret_cast<V16*>(outb + cq + ((size_t)(inew0 | (e << (B0_ + B1_))) << p.log_Rp)) = o[J][e];
        }
        msrt::wg_barrier();     // the tile and the per-tile tables are free for the next work item
        continue;
      }
      if constexpr (NSUB >= 3) { sub_items<2, 0>(tid, tile, w); msrt::wg_barrier(); }
      // ---- store
      if constexpr (PM != 0) {
        // out = k_low*Rp*r + i_done + Rp*i_new: a tile row is one 16*LPR-byte run
        const size_t k_low = f0 >> p.log_Rp, i_done0 = (f0 & (((size_t)1 << p.log_Rp) - 1)) + c0;
        T* out = dst + ((k_low << p.log_Rp) << K) + i_done0;
        T gc[VEC];                                     // behind the virtual pass: w_n^(k_low * i_done), i_done = the column
        const bool col_tw = row_tw && PM == 2 && !MERGE;
#pragma unroll
        for (int v = 0; v < VEC; v++) gc[v] = col_tw ? tw_global(p, k_low * (size_t)(c0 + v)) : F::to_tw(F::from_u64(1));
#pragma unroll 2
        for (int i = 0; i < SWEEPS; i++) {
          const int row = rb + i * RPS;
          const int inew = row_to_inew(row);
          V16 o = *reinterpret_cast<const V16*>(tile + tix(row, c0));
          const T rt = (row_tw && !MERGE) ? twr[row] : F::to_tw(F::from_u64(1));
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            T x = o[v];
            if constexpr (MERGE) { if (row_tw) x = A::mul_tw(x, ts[((row & ((1 << BL) - 1)) << LC) + c0 + v]); }   // last digit's row factor x column twiddle
            else if (row_tw) x = A::mul_tw(x, rt);
            if (col_tw && (c0 + v)) x = A::mul_tw(x, gc[v]);
            if (do_scale) x = A::mul_tw(x, p.scale);
            o[v] = x;
          }
          *reinterpret_cast<V16*>(out + ((size_t)inew << p.log_Rp)) = o;
        }
      } else {
        // first pass of a plain transform: out = f*r + i_new, i_new fastest across lanes
        for (int idx = tid * VEC; idx < R * C; idx += TH * VEC) {
          const int c = idx >> K, inew0 = idx & (R - 1);
          const size_t f = f0 + c;
          V16 o;
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const int inew = inew0 + v;
            T x = tile[tix(inew_to_row(inew), c)];
            if (!p.last) { const size_t e = (size_t)inew * f; if (e) x = A::mul_tw(x, tw_global(p, e)); }
            if (do_scale) x = A::mul_tw(x, p.scale);
            o[v] = x;
          }
          *reinterpret_cast<V16*>(dst + (f << K) + inew0) = o;
        }
      }
      msrt::wg_barrier();   // the tile and its row twiddles are free for the next work item
    }
    }
  }
};


// LAST pass of a small radix (2^K <= 32 points), whole DFT in registers: the rows of this pass are contiguous runs of n / 2^K elements,
// so a lane takes VEC neighbouring columns, loads their 2^K row elements (16-byte loads, fully coalesced), transforms, stores.  No LDS, no
// twiddle tables for Goldilocks (radix <= 64 twiddles are shifts).  A streaming kernel: it lets a 2^21..2^25-point transform run as
// 2^10 x 2^10 x 2^K (two passes on the large cooperative tiles + this copy-speed pass) instead of three passes of 2^8-row tiles.
#ifndef MS_REG_TH
#define MS_REG_TH 128   // measured at 2^27 points x 6 columns: 64..128 threads 2.56-2.59 ms, 256 threads 2.98 ms, 512 threads 2.70 ms (8-byte accesses: 2.58-3.0 ms)
#endif
#ifndef MS_REG_VECMAXK
#define MS_REG_VECMAXK 4
#endif
template <class F, class A, bool INV, int K> struct RegPassKernel {
  typedef typename F::T T;
  typedef PassParams<F> Params;
  static constexpr int THREADS = MS_REG_TH;
  static constexpr int R = 1 << K;
  static constexpr int VEC = (K <= MS_REG_VECMAXK) ? 16 / (int)sizeof(T) : 1;     // 2^K * VEC elements per lane stay in registers
  typedef T V16 __attribute__((vector_size(16)));
  static_assert(K >= 1 && K <= 5, "register pass radix");
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_HD bool applicable(const Params& p) {
    return (int)p.log_r == K && p.last && p.log_r0 == 0 && p.log_rho == 0 && (int)p.log_Rp == (int)p.log_n - K && p.n_in >= ((size_t)1 << p.log_n) &&
           ((size_t)1 << p.log_Rp) % VEC == 0;
  }
  static MS_HD unsigned grid(const Params& p) { return (unsigned)(((((size_t)1 << p.log_Rp) / VEC) + THREADS - 1) / THREADS); }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int, unsigned char*) {
    const size_t cs = (size_t)1 << p.log_Rp;                         // n / 2^K: row length = distance between the points of one DFT
    const size_t i = ((size_t)bx * THREADS + tid) * VEC;
    if (i >= cs) return;
    const T* src = p.src + (size_t)by * p.src_bstride + i;
    T* dst = p.dst + (size_t)by * p.dst_bstride + i;
    const bool do_scale = p.do_scale != 0;
    if constexpr (VEC > 1) {
      V16 v[R];
#pragma unroll
      for (int k = 0; k < R; k++) v[k] = *reinterpret_cast<const V16*>(src + (size_t)k * cs);
#pragma unroll
      for (int c = 0; c < VEC; c++) {
        T x[R];
#pragma unroll
        for (int k = 0; k < R; k++) x[k] = v[k][c];
        dif_regs<A, INV, K>(x, p.w_r, K);
#pragma unroll
        for (int e = 0; e < R; e++) { T y = x[bitrev(e, K)]; if (do_scale) y = A::mul_tw(y, p.scale); v[e][c] = y; }   // v[k][c] are all consumed: reuse as the output rows
      }
#pragma unroll
      for (int e = 0; e < R; e++) *reinterpret_cast<V16*>(dst + (size_t)e * cs) = v[e];
    } else {
      T x[R];
#pragma unroll
      for (int k = 0; k < R; k++) x[k] = src[(size_t)k * cs];
      dif_regs<A, INV, K>(x, p.w_r, K);
#pragma unroll
      for (int e = 0; e < R; e++) { T y = x[bitrev(e, K)]; if (do_scale) y = A::mul_tw(y, p.scale); dst[(size_t)e * cs] = y; }
    }
  }
};

// out[k] = in[k] * s^k  (k < n) — the coset pre-scaling of
// Radix2EvaluationDomain::get_coset(shift).fft (starks.rs:82-89).
template <class F> struct ScalePowKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256;
  static constexpr int ITEMS = 16;
  struct Params { const T* src; T* dst; size_t src_bstride, dst_bstride, n; T s, s_step /* s^THREADS */; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int, unsigned char*) {
    size_t k = (size_t)bx * (THREADS * ITEMS) + tid;
    if (k >= p.n) return;
    const T* src = p.src + (size_t)by * p.src_bstride;
    T* dst = p.dst + (size_t)by * p.dst_bstride;
    T pw = f_pow<F>(p.s, k);
    for (int j = 0; j < ITEMS && k < p.n; j++, k += THREADS) {
      dst[k] = F::mul(src[k], pw);
      pw = F::mul(pw, p.s_step);
    }
  }
};

// out[r] = zeta^r * sum_j in[r + j*m] * (zeta^m)^j  (r < min(n, m)): the size-m polynomial that agrees with the n-coefficient
// input on the coset zeta*<w_m> (x^m == zeta^m there).  Its size-m NTT is the input evaluated on that coset — the part of a
// size-D evaluation domain one rank of a sharded proof owns (ms_set_shard).  n <= m degenerates to ScalePowKernel.
template <class F> struct CosetFoldKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256;
  static constexpr int ITEMS = 16;
  struct Params { const T* src; T* dst; size_t src_bstride, dst_bstride, n, m; T s, s_step /* s^THREADS */, sm /* s^m */; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int, unsigned char*) {
    size_t r = (size_t)bx * (THREADS * ITEMS) + tid;
    const size_t lim = p.n < p.m ? p.n : p.m;
    if (r >= lim) return;
    const T* src = p.src + (size_t)by * p.src_bstride;
    T* dst = p.dst + (size_t)by * p.dst_bstride;
    T pw = f_pow<F>(p.s, r);
    const size_t full = p.n / p.m, rem = p.n - full * p.m;  // uniform: r < rem has one more term
    for (int j = 0; j < ITEMS && r < lim; j++, r += THREADS) {
      const size_t terms = full + (r < rem ? 1 : 0);  // r + t*m < n
      T acc = src[r + (terms - 1) * p.m];
      for (size_t t = terms - 1; t-- > 0;) acc = F::add(F::mul(acc, p.sm), src[r + t * p.m]);
      dst[r] = F::mul(acc, pw);
      pw = F::mul(pw, p.s_step);
    }
  }
};

}  // namespace msntt
