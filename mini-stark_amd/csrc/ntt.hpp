// ntt.hpp — radix-2 NTT / INTT / zero-padded coset evaluation over the base
// field, natural order in and out, as 1–4 HBM passes of LDS-resident tiles.
//
// Replaces ark-poly's Radix2EvaluationDomain::{ifft, fft} at the reference call
// sites  src/air.rs:154 (INTT of trace columns), src/starks.rs:89 (coset LDE)
// and src/fri.rs:350 (FRI codeword; extension limbs are independent base
// transforms because the domain points are base-field elements).
//
// Decomposition (self-sorting, no bit-reversal pass).  n = r_1 r_2 ... r_P.
// After pass p the array holds A_p[k_rest * R_p + i_done]: `i_done` = the low
// output digits already produced (R_p = r_1..r_p values), `k_rest` = the input
// index digits not yet transformed.  Pass p+1 transforms the TOP digit of
// k_rest (stride n / r) with an r-point DFT, multiplies by the inter-pass
// twiddle w_{n/R_p}^(i_new * k_low) and stores to
//        A_{p+1}[k_low * R_p * r + i_done + R_p * i_new].
// A workgroup owns a tile of r rows x C consecutive "columns" f = k_low*R_p +
// i_done, so every global access is a run of C consecutive elements (C = 16:
// 128 B for Goldilocks); the first pass writes its tile transposed (r
// consecutive elements per column).  Algorithmic HBM traffic: 2 * n * sizeof(T)
// per pass.
//
// Inside a tile the r-point DFT runs as register sub-rounds of 2^b points
// (b <= 4) with LDS exchanges between them: DIF, digits taken from the top,
// results left digit-reversed in LDS and un-reversed by the store phase.
#pragma once
#include "field.hpp"

namespace msntt {

constexpr int TILE_LOG_C = 4;       // 16 columns per tile
constexpr int MAX_LOG_R = 10;       // tile rows <= 1024
constexpr int THREADS = 256;

// sub-round digit sizes for a tile of 2^K rows (top digit first)
MS_HD int subround_count(int K) { return K <= 4 ? 1 : (K <= 8 ? 2 : 3); }
MS_HD int subround_bits(int K, int s) {
  // K<=4: {K}; 5:{3,2} 6:{3,3} 7:{4,3} 8:{4,4}; 9:{3,3,3} 10:{4,3,3}
  if (K <= 4) return K;
  if (K <= 8) { int a = (K + 1) / 2; return s == 0 ? a : K - a; }
  if (K == 9) return 3;
  return s == 0 ? 4 : 3;
}

template <class F> struct PassParams {
  typedef typename F::T T;
  const T* src; T* dst;
  size_t src_bstride, dst_bstride;  // elements between consecutive batch entries (blockIdx.y)
  size_t n_in;                      // valid input elements (zero padded up to n); first pass only
  const T* tw_lo; const T* tw_hi;   // w_n^j = tw_lo[j & lo_mask] * tw_hi[j >> lo_bits]
  const T* w_r;                     // w_r^j, j < r
  T scale;                          // multiplied into the output of the last pass (1 = none)
  u32 log_n, log_r, log_Rp, log_C, lo_bits;
  u32 first, last;
};

template <class F, int B> MS_DEV void dif_regs(typename F::T (&x)[1 << B], const typename F::T* w_r, int log_r) {
  // in-register DIF of 2^B points; output left in bit-reversed register order.
  // stage twiddles w_{2h}^j = w_r[j * r / (2h)]
#pragma unroll
  for (int s = B - 1; s >= 0; s--) {
    const int h = 1 << s;
#pragma unroll
    for (int blk = 0; blk < (1 << B); blk += 2 * h) {
#pragma unroll
      for (int j = 0; j < h; j++) {
        typename F::T a = x[blk + j], b = x[blk + j + h];
        x[blk + j] = F::add(a, b);
        typename F::T d = F::sub(a, b);
        if (j != 0) d = F::mul(d, w_r[(size_t)j << (log_r - s - 1)]);
        x[blk + j + h] = d;
      }
    }
  }
}
MS_HD int bitrev(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }

// LDS row -> output digit index (and back) for the digit-reversed tile
MS_HD int row_to_inew(int row, int K) {
  int S = subround_count(K), done = 0, inew = 0;
  for (int s = 0; s < S; s++) {
    int b = subround_bits(K, s);
    int e = (row >> (K - done - b)) & ((1 << b) - 1);
    inew |= e << done;
    done += b;
  }
  return inew;
}
MS_HD int inew_to_row(int inew, int K) {
  int S = subround_count(K), done = 0, row = 0;
  for (int s = 0; s < S; s++) {
    int b = subround_bits(K, s);
    int e = (inew >> done) & ((1 << b) - 1);
    row |= e << (K - done - b);
    done += b;
  }
  return row;
}

template <class F> struct PassKernel {
  typedef typename F::T T;
  typedef PassParams<F> Params;
  static constexpr int THREADS = msntt::THREADS;

  static MS_HD int nphases(const Params& p) { return 2 + subround_count((int)p.log_r); }
  static MS_HD size_t lds_bytes(int log_r, int log_C) {
    size_t C = (size_t)1 << log_C, CP = C + (C > 1 ? 1 : 0);
    return (((size_t)1 << log_r) * CP + ((size_t)1 << log_r)) * sizeof(T);
  }

  template <int B>
  static MS_DEV void subround(const Params& p, int tid, int nthreads, T* tile, const T* w, int s_lo) {
    const int K = (int)p.log_r, C = 1 << p.log_C, CP = C + (C > 1 ? 1 : 0);
    const int q = 1 << s_lo;                      // row distance between the 2^B points
    const int items = (1 << (K - B)) << p.log_C;  // work items in the tile
    for (int it = tid; it < items; it += nthreads) {
      const int cidx = it & (C - 1);
      const int g = it >> p.log_C;           // (hi, lo) packed
      const int lo = g & (q - 1), hi = g >> s_lo;
      const int row0 = (hi << (s_lo + B)) + lo;
      T x[1 << B];
#pragma unroll
      for (int t = 0; t < (1 << B); t++) x[t] = tile[(row0 + t * q) * CP + cidx];
      dif_regs<F, B>(x, w, K);
      // x[bitrev(e)] = y[e]; twiddle w_{q 2^B}^(e*lo) = w_r[e * lo * r / (q 2^B)]
#pragma unroll
      for (int e = 0; e < (1 << B); e++) {
        T v = x[bitrev(e, B)];
        if (e != 0 && lo != 0) v = F::mul(v, w[((size_t)(e * lo)) << (K - s_lo - B)]);
        tile[(row0 + e * q) * CP + cidx] = v;
      }
    }
  }

  static MS_DEV void phase(int ph, const Params& p, int bx, int by, int tid, int nthreads, unsigned char* lds) {
    const int K = (int)p.log_r, r = 1 << K, C = 1 << p.log_C, CP = C + (C > 1 ? 1 : 0);
    T* tile = reinterpret_cast<T*>(lds);
    T* w = tile + (size_t)r * CP;
    const size_t n = (size_t)1 << p.log_n;
    const size_t col_stride = n >> K;  // n / r : distance between tile rows in the source
    const size_t f0 = (size_t)bx << p.log_C;
    const int S = subround_count(K);
    if (ph == 0) {
      const T* src = p.src + (size_t)by * p.src_bstride;
      for (int idx = tid; idx < r * C; idx += nthreads) {
        const int row = idx >> p.log_C, cidx = idx & (C - 1);
        const size_t a = f0 + cidx + (size_t)row * col_stride;
        tile[row * CP + cidx] = (a < p.n_in) ? src[a] : (T)0;
      }
      for (int j = tid; j < r; j += nthreads) w[j] = p.w_r[j];
      return;
    }
    if (ph <= S) {
      const int s = ph - 1;
      int done = 0;
      for (int t = 0; t < s; t++) done += subround_bits(K, t);
      const int b = subround_bits(K, s);
      const int s_lo = K - done - b;
      switch (b) {
        case 1: subround<1>(p, tid, nthreads, tile, w, s_lo); break;
        case 2: subround<2>(p, tid, nthreads, tile, w, s_lo); break;
        case 3: subround<3>(p, tid, nthreads, tile, w, s_lo); break;
        default: subround<4>(p, tid, nthreads, tile, w, s_lo); break;
      }
      return;
    }
    // store phase
    T* dst = p.dst + (size_t)by * p.dst_bstride;
    const size_t lo_mask = ((size_t)1 << p.lo_bits) - 1;
    const bool do_scale = p.scale != F::from_u64(1);
    for (int idx = tid; idx < r * C; idx += nthreads) {
      int row, cidx, inew;
      if (p.first) { cidx = idx >> K; inew = idx & (r - 1); row = inew_to_row(inew, K); }  // transposed: i_new fastest
      else { row = idx >> p.log_C; cidx = idx & (C - 1); inew = row_to_inew(row, K); }
      T v = tile[row * CP + cidx];
      const size_t f = f0 + cidx;
      const size_t k_low = f >> p.log_Rp, i_done = f & (((size_t)1 << p.log_Rp) - 1);
      if (!p.last) {
        const size_t e = ((size_t)inew * k_low) << p.log_Rp;  // exponent of w_n, < n
        if (e != 0) {
          T tw = p.tw_lo[e & lo_mask];
          const size_t eh = e >> p.lo_bits;
          if (eh) tw = F::mul(tw, p.tw_hi[eh]);
          v = F::mul(v, tw);
        }
      }
      if (do_scale) v = F::mul(v, p.scale);
      const size_t out = ((k_low << p.log_Rp) << K) + i_done + ((size_t)inew << p.log_Rp);
      dst[out] = v;
    }
  }
};

// out[k] = in[k] * s^k * mult  (k < n) — the coset pre-scaling of
// Radix2EvaluationDomain::get_coset(shift).fft (starks.rs:82-89).
template <class F> struct ScalePowKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256;
  static constexpr int ITEMS = 16;
  struct Params { const T* src; T* dst; size_t src_bstride, dst_bstride, n; T s, s_step /* s^THREADS */; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int nthreads, unsigned char*) {
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

}  // namespace msntt
