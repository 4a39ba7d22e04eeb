// poly.hpp — coefficient-domain kernels of the prover path.
//
//   TransposeIn   row-major trace -> column-major columns   (src/air.rs:151-153 gather)
//   Lincomb       constraint poly = sum_t s_t * P_idx[t]     (tests/e2e_goldilocks.rs:48-59 closures)
//   Mix           validity = sum_i r^i f_i                   (src/starks.rs:108-119, quirk Q1)
//   Eval          sum_k c_k z^k at an extension point        (src/starks.rs:142-149, src/fri.rs:354-359,151-153)
//   Fold          even + alpha * odd                         (src/fri.rs:361-372)
//   SuffixHorner  H_j = sum_{k>=j} f_k z^(k-j): quotient by (x - z) is q_j = H_{j+1},
//                 remainder/evaluation is H_0                (src/fri.rs:91-101 and 159-167)
//   Degree        trimmed length of a coefficient vector     ([ark-mem] DensePolynomial trims)
//   FindFirst     first index whose leaf equals a value      (src/merkle.rs:216-225, quirk Q7)
//
// All are streaming kernels over SoA limb arrays: HBM-bound, algorithmic bytes
// = elements read + written once.
#pragma once
#include "field.hpp"

namespace mspoly {

constexpr int THREADS = 256;

// ---------------------------------------------------------------- TransposeIn
template <class F> struct TransposeInKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const u64* src; T* dst; size_t N, w, dst_stride; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f >= p.N * p.w) return;
    const size_t row = f / p.w, col = f - row * p.w;
    p.dst[col * p.dst_stride + row] = F::from_u64(p.src[f]);
  }
};
// plain widening / narrowing copies between the u64 ABI and device storage
template <class F> struct NarrowKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const u64* src; T* dst; size_t n; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f < p.n) p.dst[f] = F::from_u64(p.src[f]);
  }
};
// dst[j*out_stride + l] (u64) = src[l*limb_stride + j]   (SoA device limbs -> AoS u64 ABI order)
template <class F> struct WidenKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* src; u64* dst; size_t n, limb_stride; u32 E; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f >= p.n * p.E) return;
    const size_t j = f / p.E; const u32 l = (u32)(f - j * p.E);
    p.dst[f] = F::to_u64(p.src[(size_t)l * p.limb_stride + j]);
  }
};
// row-major u64 matrix out of column-major device columns (LDE read-back: src/air.rs:52-58 layout)
template <class F> struct TransposeOutKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* src; u64* dst; size_t rows, cols, src_stride; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f >= p.rows * p.cols) return;
    const size_t row = f / p.cols, col = f - row * p.cols;
    p.dst[f] = F::to_u64(p.src[col * p.src_stride + row]);
  }
};

// ---------------------------------------------------------------- Lincomb / Mix
constexpr int MAX_TERMS = 8;
template <class F> struct LincombKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* polys; size_t stride, n; T* dst; T s[MAX_TERMS]; int idx[MAX_TERMS]; int k; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    T acc = 0;
    for (int t = 0; t < p.k; t++) acc = F::add(acc, F::mul(p.s[t], p.polys[(size_t)p.idx[t] * p.stride + j]));
    p.dst[j] = acc;
  }
};
template <class F> struct MixKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* polys; size_t stride, n; int c; T r; T* dst; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    T acc = p.polys[(size_t)(p.c - 1) * p.stride + j];
    for (int i = p.c - 2; i >= 0; i--) acc = F::add(F::mul(acc, p.r), p.polys[(size_t)i * p.stride + j]);
    p.dst[j] = acc;
  }
};

// ---------------------------------------------------------------- Eval
// Up to MAX_POLYS polynomials with EC-limb coefficients evaluated at one E-limb
// point.  Coefficient k of poly i, limb l: base[i*poly_stride + l*limb_stride + off[i] + k*kstride].
// Each workgroup reduces a chunk of THREADS*ITEMS coefficients to one partial
// sum per polynomial; ReducePartials adds the partials.
constexpr int MAX_POLYS = 8;
template <class F, int EC, int E> struct EvalKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  static constexpr int ITEMS = 16;
  struct Params {
    const T* base; size_t poly_stride, limb_stride, kstride;
    size_t off[MAX_POLYS], count[MAX_POLYS]; int npoly;
    Ext<F, E> z, z_step;  // z_step = z^THREADS
    T* partials;          // [nblocks][npoly][E]
  };
  static MS_HD int nphases(const Params&) { return 2; }
  static MS_HD size_t lds_bytes() { return (size_t)MAX_POLYS * E * THREADS * sizeof(T); }
  static MS_DEV Ext<F, E> mul_coef(const Ext<F, E>& pw, const T* c) {
    if (EC == 1) return e_mul_base<F, E>(pw, c[0]);
    Ext<F, E> cc; for (int l = 0; l < E; l++) cc.c[l] = c[l < EC ? l : 0];
    return e_mul<F>(pw, cc);
  }
  static MS_DEV void phase(int ph, const Params& p, int bx, int, int tid, int nthreads, unsigned char* lds) {
    T* red = reinterpret_cast<T*>(lds);  // [npoly*E][THREADS]
    if (ph == 0) {
      Ext<F, E> acc[MAX_POLYS];
      for (int i = 0; i < MAX_POLYS; i++) acc[i] = e_zero<F, E>();
      size_t k = (size_t)bx * (THREADS * ITEMS) + tid;
      Ext<F, E> pw = e_pow<F, E>(p.z, k);
      for (int it = 0; it < ITEMS; it++, k += THREADS) {
#pragma unroll
        for (int i = 0; i < MAX_POLYS; i++) {
          if (i < p.npoly && k < p.count[i]) {
            T c[EC];
            const T* ptr = p.base + (size_t)i * p.poly_stride + p.off[i] + k * p.kstride;
            for (int l = 0; l < EC; l++) c[l] = ptr[(size_t)l * p.limb_stride];
            acc[i] = e_add<F, E>(acc[i], mul_coef(pw, c));
          }
        }
        pw = e_mul<F>(pw, p.z_step);
      }
#pragma unroll
      for (int i = 0; i < MAX_POLYS; i++)
        if (i < p.npoly)
          for (int l = 0; l < E; l++) red[(size_t)(i * E + l) * nthreads + tid] = acc[i].c[l];
      return;
    }
    if (tid < p.npoly * E) {
      T s = 0;
      for (int t = 0; t < nthreads; t++) s = F::add(s, red[(size_t)tid * nthreads + t]);
      p.partials[(size_t)bx * (p.npoly * E) + tid] = s;
    }
  }
};
template <class F> struct ReducePartialsKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params { const T* partials; size_t nblocks; int width; T* out; };  // out[width]
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int, int, int tid, int, unsigned char*) {
    if (tid >= p.width) return;
    T s = 0;
    for (size_t b = 0; b < p.nblocks; b++) s = F::add(s, p.partials[b * p.width + tid]);
    p.out[tid] = s;
  }
};

// ---------------------------------------------------------------- Fold
// dst_j = src[2j] + alpha * src[2j+1],  j < m = ceil(n/2)   (fri.rs:329-343 split + 361-372 fold)
template <class F, int E> struct FoldKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* src; size_t src_limb_stride, n; T* dst; size_t dst_limb_stride; Ext<F, E> alpha; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    const size_t m = (p.n + 1) / 2;
    if (j >= m) return;
    Ext<F, E> ev, od = e_zero<F, E>();
    for (int l = 0; l < E; l++) ev.c[l] = p.src[(size_t)l * p.src_limb_stride + 2 * j];
    if (2 * j + 1 < p.n) for (int l = 0; l < E; l++) od.c[l] = p.src[(size_t)l * p.src_limb_stride + 2 * j + 1];
    Ext<F, E> r = e_add<F, E>(ev, e_mul<F>(p.alpha, od));
    for (int l = 0; l < E; l++) p.dst[(size_t)l * p.dst_limb_stride + j] = r.c[l];
  }
};

// ---------------------------------------------------------------- SuffixHorner
// f_j (j < m): limb l at in[l*in_limb_stride + in_off + j*in_stride].
// Block b owns [b*BS, (b+1)*BS), BS = THREADS*SEG.
//   mode AGG  : agg[l*agg_limb_stride + b] = sum_{k in block} f_k z^(k - b*BS)
//   mode FINAL: with carry-in Hin_b = carry[l*carry_limb_stride + b] (carry == null: 0),
//               writes H_j for j >= 1 to out[l*out_limb_stride + out_off + (j-1)*out_stride]
//               (OutT = T or u64) and H_0 to h0[l].
// zpow[i] = z^(SEG * 2^i), i < 9.
constexpr int SH_SEG = 8;
constexpr int SH_BS = THREADS * SH_SEG;
template <class F, int E, class OutT> struct SuffixHornerKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params {
    const T* in; size_t in_limb_stride, in_off, in_stride, m;
    Ext<F, E> z; Ext<F, E> zpow[9];
    int final_mode;
    T* agg; size_t agg_limb_stride;
    const T* carry; size_t carry_limb_stride;
    OutT* out; size_t out_limb_stride, out_off, out_stride;
    T* h0;
    // batching over blockIdx.y: element offsets added per batch entry
    size_t in_boff, out_boff, h0_boff, agg_boff, carry_boff;
    const Ext<F, E>* zs;  // optional per-batch z / zpow table: zs[by*10 + 0] = z, [1..9] = zpow
  };
  static MS_HD int nphases(const Params&) { return 2 + 9 + 2; }
  static MS_HD size_t lds_bytes() { return ((size_t)E * SH_BS + 2 * (size_t)E * (THREADS + 1)) * sizeof(T); }
  static MS_DEV void phase(int ph, const Params& p, int bx, int by, int tid, int nthreads, unsigned char* lds) {
    T* fbuf = reinterpret_cast<T*>(lds);                 // [E][BS]
    T* sa = fbuf + (size_t)E * SH_BS;                    // [E][THREADS+1]  ping
    T* sb = sa + (size_t)E * (THREADS + 1);              // pong
    const size_t j0 = (size_t)bx * SH_BS;
    const Ext<F, E> z = p.zs ? p.zs[(size_t)by * 10] : p.z;
    if (ph == 0) {  // coalesced load
      const T* in = p.in + (size_t)by * p.in_boff;
      for (int i = tid; i < SH_BS; i += nthreads) {
        const size_t j = j0 + i;
        for (int l = 0; l < E; l++)
          fbuf[(size_t)l * SH_BS + i] = (j < p.m) ? in[(size_t)l * p.in_limb_stride + p.in_off + j * p.in_stride] : (T)0;
      }
      return;
    }
    if (ph == 1) {  // per-thread segment aggregate a_t
      Ext<F, E> a = e_zero<F, E>();
      for (int i = SH_SEG - 1; i >= 0; i--) {
        Ext<F, E> c; for (int l = 0; l < E; l++) c.c[l] = fbuf[(size_t)l * SH_BS + tid * SH_SEG + i];
        a = e_add<F, E>(e_mul<F>(a, z), c);
      }
      for (int l = 0; l < E; l++) sa[(size_t)l * (THREADS + 1) + tid] = a.c[l];
      if (tid == 0) {  // virtual element THREADS = carry-in
        for (int l = 0; l < E; l++) {
          T cv = 0;
          if (p.final_mode && p.carry) cv = p.carry[(size_t)by * p.carry_boff + (size_t)l * p.carry_limb_stride + bx];
          sa[(size_t)l * (THREADS + 1) + THREADS] = cv;
        }
      }
      return;
    }
    if (ph < 2 + 9) {  // suffix scan over THREADS+1 entries, distance d = 2^(ph-2)
      const int step = ph - 2, d = 1 << step;
      T* src = (step & 1) ? sb : sa;
      T* dst = (step & 1) ? sa : sb;
      const Ext<F, E> zp = p.zs ? p.zs[(size_t)by * 10 + 1 + step] : p.zpow[step];
      for (int t = tid; t <= THREADS; t += nthreads) {
        Ext<F, E> v; for (int l = 0; l < E; l++) v.c[l] = src[(size_t)l * (THREADS + 1) + t];
        if (t + d <= THREADS) {
          Ext<F, E> u; for (int l = 0; l < E; l++) u.c[l] = src[(size_t)l * (THREADS + 1) + t + d];
          v = e_add<F, E>(v, e_mul<F>(u, zp));
        }
        for (int l = 0; l < E; l++) dst[(size_t)l * (THREADS + 1) + t] = v.c[l];
      }
      return;
    }
    T* sc = sb;  // 9 steps: last write went to sb (step 8 is even -> dst = sb)
    if (ph == 2 + 9) {
      if (!p.final_mode) {
        if (tid == 0) for (int l = 0; l < E; l++) p.agg[(size_t)by * p.agg_boff + (size_t)l * p.agg_limb_stride + bx] = sc[(size_t)l * (THREADS + 1)];
        return;
      }
      Ext<F, E> h; for (int l = 0; l < E; l++) h.c[l] = sc[(size_t)l * (THREADS + 1) + tid + 1];
      for (int i = SH_SEG - 1; i >= 0; i--) {
        Ext<F, E> c; for (int l = 0; l < E; l++) c.c[l] = fbuf[(size_t)l * SH_BS + tid * SH_SEG + i];
        h = e_add<F, E>(e_mul<F>(h, z), c);
        for (int l = 0; l < E; l++) fbuf[(size_t)l * SH_BS + tid * SH_SEG + i] = h.c[l];
      }
      return;
    }
    if (!p.final_mode) return;
    OutT* out = p.out + (size_t)by * p.out_boff;
    for (int i = tid; i < SH_BS; i += nthreads) {
      const size_t j = j0 + i;
      if (j >= p.m) continue;
      for (int l = 0; l < E; l++) {
        const T v = fbuf[(size_t)l * SH_BS + i];
        if (j == 0) { if (p.h0) p.h0[(size_t)by * p.h0_boff + l] = v; }
        else out[(size_t)l * p.out_limb_stride + p.out_off + (j - 1) * p.out_stride] = (OutT)F::to_u64(v);
      }
    }
  }
};

// ---------------------------------------------------------------- Degree / FindFirst
// result[by] = max(j+1) over nonzero elements (0 for the zero polynomial); caller zeroes result.
template <class F, int E> struct DegreeKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* src; size_t limb_stride, n; unsigned long long* result; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    bool nz = false;
    for (int l = 0; l < E; l++) nz = nz || (p.src[(size_t)l * p.limb_stride + j] != 0);
    if (nz) msrt::atomic_max_u64(p.result, (unsigned long long)(j + 1));
  }
};
// result[t] = min index j with leaf_j == target_t (caller fills result with ~0)
template <class F, int E> struct FindFirstKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* src; size_t limb_stride, n; const T* targets /* [nt][E] */; int nt; unsigned long long* result; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    T v[E];
    for (int l = 0; l < E; l++) v[l] = p.src[(size_t)l * p.limb_stride + j];
    for (int t = 0; t < p.nt; t++) {
      bool eq = true;
      for (int l = 0; l < E; l++) eq = eq && (v[l] == p.targets[(size_t)t * E + l]);
      if (eq) msrt::atomic_min_u64(p.result + t, (unsigned long long)j);
    }
  }
};

// ---------------------------------------------------------------- query points
// Per query t (fri.rs:148-154): from Ee = even(x3), Eo = odd(x3) (h0 buffers), x1 and y3:
//   y1 = Ee + x1*Eo, y2 = Ee - x1*Eo; writes x1 y1 x2 y2 x3 y3 (E u64 limbs each) to the
//   proof blob and (y1, y2) to the find-first target list.
template <class F, int E> struct QueryPointsKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params {
    const T* h0e; const T* h0o; const T* y3;   // [nq][E] each
    const T* x1; const T* x3;                   // [nq] base elements
    int nq;
    unsigned char* blob; const size_t* blob_off;  // byte offset of the points record per query
    T* targets;                                    // [2*nq][E]
  };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int, int, int tid, int, unsigned char*) {
    if (tid >= p.nq) return;
    Ext<F, E> ee, eo, y3;
    for (int l = 0; l < E; l++) { ee.c[l] = p.h0e[tid * E + l]; eo.c[l] = p.h0o[tid * E + l]; y3.c[l] = p.y3[tid * E + l]; }
    const T x1 = p.x1[tid], x3 = p.x3[tid];
    Ext<F, E> t = e_mul_base<F, E>(eo, x1);
    Ext<F, E> y1 = e_add<F, E>(ee, t), y2 = e_sub<F, E>(ee, t);
    u64* o = reinterpret_cast<u64*>(p.blob + p.blob_off[tid]);
    Ext<F, E> X1 = e_from_base<F, E>(x1), X2 = e_from_base<F, E>(F::neg(x1)), X3 = e_from_base<F, E>(x3);
    for (int l = 0; l < E; l++) o[l] = F::to_u64(X1.c[l]);
    for (int l = 0; l < E; l++) o[E + l] = F::to_u64(y1.c[l]);
    for (int l = 0; l < E; l++) o[2 * E + l] = F::to_u64(X2.c[l]);
    for (int l = 0; l < E; l++) o[3 * E + l] = F::to_u64(y2.c[l]);
    for (int l = 0; l < E; l++) o[4 * E + l] = F::to_u64(X3.c[l]);
    for (int l = 0; l < E; l++) o[5 * E + l] = F::to_u64(y3.c[l]);
    for (int l = 0; l < E; l++) { p.targets[(2 * tid) * E + l] = y1.c[l]; p.targets[(2 * tid + 1) * E + l] = y2.c[l]; }
  }
};

}  // namespace mspoly
