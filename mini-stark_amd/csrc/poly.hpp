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
#include "ntt.hpp"

namespace mspoly {

constexpr int THREADS = 256;

// ---------------------------------------------------------------- TransposeIn
template <class F> struct TransposeInKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  // bad: optional device word (zeroed by the caller) set to 1 when an input element is not canonical (>= p)
  struct Params { const u64* src; T* dst; size_t N, w, dst_stride; T rinv; int mont; /* mont: src holds x*2^64 mod p; rinv = 2^-64 mod p */ u32* bad; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f >= p.N * p.w) return;
    const size_t row = f / p.w, col = f - row * p.w;
    const u64 raw = p.src[f];
    const bool oob = raw >= F::P;
    if (oob && p.bad) *p.bad = 1;            // racing writers all store the same value; the call then fails with MS_ERR_ARG
    T v = F::from_u64(oob ? 0 : raw);
    if (p.mont) v = F::mul(v, p.rinv);
    p.dst[col * p.dst_stride + row] = v;
  }
};
// The same for WIDE traces (r04: the 64-column AIR of BASELINE configs[4]): with one thread per element the writes of a wave scatter over 64 columns, 8 bytes each
// (3.8 ms for 2^22 rows x 64 columns = 4.4 x what the bytes cost); here a workgroup moves a 64 x 64 tile through LDS - rows in as 512-byte runs, columns out as
// 512-byte runs.  Grid: (rows / 64, ceil(w / 64)).
template <class F> struct TransposeInTiledKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256, TILE = 64;
  typedef typename TransposeInKernel<F>::Params Params;
  static MS_HD int nphases(const Params&) { return 2; }
  static MS_HD size_t lds_bytes() { return (size_t)TILE * (TILE + 1) * sizeof(T); }
  static MS_DEV void phase(int ph, const Params& p, int bx, int by, int tid, int nthreads, unsigned char* lds) {
    T* tile = reinterpret_cast<T*>(lds);   // [col][row], pitch TILE + 1
    const size_t row0 = (size_t)bx * TILE, col0 = (size_t)by * TILE;
    if (ph == 0) {
      for (int idx = tid; idx < TILE * TILE; idx += nthreads) {
        const int r = idx / TILE, c = idx % TILE;
        const size_t row = row0 + r, col = col0 + c;
        if (row >= p.N || col >= p.w) continue;
        const u64 raw = p.src[row * p.w + col];
        const bool oob = raw >= F::P;
        if (oob && p.bad) *p.bad = 1;
        T v = F::from_u64(oob ? 0 : raw);
        if (p.mont) v = F::mul(v, p.rinv);
        tile[c * (TILE + 1) + r] = v;
      }
      return;
    }
    for (int idx = tid; idx < TILE * TILE; idx += nthreads) {
      const int c = idx / TILE, r = idx % TILE;
      const size_t row = row0 + r, col = col0 + c;
      if (row >= p.N || col >= p.w) continue;
      p.dst[col * p.dst_stride + row] = tile[c * (TILE + 1) + r];
    }
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
    const T one = F::from_u64(1), minus_one = F::neg(one);
    for (int t = 0; t < p.k; t++) {   // the scalars are uniform over the launch: +1 / -1 (most of an AIR's transition constraints) are scalar branches, not multiplications - same canonical values
      const T v = p.polys[(size_t)p.idx[t] * p.stride + j];
      if (p.s[t] == one) acc = F::add(acc, v);
      else if (p.s[t] == minus_one) acc = F::sub(acc, v);
      else acc = F::add(acc, F::mul(p.s[t], v));
    }
    p.dst[j] = acc;
  }
};
// Several linear combinations of the SAME few columns in one sweep: out[o][j] = sum_u m[o][u] * col[src[u]][j].  Every source element is
// loaded once for all outputs (the three constraint columns of the Fibonacci LDE: 6 column transfers instead of 10).  m is uniform over the
// launch, so the zero tests are scalar branches; field addition is exact, so the order of the terms does not matter.
constexpr int LCM_SRC = 8, LCM_OUT = 4;
template <class F> struct LincombMultiKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* polys; size_t stride, n; T* dst[LCM_OUT]; int src[LCM_SRC]; T m[LCM_OUT][LCM_SRC]; int nsrc, nout; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    T v[LCM_SRC];
#pragma unroll
    for (int u = 0; u < LCM_SRC; u++) v[u] = (u < p.nsrc) ? p.polys[(size_t)p.src[u] * p.stride + j] : (T)0;
#pragma unroll
    for (int o = 0; o < LCM_OUT; o++) {
      if (o < p.nout) {
        T acc = 0;
#pragma unroll
        for (int u = 0; u < LCM_SRC; u++)
          if (u < p.nsrc && p.m[o][u] != 0) {
            if (p.m[o][u] == F::from_u64(1)) acc = F::add(acc, v[u]);
            else if (p.m[o][u] == F::neg(F::from_u64(1))) acc = F::sub(acc, v[u]);
            else acc = F::add(acc, F::mul(p.m[o][u], v[u]));
          }
        p.dst[o][j] = acc;
      }
    }
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
// Workgroup b owns coefficients [b*CH, (b+1)*CH), CH = THREADS*ITEMS: thread t accumulates
// sum_j c[b*CH + t + j*THREADS] * z^(t + j*THREADS) (its z^t is a product of the host-supplied
// z^(2^i), no per-thread exponentiation), the block sums the threads and writes the partial
// P_b (WITHOUT the factor z^(b*CH)).  ReducePartials then forms sum_b P_b (z^CH)^b.
// With a single block the result goes straight to `dst`.
constexpr int MAX_POLYS = 8;
// ITEMS_ = coefficients per thread: 16 for long polynomials, 4 up to 2^19 coefficients (MS_EVAL_SMALL_MAX), whose launches are latency - 16 dependent
// load + multiply-add iterations per thread made a 2^15-coefficient evaluation as slow as a 2^19-coefficient one, 28-34 us; with 4: 12-18 us, the Eval launches of
// one proof 400 -> 260 us, ReducePartials 62 -> 77 us; eight proofs in flight: neutral (profiles/r04_small_round_kernels_ab.log).
template <class F, int EC, int E, int ITEMS_ = 16> struct EvalKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  static constexpr int ITEMS = ITEMS_;
  struct Params {
    const T* base; size_t poly_stride, limb_stride, kstride;
    size_t off[MAX_POLYS], count[MAX_POLYS]; int npoly;
    Ext<F, E> zpow2[9];   // z^(2^i), i <= 8  (zpow2[8] = z^THREADS)
    T* partials;          // [nblocks][npoly][E]
    // a SINGLE workgroup whose partial sums are the values, stored to page-locked host memory: flag and aux word as ReducePartialsKernel's.  (More than one workgroup:
    // ReducePartialsKernel follows as a launch of its own - letting the last workgroup to finish sum the partials, as the fused FRI round does, was tried in r05 and
    // cost 8-9 us MORE per evaluation than the second launch: 512 workgroups each paying a device-scope release in front of the counter.)
    msrt::HostFlag flag;
    const unsigned long long* aux_src; unsigned long long* aux_dst;
    unsigned single;      // 1: this launch is that single workgroup
  };
  static MS_HD size_t lds_bytes() { return ((size_t)MAX_POLYS * E * (THREADS / 64) + 32 * (size_t)E) * sizeof(T); }
  // coefficient * power, the power in table form (e_to_tw)
  static MS_DEV Ext<F, E> mul_coef(const Ext<F, E>& pwt, const T* c) {
    if (EC == 1) { Ext<F, E> r; for (int l = 0; l < E; l++) r.c[l] = F::mul_tw(c[0], pwt.c[l]); return r; }
    Ext<F, E> cc; for (int l = 0; l < E; l++) cc.c[l] = c[l < EC ? l : 0];
    return e_mul_tw<F>(cc, pwt);
  }
  // Cooperative kernel (barrier inside): the block sum runs as a butterfly of wave shuffles (6 steps) plus one LDS hop between the
  // four waves, instead of `width` threads each adding 256 LDS values one after the other (a ~3 us dependent chain per launch, and
  // these launches sit on the proof's latency path: 2 per FRI round).
  static MS_DEV void run(const Params& p, int bx, int, int, int tid, unsigned char* lds) {
    T* red = reinterpret_cast<T*>(lds);  // [npoly*E][waves]
    constexpr int WAVES = THREADS / 64;
    // z^tid = z^(tid & 15) * z^(16 (tid >> 4)): 32 lanes build the two 16-entry tables (4 multiplications each), every thread then needs ONE multiplication
    // instead of the eight of a per-thread bit product (which were a third of a thread's multiplications with two polynomials and 16 coefficients each)
    T* tab = red + (size_t)MAX_POLYS * E * WAVES;  // [32][E]
    if (tid < 32) {
      Ext<F, E> tw = e_one<F, E>();
      const int j = tid & 15, b0 = tid & 16 ? 4 : 0;
#pragma unroll
      for (int i = 0; i < 4; i++) if ((j >> i) & 1) tw = e_mul<F>(tw, p.zpow2[b0 + i]);
      for (int l = 0; l < E; l++) tab[(size_t)tid * E + l] = tw.c[l];
    }
    msrt::wg_barrier();
    Ext<F, E> acc[MAX_POLYS];
    for (int i = 0; i < MAX_POLYS; i++) acc[i] = e_zero<F, E>();
    size_t k = (size_t)bx * (THREADS * ITEMS) + tid;
    size_t kmax = 0;
    for (int i = 0; i < MAX_POLYS; i++) if (i < p.npoly && p.count[i] > kmax) kmax = p.count[i];
    if (k < kmax) {
      Ext<F, E> plo, phi;
      for (int l = 0; l < E; l++) { plo.c[l] = tab[(size_t)(tid & 15) * E + l]; phi.c[l] = tab[(size_t)(16 + (tid >> 4)) * E + l]; }
      const Ext<F, E> zstep = e_to_tw<F, E>(p.zpow2[8]);
      Ext<F, E> pw = e_to_tw<F, E>(e_mul<F>(plo, phi));   // z^k in table form from here on: every product below is one multiplication + one reduction per limb pair
      for (int it = 0; it < ITEMS && k < kmax; it++, k += THREADS) {
#pragma unroll
        for (int i = 0; i < MAX_POLYS; i++) {
          if (i < p.npoly && k < p.count[i]) {
            T c[EC];
            const T* ptr = p.base + (size_t)i * p.poly_stride + p.off[i] + k * p.kstride;
            for (int l = 0; l < EC; l++) c[l] = ptr[(size_t)l * p.limb_stride];
            acc[i] = e_add<F, E>(acc[i], mul_coef(pw, c));
          }
        }
        pw = e_mul_tw<F>(pw, zstep);
      }
    }
    const int width = p.npoly * E;
#pragma unroll
    for (int i = 0; i < MAX_POLYS; i++) {
      if (i < p.npoly) {
#pragma unroll
        for (int l = 0; l < E; l++) {
          T v = acc[i].c[l];
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) v = F::add(v, (T)msrt::wave_shfl_xor((unsigned long long)v, m));
          if ((tid & 63) == 0) red[(size_t)(i * E + l) * WAVES + (tid >> 6)] = v;
        }
      }
    }
    msrt::wg_barrier();
    if (tid < width) {
      T s = 0;
      for (int t = 0; t < WAVES; t++) s = F::add(s, red[(size_t)tid * WAVES + t]);
      p.partials[(size_t)bx * width + tid] = s;
    }
    if (p.single) {
      if (tid == 0 && p.aux_src) *p.aux_dst = *p.aux_src;
      msrt::raise_host_flag_wg(p.flag, tid);
    }
  }
};
// out[i] = sum_b partials[b][i] * zc^b,  i < npoly (E limbs each), zc = z^CH.
// One workgroup: thread t runs Horner over its contiguous range of blocks, scales by
// (zc^S)^t (bit product of zs2[i] = (zc^S)^(2^i)) and the block adds the threads.
template <class F, int E> struct ReducePartialsKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  // sum_b P_b zc^b with thread t taking the blocks b = t, t + THREADS, t + 2 THREADS, ...: neighbouring lanes read neighbouring partials and the loads of a
  // thread do not depend on each other (r03: with a contiguous run of blocks per thread every iteration waited for its own uncoalesced load - 6.4 ms of the
  // 66 ms of a 2^24-row proof sat in this one-workgroup kernel).  zc = z^CH; zs2[i] = zc^(2^i), i < 8; zc_step = zc^THREADS.
  struct Params { const T* partials; size_t nblocks, per_thread /* ceil(nblocks / THREADS) */; int npoly; Ext<F, E> zc_step; Ext<F, E> zs2[8]; T* out; msrt::HostFlag flag; /* raised behind `out` (page-locked host memory then) */
                  const unsigned long long* aux_src; unsigned long long* aux_dst; /* optional: a device word that travels to page-locked host memory with the values (the validity polynomial's trimmed length, found by ms_mix, on DEEP-ALI's last evaluation) */ };
  static MS_HD size_t lds_bytes() { return (size_t)MAX_POLYS * E * (THREADS / 64) * sizeof(T); }
  static MS_DEV void run(const Params& p, int, int, int, int tid, unsigned char* lds) {   // cooperative: wave-shuffle block sum, as EvalKernel
    T* red = reinterpret_cast<T*>(lds);
    constexpr int WAVES = THREADS / 64;
    const int width = p.npoly * E;
    Ext<F, E> acc[MAX_POLYS];
    for (int i = 0; i < MAX_POLYS; i++) acc[i] = e_zero<F, E>();
    if ((size_t)tid < p.nblocks) {
      for (size_t j = p.per_thread; j-- > 0;) {   // Horner in zc^THREADS over this thread's blocks, last one first
        const size_t b = (size_t)tid + j * THREADS;
        const bool have = b < p.nblocks;
#pragma unroll
        for (int i = 0; i < MAX_POLYS; i++) {
          if (i < p.npoly) {
            Ext<F, E> v; for (int l = 0; l < E; l++) v.c[l] = have ? p.partials[b * width + i * E + l] : (T)0;
            acc[i] = e_add<F, E>(e_mul<F>(acc[i], p.zc_step), v);
          }
        }
      }
      Ext<F, E> sc = e_one<F, E>();   // zc^tid
#pragma unroll
      for (int i = 0; i < 8; i++) if ((tid >> i) & 1) sc = e_mul<F>(sc, p.zs2[i]);
#pragma unroll
      for (int i = 0; i < MAX_POLYS; i++) if (i < p.npoly) acc[i] = e_mul<F>(acc[i], sc);
    }
#pragma unroll
    for (int i = 0; i < MAX_POLYS; i++) {
      if (i < p.npoly) {
#pragma unroll
        for (int l = 0; l < E; l++) {
          T v = acc[i].c[l];
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) v = F::add(v, (T)msrt::wave_shfl_xor((unsigned long long)v, m));
          if ((tid & 63) == 0) red[(size_t)(i * E + l) * WAVES + (tid >> 6)] = v;
        }
      }
    }
    msrt::wg_barrier();
    if (tid < width) {
      T s = 0;
      for (int t = 0; t < WAVES; t++) s = F::add(s, red[(size_t)tid * WAVES + t]);
      p.out[tid] = s;
    }
    if (tid == 0 && p.aux_src) *p.aux_dst = *p.aux_src;
    msrt::raise_host_flag_wg(p.flag, tid);
  }
};

// ---------------------------------------------------------------- Fold
// dst_j = src[2j] + alpha * src[2j+1],  j < m = ceil(n/2)   (fri.rs:329-343 split + 361-372 fold)
template <class F, int E> struct FoldKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  // first_out (optional; a rank of a sharded proof): the first folded element also goes to first_out[0 .. E) and E zero limbs behind it - the payload of the scan's carry
  // exchange ([first element | aggregate]; the aggregate launch overwrites the zeros), which took a fill and a copy launch of its own until r05
  struct Params { const T* src; size_t src_limb_stride, n; T* dst; size_t dst_limb_stride; Ext<F, E> alpha; T* first_out; };
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
    if (j == 0 && p.first_out) for (int l = 0; l < E; l++) { p.first_out[l] = r.c[l]; p.first_out[E + l] = 0; }
  }
};

// ---------------------------------------------------------------- SuffixHorner
// A job describes one vector f_j (j < m): limb l at in[l*in_limb_stride + in_off + j*in_stride].
// Block b of a job owns [b*BS, (b+1)*BS), BS = THREADS*SEG.  Jobs are batched over blockIdx.y
// (a device table, or one job inline in the kernel arguments).
//   mode AGG  : agg[l*agg_limb_stride + b] = sum_{k in block} f_k z^(k - b*BS)
//   mode FINAL: with carry-in Hin_b = carry[l*carry_limb_stride + b] (carry == null: 0),
//               writes H_j for j >= 1 to out[l*out_limb_stride + out_off + (j-1)*out_stride]
//               (u64 or T elements, out == null: discarded) and H_0 to h0[l];
//               tail_zero: also stores 0 at out index m-1 (the output is the carry array of
//               the level below, whose last block has no carry-in); tail_zero == 2: stores the
//               carry-in propagated to position m instead (jobs with a carry-in at the top level).
// zpow[i] = z^(SEG * 2^i), i < 9.
constexpr int SH_SEG = 8;
constexpr int SH_BS = THREADS * SH_SEG;
template <class F, int E> struct SHJob {
  typedef typename F::T T;
  const T* in; size_t in_limb_stride, in_off, in_stride, m;
  void* out; size_t out_limb_stride, out_off, out_stride;
  T* h0;
  T* agg; size_t agg_limb_stride;
  const T* carry; size_t carry_limb_stride;
  u32 out_u64, tail_zero;
  u32 out_h0;   // H_0 also goes to `out`, one out_stride below out_off (a rank of a sharded proof whose range does not start at coefficient 0: its H_0 is a quotient coefficient)
  Ext<F, E> z; Ext<F, E> zpow[9];
};
template <class F, int E> struct SuffixHornerKernel {
  typedef typename F::T T;
  typedef SHJob<F, E> Job;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const Job* jobs; Job inline_job; int final_mode; };
  static MS_HD size_t lds_bytes() { return ((size_t)E * SH_BS + 2 * (size_t)E * (THREADS + 1)) * sizeof(T); }
  // Cooperative kernel (r04; thirteen barrier-separated phases of the generic phase harness until then): the job is read ONCE - as phases, every step fetched its
  // fields from the job table again behind a barrier that also drains vector memory, and a launch cost 17-19 us whatever its size; these launches sit on the proof's
  // latency path (one to three per FRI round).
  static MS_DEV void run(const Params& p, int bx, int by, int, int tid, unsigned char* lds) {
    const Job jb = p.jobs ? p.jobs[by] : p.inline_job;
    const size_t nb = jb.m ? (jb.m + SH_BS - 1) / SH_BS : 1;
    if ((size_t)bx >= nb) return;   // (the whole workgroup)
    T* fbuf = reinterpret_cast<T*>(lds);                 // [E][BS]
    T* sa = fbuf + (size_t)E * SH_BS;                    // [E][THREADS+1]  ping
    T* sb = sa + (size_t)E * (THREADS + 1);              // pong
    const size_t j0 = (size_t)bx * SH_BS;
    // coalesced load
    for (int i = tid; i < SH_BS; i += THREADS) {
      const size_t j = j0 + i;
      for (int l = 0; l < E; l++)
        fbuf[(size_t)l * SH_BS + i] = (j < jb.m) ? jb.in[(size_t)l * jb.in_limb_stride + jb.in_off + j * jb.in_stride] : (T)0;
    }
    msrt::wg_barrier();
    const Ext<F, E> z = e_to_tw<F, E>(jb.z);   // table form: the multiplicand of every Horner step (e_mul_tw)
    {  // per-thread segment aggregate a_t
      Ext<F, E> a = e_zero<F, E>();
#pragma unroll
      for (int i = SH_SEG - 1; i >= 0; i--) {
        Ext<F, E> c; for (int l = 0; l < E; l++) c.c[l] = fbuf[(size_t)l * SH_BS + tid * SH_SEG + i];
        a = e_add<F, E>(e_mul_tw<F>(a, z), c);
      }
      for (int l = 0; l < E; l++) sa[(size_t)l * (THREADS + 1) + tid] = a.c[l];
      if (tid == 0) {  // virtual element THREADS = carry-in
        for (int l = 0; l < E; l++) {
          T cv = 0;
          if (p.final_mode && jb.carry) cv = jb.carry[(size_t)l * jb.carry_limb_stride + bx];
          sa[(size_t)l * (THREADS + 1) + THREADS] = cv;
        }
      }
    }
    msrt::wg_barrier();
#pragma unroll
    for (int step = 0; step < 9; step++) {  // suffix scan over THREADS+1 entries, distance d = 2^step
      const int d = 1 << step;
      T* src = (step & 1) ? sb : sa;
      T* dst = (step & 1) ? sa : sb;
      const Ext<F, E> zp = e_to_tw<F, E>(jb.zpow[step]);
      for (int t = tid; t <= THREADS; t += THREADS) {
        Ext<F, E> v; for (int l = 0; l < E; l++) v.c[l] = src[(size_t)l * (THREADS + 1) + t];
        if (t + d <= THREADS) {
          Ext<F, E> u; for (int l = 0; l < E; l++) u.c[l] = src[(size_t)l * (THREADS + 1) + t + d];
          v = e_add<F, E>(v, e_mul_tw<F>(u, zp));
        }
        for (int l = 0; l < E; l++) dst[(size_t)l * (THREADS + 1) + t] = v.c[l];
      }
      msrt::wg_barrier();
    }
    T* sc = sb;  // 9 steps: the last one (step 8, even) wrote sb
    if (!p.final_mode) {
      if (tid == 0) for (int l = 0; l < E; l++) jb.agg[(size_t)l * jb.agg_limb_stride + bx] = sc[(size_t)l * (THREADS + 1)];
      return;
    }
    {
      Ext<F, E> h; for (int l = 0; l < E; l++) h.c[l] = sc[(size_t)l * (THREADS + 1) + tid + 1];
#pragma unroll
      for (int i = SH_SEG - 1; i >= 0; i--) {
        Ext<F, E> c; for (int l = 0; l < E; l++) c.c[l] = fbuf[(size_t)l * SH_BS + tid * SH_SEG + i];
        h = e_add<F, E>(e_mul_tw<F>(h, z), c);
        for (int l = 0; l < E; l++) fbuf[(size_t)l * SH_BS + tid * SH_SEG + i] = h.c[l];
      }
    }
    msrt::wg_barrier();
    for (int i = tid; i < SH_BS; i += THREADS) {
      const size_t j = j0 + i;
      if (j >= jb.m && !(j == 0)) continue;
      for (int l = 0; l < E; l++) {
        const T v = fbuf[(size_t)l * SH_BS + i];
        if (j == 0) {
          if (jb.h0) jb.h0[l] = v;
          if (jb.out_h0 && jb.out) {
            const size_t o = (size_t)l * jb.out_limb_stride + jb.out_off - jb.out_stride;
            if (jb.out_u64) reinterpret_cast<u64*>(jb.out)[o] = F::to_u64(v); else reinterpret_cast<T*>(jb.out)[o] = v;
          }
        } else if (jb.out) {
          const size_t o = (size_t)l * jb.out_limb_stride + jb.out_off + (j - 1) * jb.out_stride;
          if (jb.out_u64) reinterpret_cast<u64*>(jb.out)[o] = F::to_u64(v); else reinterpret_cast<T*>(jb.out)[o] = v;
        }
      }
      if (jb.tail_zero && jb.out && j + 1 == jb.m) {
        for (int l = 0; l < E; l++) {
          // tail_zero == 2 (a job with an external carry-in at its top level): the last block of the level below is not the end of the vector - its carry-in is the
          // suffix sum at position m of THIS level = the block's carry-in propagated through the zero padding behind element m - 1 (fbuf holds H at every position)
          T tv = 0;
          if (jb.tail_zero == 2) tv = (i + 1 < SH_BS) ? fbuf[(size_t)l * SH_BS + i + 1] : sc[(size_t)l * (THREADS + 1) + THREADS];
          const size_t o = (size_t)l * jb.out_limb_stride + jb.out_off + j * jb.out_stride;
          if (jb.out_u64) reinterpret_cast<u64*>(jb.out)[o] = F::to_u64(tv); else reinterpret_cast<T*>(jb.out)[o] = tv;
        }
      }
    }
  }
};

// ---------------------------------------------------------------- FRI fold in the evaluation domain
// The codeword of FRI round i+1 straight from the codeword of round i, without a transform:
// with f = even(x^2) + x odd(x^2) the round polynomial is g(y) = (even(y) + alpha odd(y) - c) / (y - z)
// (fri.rs:96-101: fold, subtract B(alpha), exact division by (x - z)), and on the next domain y = x^2
//   even(y) = (f(x) + f(-x)) / 2,   odd(y) = (f(x) - f(-x)) / (2x),   1/(y - z) = C(y) / n(y)
// with C the product of the conjugates of (y - z) and n its norm down to the base field:
//   Fp2 (Goldilocks):  C = (y - z0) + z1 u,                    n = (y - z0)^2 - 7 z1^2
//   Fp4 (BabyBear):    A = (y - z0) - z1 u, M = A^2 - (z2 + z3 u)^2 (u - 11), C = (A conj(M), (z2 + z3 u) conj(M)), n = M0^2 - 11 M1^2.
// Field arithmetic is exact, so these are the values the size-D/2 NTT of g's coefficients (fri.rs:350) would give.
// z outside the base field keeps the norm non-zero; the host falls back to the NTT otherwise.
// One thread owns ITEMS outputs and inverts their norms together (Montgomery's trick + one field inversion).
// Index maps: output j = t*m_out + i (t < groups), inputs f(x) at t*2*m_out + i and f(-x) m_out further, domain index
// r = i (replicated round) or 2*(rank + W*i) + t (sharded round: this rank's two cosets).
template <class F> MS_HD typename F::T fold_base_inv(typename F::T x) { return f_inv<F>(x); }
template <> MS_HD u64 fold_base_inv<GL>(u64 x) { return gl_inv_chain(x); }
// ITEMS_: outputs per thread.  8 amortises the field inversion over eight norms (the throughput choice); 1 is the LATENCY choice for the small late rounds, where a
// launch is one wave deep anyway and eight serial outputs per thread only stretch it (with 8 a launch took 22 us whatever its size; one proof alone on the GPU:
// 118.6 -> 120.4 proofs/s, eight in flight: neutral - profiles/r04_small_round_kernels_ab.log).
template <class F, int E, int ITEMS_ = 8> struct FriFoldEvalKernel {
  typedef typename F::T T;
  typedef Ext<F, E> X;
  static_assert(E == 2 || E == 4, "quadratic or quartic tower");
  static constexpr int THREADS = mspoly::THREADS;
  static constexpr int ITEMS = ITEMS_;
  struct Params {
    const T* src; size_t src_limb_stride; T* dst; size_t dst_limb_stride;
    size_t m_out; u32 log_m /* log2 m_out */, groups, shard_W, shard_k;
    const T* tw_lo; const T* tw_hi; u32 lo_bits, log_D;   // F::to_tw(w_D^e) = tw_lo[e & mask] * tw_hi[e >> lo_bits] (the size-D forward plan)
    X alpha, c2 /* 2 (B0 + B1 alpha) */, z;
    T inv2;
  };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV T w_tab(const Params& p, size_t e) {
    T tw = p.tw_lo[e & (((size_t)1 << p.lo_bits) - 1)];
    const size_t eh = e >> p.lo_bits;
    if (eh) tw = F::mul_tw(tw, p.tw_hi[eh]);
    return tw;
  }
  // conjugate product C and base-field norm n of (y - z), yd = y - z0
  static MS_DEV void inv_parts(const Params& p, T yd, X* C, T* n) {
    const T nr = F::from_u64(F::NR2);
    if constexpr (E == 2) {
      C->c[0] = yd; C->c[1] = p.z.c[1];
      *n = F::sub(F::mul(yd, yd), F::mul(nr, F::mul(p.z.c[1], p.z.c[1])));
    } else {
      Ext<F, 2> A, zb, K;
      A.c[0] = yd; A.c[1] = F::neg(p.z.c[1]);
      zb.c[0] = p.z.c[2]; zb.c[1] = p.z.c[3];
      K = e_mul_nr4<F>(e_mul<F>(zb, zb));
      const Ext<F, 2> M = e_sub<F, 2>(e_mul<F>(A, A), K);
      Ext<F, 2> Mc; Mc.c[0] = M.c[0]; Mc.c[1] = F::neg(M.c[1]);
      *n = F::sub(F::mul(M.c[0], M.c[0]), F::mul(nr, F::mul(M.c[1], M.c[1])));
      const Ext<F, 2> c0 = e_mul<F>(A, Mc), c1 = e_mul<F>(zb, Mc);
      C->c[0] = c0.c[0]; C->c[1] = c0.c[1]; C->c[2] = c1.c[0]; C->c[3] = c1.c[1];
    }
  }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t total = p.m_out * p.groups, Dm = ((size_t)1 << p.log_D) - 1;
    const size_t j0 = (size_t)bx * ((size_t)nthreads * ITEMS) + tid;
    X tt[ITEMS]; T nrm[ITEMS], pre[ITEMS];
    T run = F::from_u64(1);
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
      const size_t j = j0 + (size_t)it * nthreads;
      tt[it] = e_zero<F, E>(); nrm[it] = F::from_u64(1);
      if (j < total) {
        const size_t t = j >> p.log_m, i = j & (p.m_out - 1);
        const size_t ia = t * 2 * p.m_out + i, ib = ia + p.m_out;
        const size_t r = p.shard_W ? 2 * ((size_t)p.shard_k + (size_t)p.shard_W * i) + t : i;
        X a, b;
#pragma unroll
        for (int l = 0; l < E; l++) { a.c[l] = p.src[(size_t)l * p.src_limb_stride + ia]; b.c[l] = p.src[(size_t)l * p.src_limb_stride + ib]; }
        const X s = e_add<F, E>(a, b), d = e_sub<F, E>(a, b);
        const T xinv = w_tab(p, (Dm + 1 - r) & Dm);                       // table form of x^-1 = w_D^(D - r)
        X u;                                                               // alpha / x
#pragma unroll
        for (int l = 0; l < E; l++) u.c[l] = F::mul_tw(p.alpha.c[l], xinv);
        const X num = e_sub<F, E>(e_add<F, E>(s, e_mul<F>(u, d)), p.c2);   // 2 (folded - c)
        const T y = F::from_tw(w_tab(p, (2 * r) & Dm));                   // y = x^2
        X C;
        inv_parts(p, F::sub(y, p.z.c[0]), &C, &nrm[it]);
        tt[it] = e_mul<F>(num, C);
      }
      pre[it] = run;                 // product of the norms before this one
      run = F::mul(run, nrm[it]);
    }
    T inv = F::mul(fold_base_inv<F>(run), p.inv2);   // 1 / (2 * product of all norms)
#pragma unroll
    for (int it = ITEMS - 1; it >= 0; it--) {
      const size_t j = j0 + (size_t)it * nthreads;
      const T ni = F::mul(inv, pre[it]);             // 1 / (2 norm_it)
      inv = F::mul(inv, nrm[it]);
      if (j < total) {
#pragma unroll
        for (int l = 0; l < E; l++) p.dst[(size_t)l * p.dst_limb_stride + j] = F::mul(tt[it].c[l], ni);
      }
    }
  }
};

// ---------------------------------------------------------------- Degree / FindFirst
// result = max(j+1) over nonzero elements (0 for the zero polynomial); caller zeroes result.
// Elements are visited from the top: the first workgroups to run see the leading coefficient, publish the
// answer, and every later workgroup leaves after one read of `result` (no loads, no atomics).
template <class F, int E> struct DegreeKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  // idx0: index of src[0] in the whole vector (a rank's part of a distributed polynomial reports GLOBAL lengths, which the ranks then max-reduce)
  struct Params { const T* src; size_t limb_stride, n; unsigned long long* result; size_t idx0; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t r = (size_t)bx * nthreads + tid;
    if (r >= p.n) return;
    const size_t j = p.n - 1 - r;
    if ((unsigned long long)(p.idx0 + j + 1) <= *reinterpret_cast<volatile unsigned long long*>(p.result)) return;
    bool nz = false, nz_next = false;
    for (int l = 0; l < E; l++) nz = nz || (p.src[(size_t)l * p.limb_stride + j] != 0);
    // only the top of a run of non-zero elements can be the answer: one atomic for a dense polynomial, not one per element
    if (nz && j + 1 < p.n) for (int l = 0; l < E; l++) nz_next = nz_next || (p.src[(size_t)l * p.limb_stride + j + 1] != 0);
    if (nz && !nz_next) msrt::atomic_max_u64(p.result, (unsigned long long)(p.idx0 + j + 1));
  }
};
// per window: result[t] = min index j with leaf_j == target_t (caller fills result with ~0).
// Jobs batched over blockIdx.y (device table) or one inline job.
// Sharded codewords (ms_set_shard): the local element j = t*map_m + i is global element map_g*(map_k + map_W*i) + t and the
// result is the minimum GLOBAL index (map_g = 0: identity).
template <class F, int E> struct FindJob {
  const typename F::T* src; size_t limb_stride, n;
  const typename F::T* targets /* [nt][E] */; int nt; unsigned long long* result;
  u32 map_g, map_W, map_k; size_t map_m;
};
template <class F, int E> struct FindFirstKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const FindJob<F, E>* jobs; FindJob<F, E> inline_job; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& pp, int bx, int by, int tid, int nthreads, unsigned char*) {
    const FindJob<F, E>& p = pp.jobs ? pp.jobs[by] : pp.inline_job;
    const size_t j = (size_t)bx * nthreads + tid;
    if (j >= p.n) return;
    T v[E];
    for (int l = 0; l < E; l++) v[l] = p.src[(size_t)l * p.limb_stride + j];
    for (int t = 0; t < p.nt; t++) {
      bool eq = true;
      for (int l = 0; l < E; l++) eq = eq && (v[l] == p.targets[(size_t)t * E + l]);
      if (eq) {
        unsigned long long gj = (unsigned long long)j;
        if (p.map_g) { const size_t tt = j / p.map_m, ii = j - tt * p.map_m; gj = (unsigned long long)p.map_g * (p.map_k + (unsigned long long)p.map_W * ii) + tt; }
        msrt::atomic_min_u64(p.result + t, gj);
      }
    }
  }
};

// ---------------------------------------------------------------- query points
// One thread per (window i, query j) (fri.rs:148-154).  h0[(i*nq + j)*2 + s] holds even(x3) (s=0) /
// odd(x3) (s=1) of round i's polynomial at x3 = x1^2 — the H_0 outputs of the suffix Horner runs;
// window W (the last round) is only evaluated.  y1 = Ee + x1*Eo, y2 = Ee - x1*Eo and
// y3 = p_{i+1}(x3) = Ee' + x1'*Eo' with x1' = x3 (the next window's y1).
// Writes x1 y1 x2 y2 x3 y3 | qlen into the proof blob and (y1, y2) to the find-first targets.
template <class F, int E> struct QueryPointsKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params {
    const T* h0;            // [(W+1)*nq*2][E]
    const T* x1;            // [(W+1)*nq] base elements
    const u64* qlen;        // [W]
    int W, nq;
    unsigned char* blob; const size_t* rec_off;  // [W*nq] byte offset of the record
    T* targets;             // [W][2*nq][E]
  };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const int t = bx * nthreads + tid;
    if (t >= p.W * p.nq) return;
    const int i = t / p.nq, j = t - i * p.nq;
    Ext<F, E> ee, eo, ne, no;
    for (int l = 0; l < E; l++) {
      ee.c[l] = p.h0[((size_t)t * 2) * E + l]; eo.c[l] = p.h0[((size_t)t * 2 + 1) * E + l];
      ne.c[l] = p.h0[((size_t)(t + p.nq) * 2) * E + l]; no.c[l] = p.h0[((size_t)(t + p.nq) * 2 + 1) * E + l];
    }
    const T x1 = p.x1[t], x3 = p.x1[t + p.nq];
    Ext<F, E> d = e_mul_base<F, E>(eo, x1);
    Ext<F, E> y1 = e_add<F, E>(ee, d), y2 = e_sub<F, E>(ee, d);
    Ext<F, E> y3 = e_add<F, E>(ne, e_mul_base<F, E>(no, x3));
    u64* o = reinterpret_cast<u64*>(p.blob + p.rec_off[t]);
    Ext<F, E> X1 = e_from_base<F, E>(x1), X2 = e_from_base<F, E>(F::neg(x1)), X3 = e_from_base<F, E>(x3);
    for (int l = 0; l < E; l++) o[l] = F::to_u64(X1.c[l]);
    for (int l = 0; l < E; l++) o[E + l] = F::to_u64(y1.c[l]);
    for (int l = 0; l < E; l++) o[2 * E + l] = F::to_u64(X2.c[l]);
    for (int l = 0; l < E; l++) o[3 * E + l] = F::to_u64(y2.c[l]);
    for (int l = 0; l < E; l++) o[4 * E + l] = F::to_u64(X3.c[l]);
    for (int l = 0; l < E; l++) o[5 * E + l] = F::to_u64(y3.c[l]);
    o[6 * E] = p.qlen[i];
    T* tg = p.targets + ((size_t)i * 2 * p.nq + 2 * j) * E;
    for (int l = 0; l < E; l++) { tg[l] = y1.c[l]; tg[E + l] = y2.c[l]; }
  }
};


// ---------------------------------------------------------------- one proof over several ranks: coefficient-domain work by coefficient range
// ("fold/split/lincomb shard by coefficient index; the scan needs a carry all-gather; the evaluations an e-limb sum of partial dot products").
// Rank r of W holds the coefficients [r*S, (r+1)*S) of a polynomial.  A sum over all coefficients is the Horner combination of the ranks' partial sums in z^S;
// a suffix sum (SuffixHorner) takes the sum over the higher ranks as carry-in.  All ranks see the same all-gathered partials and compute the same field
// elements: the results are bit-identical to the unsharded kernels' (exact arithmetic), which the gloo tests check against the oracle at world 2 / 4 / 8.
//
// ShardCombine: out[i] = sum_r gathered[r][i] * zstep^r for n extension elements (a rank's payload = n*E limbs, rank_stride limbs apart)
template <class F, int E> struct ShardCombineKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params { const T* gathered; size_t rank_stride; u32 n, W; Ext<F, E> zstep; T* out; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const u32 i = (u32)bx * (u32)nthreads + (u32)tid;
    if (i >= p.n) return;
    Ext<F, E> acc = e_zero<F, E>();
    for (int r = (int)p.W - 1; r >= 0; r--) {
      Ext<F, E> v; for (int l = 0; l < E; l++) v.c[l] = p.gathered[(size_t)r * p.rank_stride + (size_t)i * E + l];
      acc = e_add<F, E>(e_mul<F>(acc, p.zstep), v);
    }
    for (int l = 0; l < E; l++) p.out[(size_t)i * E + l] = acc.c[l];
  }
};
// ShardCarry: the suffix sums at the rank boundaries from the all-gathered per-rank aggregates.  With T_W = 0,
//     T_r = F_r + zA * (A_r + zB * T_(r+1))          A_r: rank r's aggregate (E limbs at agg_off of its payload), F_r: an optional leading element (first_off)
// For this rank k the job stores  carry_out = T_(k+1) * scale  (the top-level carry-in of its SuffixHorner job: that kernel places a top-level carry at the padded
// position BS^levels, so scale = z^(m - BS^levels) moves it to the job's true end m),  tail_out = T_(k+1) itself, and h0_out = T_0 (the sum over everything).
//   DEEP quotient of a FRI round (fold_commit): the job runs over the rank's folded elements f[1..cnt), F_r = f[0], zA = z, zB = z^(S-1);
//   query quotients: the job runs over the rank's whole half-range, no F, zA = 1, zB = x3^(S/2).
template <class F, int E> struct CarryJob {
  typedef typename F::T T;
  u32 first_off, agg_off, has_first, pad_;
  Ext<F, E> zA, zB, scale;
  T* carry_out; T* tail_out; size_t tail_stride; T* h0_out;
};
template <class F, int E> struct ShardCarryKernel {
  typedef typename F::T T;
  typedef CarryJob<F, E> Job;
  static constexpr int THREADS = 64;
  struct Params { const Job* jobs; Job inline_job; u32 njobs, W, rank; const T* gathered; size_t rank_stride; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const u32 q = (u32)bx * (u32)nthreads + (u32)tid;
    if (q >= p.njobs) return;
    const Job& jb = p.jobs ? p.jobs[q] : p.inline_job;
    Ext<F, E> t = e_zero<F, E>();
    for (int r = (int)p.W - 1; r >= 0; r--) {
      if ((u32)r == p.rank) {   // t = T_(rank+1)
        if (jb.carry_out) { const Ext<F, E> c = e_mul<F>(t, jb.scale); for (int l = 0; l < E; l++) jb.carry_out[l] = c.c[l]; }
        if (jb.tail_out) for (int l = 0; l < E; l++) jb.tail_out[(size_t)l * jb.tail_stride] = t.c[l];
        if (!jb.h0_out) return;
      }
      const T* pay = p.gathered + (size_t)r * p.rank_stride;
      Ext<F, E> a; for (int l = 0; l < E; l++) a.c[l] = pay[jb.agg_off + l];
      Ext<F, E> x = e_add<F, E>(a, e_mul<F>(jb.zB, t));
      if (jb.has_first) {
        Ext<F, E> f; for (int l = 0; l < E; l++) f.c[l] = pay[jb.first_off + l];
        t = e_add<F, E>(f, e_mul<F>(jb.zA, x));
      } else t = x;
    }
    if (jb.h0_out) for (int l = 0; l < E; l++) jb.h0_out[l] = t.c[l];
  }
};
// the all-gathered parts of a distributed polynomial (rank r: E limbs of S coefficients, rank_stride limbs apart) -> one replicated vector of `count` coefficients
template <class F, int E> struct GatherPolyKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  struct Params { const T* gathered; size_t rank_stride, S, count; T* dst; size_t dst_limb_stride; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t f = (size_t)bx * nthreads + tid;
    if (f >= p.count * E) return;
    const size_t l = f / p.count, g = f - l * p.count, r = g / p.S, i = g - r * p.S;
    p.dst[l * p.dst_limb_stride + g] = p.gathered[r * p.rank_stride + l * p.S + i];
  }
};

// ---------------------------------------------------------------- build-defined degree-3 composition (ms_mix_cubic)
// BASELINE configs[4] names "degree-3 constraints", which the reference cannot express (quirk Q1: its `validity` polynomial is the REMAINDER of the division by
// the vanishing polynomial, so any constraint polynomial with >= N coefficients panics).  The build-defined variant, with the TRUE quotient:
//   C_t(x)      = P_j(w x) - P_a(x) P_b(x) P_c(x) - s P_d(x)            transition constraint t = (j, a, b, c, d, s); w = the trace domain's generator
//   validity(x) = (sum_t r^t C_t(x)) (x - w^(N-1)) / (x^N - 1)           exact when every C_t vanishes on rows 0 .. N-2; 2N coefficients
// evaluated pointwise on the committed LDE domain x_i = shift g_L^i (column-major LDE, P(w x_i) = the value `blowup` rows further on); x^N - 1 takes only
// `blowup` distinct values there, whose inverses come in a table.  The caller interpolates (size-L INTT) and un-shifts.
template <class F> struct CubicSpec { u32 j, a, b, c, d; typename F::T s, rpow; };
template <class F> struct CubicComposeKernel {
  typedef typename F::T T;
  static constexpr int THREADS = mspoly::THREADS;
  static constexpr int ITEMS = 8;
  struct Params { const T* lde; size_t L; u32 blowup, ncons; const CubicSpec<F>* spec; const T* den_inv; T shift, gL, gL_step /* gL^THREADS */, w_last /* w^(N-1) */; T* out; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int, unsigned char*) {
    size_t i = (size_t)bx * (THREADS * ITEMS) + tid;
    if (i >= p.L) return;
    T x = F::mul(p.shift, f_pow<F>(p.gL, i));
    for (int it = 0; it < ITEMS && i < p.L; it++, i += THREADS) {
      const size_t in = i + p.blowup < p.L ? i + p.blowup : i + p.blowup - p.L;   // w x_i = x_(i + blowup)
      T acc = 0;
      for (u32 t = 0; t < p.ncons; t++) {
        const CubicSpec<F> c = p.spec[t];
        T v = F::mul(F::mul(p.lde[(size_t)c.a * p.L + i], p.lde[(size_t)c.b * p.L + i]), p.lde[(size_t)c.c * p.L + i]);
        v = F::add(v, F::mul(c.s, p.lde[(size_t)c.d * p.L + i]));
        acc = F::add(acc, F::mul(c.rpow, F::sub(p.lde[(size_t)c.j * p.L + in], v)));
      }
      p.out[i] = F::mul(F::mul(acc, F::sub(x, p.w_last)), p.den_inv[i % p.blowup]);
      x = F::mul(x, p.gL_step);
    }
  }
};

// ms_arith_selftest: one operation of the NTT tiles' arithmetic class per element (A = GLM for Goldilocks: inline asm with hand-managed
// gfx950 wait states, which only a run on the device can check; BB for BabyBear).  Ops: include/ministark.h.
template <int S> struct ArithShift { template <class A> static MS_DEV u64 run(u64 x, int s) { return s == S ? msntt::gl_mul_pow2<S, A>(x) : ArithShift<S - 1>::template run<A>(x, s); } };
template <> struct ArithShift<0> { template <class A> static MS_DEV u64 run(u64 x, int) { return x; } };
template <class F, class A> struct ArithKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256;
  struct Params { const u64* a; const u64* b; u64* out; size_t n; int op; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int, unsigned char*) {
    const size_t i = (size_t)bx * THREADS + tid;
    if (i >= p.n) return;
    const T a = F::from_u64(p.a[i]), b = F::from_u64(p.b[i]);
    T r = 0;
    if (p.op == 0) r = A::add(a, b);
    else if (p.op == 1) r = A::sub(a, b);
    else if (p.op == 2) r = A::mul(a, b);
    else if (p.op == 3) r = A::mul_tw(a, F::to_tw(b));
    if constexpr (F::ID == 0) {
      if (p.op == 4) r = A::mul_x32(a);
      else if (p.op == 5) r = A::mul_x64(a);
      else if (p.op == 6) r = ArithShift<95>::template run<A>(a, (int)(p.b[i] % 96));
      else if (p.op == 7) r = A::fold_small(a, (u32)p.b[i] & 0x7FFFFFFFu);
    }
    p.out[i] = F::to_u64(r);
  }
};
}  // namespace mspoly
