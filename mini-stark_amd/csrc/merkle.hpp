// merkle.hpp — SHA-256 Merkle commitment kernels.
//
// Replaces MerkleTree::new (src/merkle.rs:81-148):
//   leaf pass  : one digest per group of `leafs_per_node` field elements,
//                D(to_string(x0) || to_string(x1) || ...)         (merkle.rs:124-128,162-168)
//   inner pass : D(child_0 || ... || child_{ic-1}), level by level, nodes
//                appended level-major, root last                   (merkle.rs:131-140,171-177)
// [ark-mem] `to_string` is arkworks' Display: canonical decimal for Fp (ZERO
// prints as the empty string when zero_as_empty=1), and
// "QuadExtField(c0 + c1 * u)" (nested) for the extension towers.
//
// One thread owns one digest.  An element becomes 4-digit decimal chunks packed as
// big-endian ASCII words (no per-digit work), appended through a 64-bit funnel
// into a per-thread ring of message words that lives in LDS word-interleaved
// across the workgroup (bank-conflict free); complete 64-byte blocks are
// compressed with SHA-256 state and schedule in registers.  Both passes are
// integer-VALU-issue bound, not HBM bound: algorithmic traffic is
// lpn*E*sizeof(T) + 32 bytes per leaf group and 96 bytes per inner node against
// 2-3 compressions (~1440 instructions each).
#pragma once
#include "field.hpp"

namespace msmerkle {

constexpr int THREADS = 256;
struct alignas(16) uint4_t { u32 x, y, z, w; };

MS_HD u32 rotr32(u32 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}
MS_HD u32 bswap32(u32 x) { return __builtin_bswap32(x); }
// gfx950 v_bitop3_b32 (arbitrary 3-input bit function) issues in ~2.7 cycles per wave-instruction where v_bfi_b32 and
// the other 3-source VOP3 ops take ~4.5 (tools/valu_rate.hip, profiles/r01_valu_issue_rate.txt).
MS_HD u32 ch3(u32 e, u32 f, u32 g) {  // e ? f : g
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA);
#else
  return g ^ (e & (f ^ g));
#endif
}
MS_HD u32 maj3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
#else
  return (a & b) | (c & (a | b));
#endif
}
// a ^ b ^ c in one instruction on gfx950 (v_bitop3_b32, truth table 0x96)
MS_HD u32 xor3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

struct Sha256 {
  u32 st[8];
  MS_HD void init() {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
  }
  // one compression of the 16 big-endian words w[0..15] (clobbers w)
  MS_HD void compress(u32 (&w)[16]) {
    constexpr u32 K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      if (i >= 16) {
        u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        u32 s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
        u32 s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
        w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
      }
      u32 S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
      u32 ch = ch3(e, f, g);
      u32 t1 = h + S1 + ch + K[i] + w[i & 15];
      u32 S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
      u32 mj = maj3(a, b, c);
      u32 t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  }
};

// byte stream -> SHA-256.  The message bytes are packed into big-endian words in a
// per-thread ring of RW words that lives in LDS word-interleaved across the
// workgroup (word i of thread t at ring[i*nthreads + t]: conflict free).
// `put` never compresses; the owner calls `drain` after each element, so the
// 64-round compression is instantiated exactly ONCE in the kernel (inlining it
// at every put() site made the kernel I-cache bound).
template <int RW> struct ShaStream {
  static_assert((RW & (RW - 1)) == 0 && RW >= 32, "ring must be a power of two >= 2 blocks");
  Sha256 h;
  u32* ring; int nthreads, tid;
  u32 acc;     // the low `pbits` bits are bytes not yet emitted as a whole word
  u32 pbits;   // 8 * pending bytes: 0, 8, 16 or 24
  u32 total;   // bytes appended
  u32 done;    // bytes compressed
  MS_HD void init(u32* lds_words, int nthreads_, int tid_) { h.init(); ring = lds_words; nthreads = nthreads_; tid = tid_; acc = 0; pbits = 0; total = 0; done = 0; }
  MS_HD void emit(u32 word) { ring[((total >> 2) & (RW - 1)) * nthreads + tid] = word; }
  // append four bytes (most significant first): the stream word that completes is one funnel shift of (pending, w)
  MS_HD void append4(u32 w) {
#if defined(__HIP_DEVICE_COMPILE__)
    emit(__builtin_amdgcn_alignbit(acc, w, pbits));
#else
    emit((u32)((((u64)acc << 32) | w) >> pbits));
#endif
    acc = w;
    total += 4;
  }
  // append the k (1..4) low-order bytes of w, most significant first
  MS_HD void append(u32 w, u32 k) {
    const u32 kb = 8 * k;
    const u64 a64 = ((u64)acc << kb) | (u64)(k == 4 ? w : (w & ((1u << kb) - 1u)));
    const u32 nb = pbits + kb;
    if (nb >= 32) { emit((u32)(a64 >> (nb - 32))); pbits = nb - 32; }
    else pbits = nb;
    acc = (u32)a64;
    total += k;
  }
  MS_HD void put(u32 ch) { append(ch, 1); }
  MS_HD void drain() {
    while (total - done >= 64) {
      u32 w[16];
      // `done` is a multiple of 64 bytes, so the block starts at a ring word that is a multiple of 16: no per-word wrap
      const u32* blk = ring + ((done >> 2) & (RW - 1)) * nthreads + tid;
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = blk[i * nthreads];
      h.compress(w);
      done += 64;
    }
  }
  // FIPS 180-4 padding; caller drains afterwards
  MS_HD void pad() {
    const u64 bits = (u64)total * 8;
    append(0x80, 1);
    while (total & 3) append(0, 1);
    while ((total & 63) != 56) append4(0);
    append4((u32)(bits >> 32));
    append4((u32)bits);
  }
};

// four decimal digits of c < 10^4 as big-endian ASCII
MS_HD u32 pack4(u32 c) {
  const u32 q = (c * 5243u) >> 19, r = c - q * 100u;      // c / 100, c % 100
  const u32 d3 = (q * 205u) >> 11, d2 = q - d3 * 10u;      // q / 10, q % 10   (q < 100)
  const u32 d1 = (r * 205u) >> 11, d0 = r - d1 * 10u;
  return 0x30303030u + (d3 << 24) + (d2 << 16) + (d1 << 8) + d0;
}
// canonical decimal of a base element, most significant digit first, no leading zeros;
// ZERO -> "" (zero_as_empty) or "0".  The value is split into 4-digit chunks that are appended as packed
// words, skipping the leading zero characters of the zero-padded string.
template <class F, class S> MS_HD void put_dec(S& s, typename F::T v_, int zero_as_empty) {
  const u64 v = F::to_u64(v_);
  if (v == 0) { if (!zero_as_empty) s.append('0', 1); return; }
  constexpr int NCH = (F::MAX_DIGITS + 3) / 4;  // 5 (Goldilocks, 20 digits) / 3 (BabyBear, 10 digits -> 12 chars)
  u32 c[NCH];
  if (NCH == 5) {
    const u64 hi = v / 100000000ULL; const u32 lo = (u32)(v - hi * 100000000ULL);       // hi < 1.85e11
    const u32 hi2 = (u32)(hi / 100000000ULL); const u32 mid = (u32)(hi - (u64)hi2 * 100000000ULL);
    c[0] = hi2; c[1] = mid / 10000u; c[2] = mid - c[1] * 10000u;
    c[NCH - 2] = lo / 10000u; c[NCH - 1] = lo - c[NCH - 2] * 10000u;
  } else {
    const u32 x = (u32)v;
    c[0] = x / 100000000u; const u32 r = x - c[0] * 100000000u;
    c[1] = r / 10000u; c[NCH - 1] = r - c[1] * 10000u;
  }
  u32 lead = 0; bool found = false;  // leading zero characters
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const u32 z = c[j] == 0 ? 4u : (c[j] < 10u ? 3u : (c[j] < 100u ? 2u : (c[j] < 1000u ? 1u : 0u)));
    if (!found) lead += z;
    found = found || (c[j] != 0);
  }
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const int k = 4 * (j + 1) - (int)lead;
    if (k >= 4) s.append4(pack4(c[j]));             // every chunk after the leading one
    else if (k > 0) s.append(pack4(c[j]), (u32)k);  // leading chunk: 1..3 digits
  }
}
template <class F, int E> struct Display {
  template <class S> static MS_HD void put(S& s, const typename F::T* c, int zae) {
    s.append4(0x51756164u); s.append4(0x45787446u); s.append4(0x69656c64u); s.append('(', 1);     // "QuadExtField("
    Display<F, E / 2>::put(s, c, zae);
    s.append(0x202b20u, 3);                                                                          // " + "
    Display<F, E / 2>::put(s, c + E / 2, zae);
    s.append4(0x202a2075u); s.append(')', 1);                                                        // " * u)"
  }
};
template <class F> struct Display<F, 1> {
  template <class S> static MS_HD void put(S& s, const typename F::T* c, int zae) { put_dec<F>(s, c[0], zae); }
};

// Leaf-group hashing.  Element f of the committed vector lives at
//   base + (f % width) * col_stride + (f / width) * row_stride + limb * limb_stride
// (row-major trace: width=1,row_stride=1; column-major LDE: width=c,col_stride=L;
//  FRI codeword: width=1, limb_stride=D).
template <class F, int E> struct LeafHashKernel {
  typedef typename F::T T;
  static constexpr int THREADS = msmerkle::THREADS;
  struct Params {
    const T* base; size_t col_stride, row_stride, limb_stride;
    u32 width, lpn; int zero_as_empty;
    size_t ngroups;
    u32* nodes;  // 8 words per digest, standard byte order in memory
  };
  // ring words: 63 leftover bytes + the longest element string must fit
  static constexpr int ELEM_MAX = (E == 1) ? F::MAX_DIGITS : (E == 2 ? 21 + 2 * F::MAX_DIGITS : 63 + 4 * F::MAX_DIGITS);
  static constexpr int RW = (63 + ELEM_MAX + 3) / 4 <= 32 ? 32 : 64;
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_HD size_t lds_bytes() { return (size_t)RW * THREADS * sizeof(u32); }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char* lds) {
    const size_t g = (size_t)bx * nthreads + tid;
    if (g >= p.ngroups) return;
    ShaStream<RW> s; s.init(reinterpret_cast<u32*>(lds), nthreads, tid);
    size_t f = g * p.lpn;
    size_t row = f / p.width; u32 col = (u32)(f - row * p.width);
    for (u32 i = 0; i <= p.lpn; i++) {
      if (i < p.lpn) {
        T c[E];
        const T* ptr = p.base + (size_t)col * p.col_stride + row * p.row_stride;
#pragma unroll
        for (int k = 0; k < E; k++) c[k] = ptr[(size_t)k * p.limb_stride];
        Display<F, E>::put(s, c, p.zero_as_empty);
        if (++col == p.width) { col = 0; row++; }
      } else {
        s.pad();
      }
      s.drain();  // the only compression site
    }
    uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + g * 8);
    uint4_t o0, o1;
    o0.x = bswap32(s.h.st[0]); o0.y = bswap32(s.h.st[1]); o0.z = bswap32(s.h.st[2]); o0.w = bswap32(s.h.st[3]);
    o1.x = bswap32(s.h.st[4]); o1.y = bswap32(s.h.st[5]); o1.z = bswap32(s.h.st[6]); o1.w = bswap32(s.h.st[7]);
    out[0] = o0; out[1] = o1;
  }
};

// Inner levels.  `nlevels` consecutive levels are processed by the launch: with
// nlevels > 1 the grid must be a single workgroup (fused tree top).
// IC > 0: inner_children fixed at compile time (IC = 2 is the prover's tree, starks.rs:290-301: the padding block's
// message schedule then folds to literals); IC = 0: taken from Params.
struct InnerHashParams { u32* nodes; size_t child_off, nchildren; u32 ic; u32 nlevels; };
template <int IC> struct InnerHashKernelT {
  static constexpr int THREADS = msmerkle::THREADS;
  typedef InnerHashParams Params;
  static MS_HD int nphases(const Params& p) { return (int)p.nlevels; }
  static MS_DEV void phase(int ph, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const u32 ic = IC ? (u32)IC : p.ic;
    size_t child_off = p.child_off, nchildren = p.nchildren;
    for (int l = 0; l < ph; l++) { child_off += nchildren; nchildren /= ic; }
    const size_t nparents = nchildren / ic;
    const size_t stride = (p.nlevels > 1) ? (size_t)nthreads : 0;
    for (size_t g = (size_t)bx * nthreads + tid; g < nparents; g += stride) {
      const u32* ch = p.nodes + (child_off + g * ic) * 8;
      Sha256 h; h.init();
      u32 w[16];
      for (u32 b = 0; b < ic / 2; b++) {
        // two children = one 64-byte block, fetched as four 16-byte loads
        const uint4_t* c4 = reinterpret_cast<const uint4_t*>(ch + b * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint4_t v = c4[q];
          w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
        }
        h.compress(w);
      }
      // padding block: 0x80, zeros, bit length.  With the constant length of the binary tree the whole
      // message schedule of this block folds to literals at compile time.
      w[0] = 0x80000000u;
#pragma unroll
      for (int i = 1; i < 15; i++) w[i] = 0;
      w[15] = ic * 256u;  // message bits (ic * 32 bytes)
      h.compress(w);
      uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + (child_off + nchildren + g) * 8);
      uint4_t o0, o1;
      o0.x = bswap32(h.st[0]); o0.y = bswap32(h.st[1]); o0.z = bswap32(h.st[2]); o0.w = bswap32(h.st[3]);
      o1.x = bswap32(h.st[4]); o1.y = bswap32(h.st[5]); o1.z = bswap32(h.st[6]); o1.w = bswap32(h.st[7]);
      out[0] = o0; out[1] = o1;
      if (stride == 0) break;
    }
  }
};

typedef InnerHashKernelT<0> InnerHashKernel;
typedef InnerHashKernelT<2> InnerHashKernel2;

// MerklePath extraction (src/merkle.rs:216-288), one thread per opened leaf.  Each job names a
// tree (FRI codeword view, width = 1) and the device word holding the leaf index; writes
//   u64 leaf_index | lpn*E u64 limbs | u64 nlevels | nlevels * ic * 32 bytes      at out.
template <class F, int E> struct PathJob {
  const typename F::T* leafs; size_t limb_stride;
  const u32* nodes; size_t leaf_num; u32 lpn, ic, nlevels /* levels-1 */;
  const unsigned long long* idx;
  unsigned char* out;
};
template <class F, int E> struct PathKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params { const PathJob<F, E>* jobs; u32 njobs; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& pp, int bx, int, int tid, int nthreads, unsigned char*) {
    const u32 t = (u32)bx * nthreads + tid;
    if (t >= pp.njobs) return;
    const PathJob<F, E>& p = pp.jobs[t];
    u64* o = reinterpret_cast<u64*>(p.out);
    const size_t li = (size_t)*p.idx;
    if (li >= p.leaf_num) return;  // value not found: the host reports MS_ERR_LEAF_NOT_FOUND
    *o++ = li;
    const size_t start = li - li % p.lpn;  // merkle.rs:230-236
    for (u32 i = 0; i < p.lpn; i++)
      for (int k = 0; k < E; k++) *o++ = F::to_u64(p.leafs[(size_t)k * p.limb_stride + start + i]);
    *o++ = p.nlevels;
    u32* o32 = reinterpret_cast<u32*>(o);
    size_t cur = li / p.lpn;       // index inside the current level
    size_t level_off = 0, level_n = p.leaf_num / p.lpn;
    for (u32 l = 0; l < p.nlevels; l++) {  // merkle.rs:241-265
      const size_t s = cur - cur % p.ic;
      const u32* src = p.nodes + (level_off + s) * 8;
      for (u32 i = 0; i < p.ic * 8; i++) *o32++ = src[i];
      level_off += level_n; level_n /= p.ic; cur /= p.ic;
    }
  }
};

}  // namespace msmerkle
