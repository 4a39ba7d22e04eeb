// oracle/ministark_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference prover's
// low-degree-extension + FRI + Merkle hot path (alv-around/mini-stark), used
// as the parity checker for the HIP backend and as the timed "port" CPU
// baseline.  Nothing under mini-stark_amd/ may link, import or call this file;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY STATUS (see DESIGN.md "Oracle"):
//   * pinned by the reference's own fixtures: util KATs (src/util.rs:50-96),
//     query-count KATs (src/starks.rs:349-374), Merkle node counts / parent
//     indices / path lengths (src/merkle.rs:399-481) and the SHA-256 tree
//     roots printed by the reference's scripts/merkle_tree.py
//     (tests/golden/merkle_script_roots.json);
//   * mathematically pinned: NTT / INTT / coset-LDE / fold / DEEP / query
//     quotients (exact field arithmetic with the roots of unity arkworks
//     derives from src/field.rs:44-45,73-74);
//   * PARITY UNPINNED: the arkworks `Display` strings that feed the leaf
//     hashes ("" for zero, "QuadExtField(a + b * u)") are restated from memory
//     of ark-ff 0.5.0 (the crate source is not in this container; Cargo.lock
//     pins ark-ff 0.5.0 / ark-poly 0.5.0), and the nimue transcript is not
//     restated at all (challenges are inputs).
//
// Every function cites the reference file:line it follows.
//
// Build: see oracle/Makefile  (g++ -O3 -shared -fPIC).

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

// Threads for the embarrassingly parallel loops (leaf groups, tree levels, columns, limbs).  Default 1: the
// reference is single-threaded (README.md:33 lists rayon as not done).  or_set_threads(n) is used only by the
// bench's "all cores" CPU baseline.
static int g_threads = 1;

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------
// status codes (mirror include/ministark.h; src/error.rs:13-21 + panics)
// ---------------------------------------------------------------------------
enum {
  OR_OK = 0,
  OR_ERR_SHAPE = -1,           // reference assert!/panic conditions
  OR_ERR_LEAF_NOT_FOUND = -2,  // src/error.rs:15-18
  OR_ERR_OUT_OF_RANGE = -3,    // src/error.rs:19-21
  OR_ERR_STATE = -4,
  OR_ERR_ARG = -5,
};

// ---------------------------------------------------------------------------
// src/util.rs:4-44
// ---------------------------------------------------------------------------
static bool is_power_of_two(u64 n) {  // util.rs:4-14 (0 counts as a power)
  return (n & (n - 1)) == 0;
}
static int ctz64(u64 x) { return x ? __builtin_ctzll(x) : 64; }
// util.rs:16-28 ; returns -1 "not a power of 2", -2 "not a power of base"
static long logarithm_of_two_k(u64 number, u64 base) {
  int log_n = ctz64(base);
  if (!is_power_of_two(number)) return -1;
  int p2 = ctz64(number);
  if (p2 % log_n != 0) return -2;
  return p2 / log_n;
}
// util.rs:30-44
static u64 ceil_log2_k(u64 number, u64 base) {
  if (number == 1) return 1;
  u64 log2_base = ctz64(base);
  u64 log2_number = ctz64(number);
  if (is_power_of_two(number) && log2_number % log2_base == 0) return log2_number;
  u64 next_power_2 = 64 - __builtin_clzll(number);
  return ((next_power_2 + log2_base - 1) / log2_base) * log2_base;
}

// ---------------------------------------------------------------------------
// SHA-256 (FIPS 180-4) — the only digest the reference's tests use
// (tests/e2e_goldilocks.rs:6, src/merkle.rs:345)
// ---------------------------------------------------------------------------
static const u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_compress(u32 st[8], const u8 blk[64]) {
  u32 w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((u32)blk[4 * i] << 24) | ((u32)blk[4 * i + 1] << 16) | ((u32)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    u32 s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    u32 s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i++) {
    u32 S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + K256[i] + w[i];
    u32 S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

static void sha256(const u8* msg, size_t len, u8 out[32]) {
  u32 st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t i = 0;
  for (; i + 64 <= len; i += 64) sha256_compress(st, msg + i);
  u8 tail[128];
  size_t rem = len - i;
  memcpy(tail, msg + i, rem);
  tail[rem] = 0x80;
  size_t tl = (rem + 9 <= 64) ? 64 : 128;
  memset(tail + rem + 1, 0, tl - rem - 1);
  u64 bits = (u64)len * 8;
  for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (u8)(bits >> (8 * k));
  sha256_compress(st, tail);
  if (tl == 128) sha256_compress(st, tail + 64);
  for (int k = 0; k < 8; k++) {
    out[4 * k] = st[k] >> 24; out[4 * k + 1] = st[k] >> 16; out[4 * k + 2] = st[k] >> 8; out[4 * k + 3] = st[k];
  }
}

struct Digest { u8 b[32]; };

// ---------------------------------------------------------------------------
// src/field.rs:36-109 — base fields (canonical representation)
// ---------------------------------------------------------------------------
struct GL {  // field.rs:43-47
  static constexpr u64 P = 0xFFFFFFFF00000001ULL;
  static constexpr u64 GENERATOR = 7;     // field.rs:45
  static constexpr int TWO_ADICITY = 32;  // p-1 = 2^32 * (2^32-1)
  static constexpr int ID = 0;
  static inline u64 add(u64 a, u64 b) { u64 s = a + b; if (s < a || s >= P) s -= P; return s; }
  static inline u64 sub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
  // One 64x64 -> 128 multiplication and the special-form reduction 2^64 == 2^32 - 1, 2^96 == -1 (mod p): at least as fast as
  // the one-limb Montgomery multiplication arkworks' MontBackend performs at this call site (a `% P` on u128 is a
  // library call and flattered the GPU/CPU ratio).  Exact: same canonical result.
  static inline u64 mul(u64 a, u64 b) {
    const u128 t = (u128)a * b;
    const u64 lo = (u64)t, hi = (u64)(t >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFULL;            // borrow: - 2^64 == - (2^32 - 1)
    const u64 t1 = hl * 0xFFFFFFFFULL;
    u64 r = t0 + t1;
    if (r < t1 || r >= P) r -= P;                // carry (+ 2^64 == + 2^32 - 1 == - P mod 2^64) or plain >= p
    return r;
  }
};
struct BB {  // field.rs:72-76
  static constexpr u64 P = 2013265921ULL;
  static constexpr u64 GENERATOR = 440564289ULL;  // field.rs:74 (quirk Q8: order 2^27 element)
  static constexpr int TWO_ADICITY = 27;          // p-1 = 2^27 * 15
  static constexpr int ID = 1;
  static inline u64 add(u64 a, u64 b) { u64 s = a + b; if (s >= P) s -= P; return s; }
  static inline u64 sub(u64 a, u64 b) { return a >= b ? a - b : a + P - b; }
  static inline u64 mul(u64 a, u64 b) { return (a * b) % P; }
};

template <class F> static u64 f_pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) { if (e & 1) r = F::mul(r, a); a = F::mul(a, a); e >>= 1; }
  return r;
}
template <class F> static u64 f_inv(u64 a) { return f_pow<F>(a, F::P - 2); }

// [ark-mem] ark-ff 0.5 MontConfig derive: TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((p-1)/2^s);
// FftField::get_root_of_unity(n): square it (s - log2 n) times.  Used by
// Radix2EvaluationDomain::new (air.rs:74, starks.rs:82, fri.rs:315).
template <class F> static u64 root_of_unity(u64 n) {
  int k = ctz64(n);
  u64 w = f_pow<F>(F::GENERATOR, (F::P - 1) >> F::TWO_ADICITY);
  for (int i = k; i < F::TWO_ADICITY; i++) w = F::mul(w, w);
  return w;
}

// ---------------------------------------------------------------------------
// extension towers.  Elements are E consecutive base limbs:
//   E=1: base;  E=2: c0 + c1*u  (u^2 = NR);  E=4 (BabyBear): (a0+a1 u) + (a2+a3 u) v,
//   v^2 = (2013265910 + u)  (field.rs:50-62, 78-109).
// ---------------------------------------------------------------------------
template <class F, int E> struct Ext;

template <class F> struct Ext<F, 1> {
  u64 c[1];
  static constexpr int DEG = 1;
};
template <class F> struct Ext<F, 2> { u64 c[2]; static constexpr int DEG = 2; };
template <class F> struct Ext<F, 4> { u64 c[4]; static constexpr int DEG = 4; };

template <class F> static inline u64 nr2();  // quadratic non-residue
template <> inline u64 nr2<GL>() { return 7; }    // field.rs:55
template <> inline u64 nr2<BB>() { return 11; }   // field.rs:84

template <class F, int E> static inline Ext<F, E> e_zero() { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = 0; return r; }
template <class F, int E> static inline Ext<F, E> e_from_base(u64 b) { Ext<F, E> r = e_zero<F, E>(); r.c[0] = b; return r; }
template <class F, int E> static inline Ext<F, E> e_one() { return e_from_base<F, E>(1); }
template <class F, int E> static inline bool e_is_zero(const Ext<F, E>& a) { for (int i = 0; i < E; i++) if (a.c[i]) return false; return true; }
template <class F, int E> static inline bool e_eq(const Ext<F, E>& a, const Ext<F, E>& b) { for (int i = 0; i < E; i++) if (a.c[i] != b.c[i]) return false; return true; }
template <class F, int E> static inline Ext<F, E> e_add(const Ext<F, E>& a, const Ext<F, E>& b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::add(a.c[i], b.c[i]); return r; }
template <class F, int E> static inline Ext<F, E> e_sub(const Ext<F, E>& a, const Ext<F, E>& b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::sub(a.c[i], b.c[i]); return r; }
template <class F, int E> static inline Ext<F, E> e_neg(const Ext<F, E>& a) { return e_sub<F, E>(e_zero<F, E>(), a); }
template <class F, int E> static inline Ext<F, E> e_mul_base(const Ext<F, E>& a, u64 b) { Ext<F, E> r; for (int i = 0; i < E; i++) r.c[i] = F::mul(a.c[i], b); return r; }

template <class F> static inline Ext<F, 1> e_mul(const Ext<F, 1>& a, const Ext<F, 1>& b) { Ext<F, 1> r; r.c[0] = F::mul(a.c[0], b.c[0]); return r; }
template <class F> static inline Ext<F, 2> e_mul(const Ext<F, 2>& a, const Ext<F, 2>& b) {
  Ext<F, 2> r;
  r.c[0] = F::add(F::mul(a.c[0], b.c[0]), F::mul(nr2<F>(), F::mul(a.c[1], b.c[1])));
  r.c[1] = F::add(F::mul(a.c[0], b.c[1]), F::mul(a.c[1], b.c[0]));
  return r;
}
// Fp4 = Fp2[v]/(v^2 - (NRc0 + NRc1*u)); BabyBear: NR = (2013265910, 1)  (field.rs:98)
template <class F> static inline Ext<F, 2> nr4();
template <> inline Ext<BB, 2> nr4<BB>() { Ext<BB, 2> r; r.c[0] = 2013265910ULL; r.c[1] = 1; return r; }
template <> inline Ext<GL, 2> nr4<GL>() { Ext<GL, 2> r; r.c[0] = 0; r.c[1] = 1; return r; }  // unused by the reference
template <class F> static inline Ext<F, 4> e_mul(const Ext<F, 4>& a, const Ext<F, 4>& b) {
  Ext<F, 2> a0{{a.c[0], a.c[1]}}, a1{{a.c[2], a.c[3]}}, b0{{b.c[0], b.c[1]}}, b1{{b.c[2], b.c[3]}};
  Ext<F, 2> r0 = e_add<F, 2>(e_mul<F>(a0, b0), e_mul<F>(nr4<F>(), e_mul<F>(a1, b1)));
  Ext<F, 2> r1 = e_add<F, 2>(e_mul<F>(a0, b1), e_mul<F>(a1, b0));
  Ext<F, 4> r; r.c[0] = r0.c[0]; r.c[1] = r0.c[1]; r.c[2] = r1.c[0]; r.c[3] = r1.c[1];
  return r;
}
template <class F> static inline Ext<F, 1> e_inv(const Ext<F, 1>& a) { Ext<F, 1> r; r.c[0] = f_inv<F>(a.c[0]); return r; }
template <class F> static inline Ext<F, 2> e_inv(const Ext<F, 2>& a) {
  // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 - NR a1^2)
  u64 n = F::sub(F::mul(a.c[0], a.c[0]), F::mul(nr2<F>(), F::mul(a.c[1], a.c[1])));
  u64 ni = f_inv<F>(n);
  Ext<F, 2> r; r.c[0] = F::mul(a.c[0], ni); r.c[1] = F::mul(F::sub(0, a.c[1]), ni);
  return r;
}
template <class F> static inline Ext<F, 4> e_inv(const Ext<F, 4>& a) {
  Ext<F, 2> a0{{a.c[0], a.c[1]}}, a1{{a.c[2], a.c[3]}};
  Ext<F, 2> n = e_sub<F, 2>(e_mul<F>(a0, a0), e_mul<F>(nr4<F>(), e_mul<F>(a1, a1)));
  Ext<F, 2> ni = e_inv<F>(n);
  Ext<F, 2> r0 = e_mul<F>(a0, ni), r1 = e_mul<F>(e_neg<F, 2>(a1), ni);
  Ext<F, 4> r; r.c[0] = r0.c[0]; r.c[1] = r0.c[1]; r.c[2] = r1.c[0]; r.c[3] = r1.c[1];
  return r;
}

// ---------------------------------------------------------------------------
// [ark-mem] Display (src/merkle.rs:162-168 hashes `child.to_string()`):
//   Fp:  `self.into_bigint().to_string().trim_start_matches('0')` — canonical
//        decimal; ZERO prints as the EMPTY string (zero_as_empty=1).  With
//        zero_as_empty=0 zero prints "0" (the format scripts/merkle_tree.py uses).
//   QuadExtField: "QuadExtField({c0} + {c1} * u)", nested for Fp4.
// ---------------------------------------------------------------------------
static inline void append_dec(std::string& s, u64 v, int zero_as_empty) {
  if (v == 0) { if (!zero_as_empty) s.push_back('0'); return; }
  char buf[24]; int n = 0;
  while (v) { buf[n++] = '0' + (v % 10); v /= 10; }
  while (n) s.push_back(buf[--n]);
}
static void append_display(std::string& s, const u64* c, int E, int zero_as_empty) {
  if (E == 1) { append_dec(s, c[0], zero_as_empty); return; }
  s += "QuadExtField(";
  append_display(s, c, E / 2, zero_as_empty);
  s += " + ";
  append_display(s, c + E / 2, E / 2, zero_as_empty);
  s += " * u)";
}

// ---------------------------------------------------------------------------
// src/merkle.rs:81-289 — MerkleTree (generic over element width E limbs)
// ---------------------------------------------------------------------------
struct MerkleTree {
  int E = 1;
  size_t lpn = 0, ic = 0, levels = 0;
  int zero_as_empty = 1;
  std::vector<u64> leafs;     // leaf_num * E limbs  (merkle.rs:143 copies the inputs)
  std::vector<Digest> nodes;  // level-major, root last (merkle.rs:119-140)
  size_t leaf_num() const { return leafs.size() / E; }
  size_t node_number() const { return leaf_num() + nodes.size(); }  // merkle.rs:157-159

  // merkle.rs:162-168
  Digest from_leafs(const u64* group, size_t n) const {
    std::string s;
    for (size_t i = 0; i < n; i++) append_display(s, group + i * E, E, zero_as_empty);
    Digest d; sha256((const u8*)s.data(), s.size(), d.b); return d;
  }
  // merkle.rs:171-177
  static Digest from_nodes(const Digest* ch, size_t n) {
    Digest d; sha256((const u8*)ch, n * 32, d.b); return d;
  }

  // merkle.rs:81-148
  int build(const u64* inputs, size_t leaf_num_, int E_, size_t lpn_, size_t ic_, int zae) {
    E = E_; lpn = lpn_; ic = ic_; zero_as_empty = zae;
    if (lpn == 0 || ic < 2 || !is_power_of_two(ic)) return OR_ERR_SHAPE;  // util.rs:17 assert
    size_t leaf_num = leaf_num_;
    size_t node_num = leaf_num / lpn;
    long lg = logarithm_of_two_k(node_num, ic);  // merkle.rs:93-96
    if (lg < 0) return OR_ERR_SHAPE;
    levels = (size_t)lg + 1;
    if (leaf_num % lpn != 0) return OR_ERR_SHAPE;  // merkle.rs:99
    if (levels - 1 >= 64) return OR_ERR_SHAPE;     // empty input: pow overflows in the reference
    {
      u64 pw = 1;
      for (size_t i = 0; i + 1 < levels; i++) pw *= ic;
      if (pw != leaf_num / lpn) return OR_ERR_SHAPE;  // merkle.rs:100-104
    }
    // merkle.rs:116-118 geometric series
    size_t total = 0; { size_t m = node_num; for (;;) { total += m; if (m == 1) break; m /= ic; } }
    nodes.clear();
    nodes.resize(total);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (size_t g = 0; g < node_num; g++) nodes[g] = from_leafs(inputs + g * lpn * E, lpn);  // merkle.rs:124-128
    // merkle.rs:131-140 (each parent hashes inner_children consecutive earlier nodes), level by level
    for (size_t child0 = 0, m = node_num; m > 1; child0 += m, m /= ic) {
      const size_t parent0 = child0 + m;
#pragma omp parallel for num_threads(g_threads) schedule(static)
      for (size_t j = 0; j < m / ic; j++) nodes[parent0 + j] = from_nodes(&nodes[child0 + j * ic], ic);
    }
    leafs.assign(inputs, inputs + leaf_num * E);
    return OR_OK;
  }
  Digest root() const { return nodes.back(); }  // merkle.rs:151-154

  // merkle.rs:188-207
  int get_parent_idx(size_t index, size_t* out) const {
    size_t root_idx = node_number() - 1;
    if (index >= root_idx) return OR_ERR_OUT_OF_RANGE;
    if (index < leaf_num()) *out = leaf_num() + index / lpn;
    else *out = index + (node_number() - index + 1) / ic;
    return OR_OK;
  }
  // merkle.rs:216-225 (first match wins, quirk Q7)
  int get_leaf_index(const u64* leaf, size_t* out) const {
    size_t n = leaf_num();
    for (size_t i = 0; i < n; i++) {
      bool eq = true;
      for (int k = 0; k < E; k++) if (leafs[i * E + k] != leaf[k]) { eq = false; break; }
      if (eq) { *out = i; return OR_OK; }
    }
    return OR_ERR_LEAF_NOT_FOUND;
  }
  // merkle.rs:272-288 (+230-265).  Serialises a MerklePath:
  //   u64 leaf_index | lpn*E limbs (leaf_neighbours) | u64 nlevels | nlevels * ic * 32 bytes
  int open_index(size_t leaf_index, std::vector<u8>& out) const {
    auto put64 = [&](u64 v) { for (int k = 0; k < 8; k++) out.push_back((u8)(v >> (8 * k))); };
    put64(leaf_index);
    size_t start = leaf_index - leaf_index % lpn;  // merkle.rs:230-236
    for (size_t i = start * E; i < (start + lpn) * E; i++) put64(leafs[i]);
    size_t cur;
    int rc = get_parent_idx(leaf_index, &cur);  // merkle.rs:276
    if (rc) return rc;
    put64(levels - 1);
    for (size_t l = 1; l < levels; l++) {  // merkle.rs:253-265
      size_t shifted = cur - leaf_num();   // merkle.rs:241-248
      size_t s = shifted - shifted % ic;
      const u8* p = (const u8*)&nodes[s];
      out.insert(out.end(), p, p + ic * 32);
      rc = get_parent_idx(cur, &cur);
      if (rc) return rc;
    }
    return OR_OK;
  }
  int generate_proof(const u64* leaf, std::vector<u8>& out) const {
    size_t idx;
    int rc = get_leaf_index(leaf, &idx);
    if (rc) return rc;
    return open_index(idx, out);
  }
};

// merkle.rs:312-338  MerkleRoot::check_proof on a serialised path (see open_index)
static int merkle_check_proof(const u8 root[32], const u8* path, size_t path_len, int E, size_t lpn, size_t ic, int zae) {
  MerkleTree t; t.E = E; t.zero_as_empty = zae;
  if (path_len < 8 + lpn * E * 8 + 8) return 0;
  const u64* neigh = (const u64*)(path + 8);
  Digest prev = t.from_leafs(neigh, lpn);
  u64 nlev; memcpy(&nlev, path + 8 + lpn * E * 8, 8);
  const u8* p = path + 16 + lpn * E * 8;
  if (path_len != 16 + lpn * E * 8 + nlev * ic * 32) return 0;
  for (u64 l = 0; l < nlev; l++) {
    bool found = false;
    for (size_t k = 0; k < ic; k++) if (!memcmp(p + k * 32, prev.b, 32)) found = true;
    if (!found) return 0;
    prev = MerkleTree::from_nodes((const Digest*)p, ic);
    p += ic * 32;
  }
  return memcmp(prev.b, root, 32) == 0;
}

// ---------------------------------------------------------------------------
// [ark-mem] ark-poly 0.5 Radix2EvaluationDomain fft/ifft: natural order in and
// out, ifft scales by 1/n; coset fft distributes offset^k first.  Restated as
// a textbook bit-reverse + iterative DIT radix-2 (values are unique: exact
// field arithmetic).  `stride` walks one limb of an AoS extension vector.
// ---------------------------------------------------------------------------
template <class F> struct NttPlan {
  size_t n; std::vector<u64> tw;  // tw[j] = w^j, j < n/2
  NttPlan(size_t n_, bool inverse) : n(n_), tw(n_ / 2 ? n_ / 2 : 1) {
    u64 w = root_of_unity<F>(n);
    if (inverse) w = f_inv<F>(w);
    u64 x = 1;
    for (size_t j = 0; j < n / 2; j++) { tw[j] = x; x = F::mul(x, w); }
    if (n < 2) tw[0] = 1;
  }
};
template <class F> static void ntt_inplace(const NttPlan<F>& pl, u64* a, size_t stride) {
  const size_t n = pl.n;
  const int lg = ctz64(n);
  // the OpenMP clauses only take effect for the all-cores baseline of large single transforms (or_set_threads > 1 and not
  // already inside the per-column parallel loop: nested regions run on one thread)
  const bool par = g_threads > 1 && n >= 8192;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
  for (size_t i = 0; i < n; i++) {
    size_t j = 0;
    for (int b = 0; b < lg; b++) j |= ((i >> b) & 1) << (lg - 1 - b);
    if (i < j) { u64 t = a[i * stride]; a[i * stride] = a[j * stride]; a[j * stride] = t; }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len >> 1, step = n / len;
    // n/2 independent butterflies per stage, indexed b = (block, j)
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
    for (size_t b = 0; b < n / 2; b++) {
      const size_t j = b & (half - 1), i = (b - j) * 2;
      u64 u = a[(i + j) * stride];
      u64 v = F::mul(a[(i + j + half) * stride], pl.tw[j * step]);
      a[(i + j) * stride] = F::add(u, v);
      a[(i + j + half) * stride] = F::sub(u, v);
    }
  }
}

// ---------------------------------------------------------------------------
// DensePolynomial helpers ([ark-mem] semantics: from_coefficients_vec trims
// trailing zeros; degree() of zero = 0; evaluate = Horner; `/` = long division
// keeping the quotient).  Coefficients are AoS vectors of Ext<F,E>.
// ---------------------------------------------------------------------------
template <class F, int E> using Poly = std::vector<Ext<F, E>>;
template <class F, int E> static void p_trim(Poly<F, E>& p) { while (!p.empty() && e_is_zero<F, E>(p.back())) p.pop_back(); }
template <class F, int E> static size_t p_degree(const Poly<F, E>& p) { return p.empty() ? 0 : p.size() - 1; }
template <class F, int E> static Ext<F, E> p_eval(const Poly<F, E>& p, const Ext<F, E>& x) {
  Ext<F, E> acc = e_zero<F, E>();
  for (size_t i = p.size(); i-- > 0;) acc = e_add<F, E>(e_mul<F>(acc, x), p[i]);
  return acc;
}
template <class F, int E> static Poly<F, E> p_sub(const Poly<F, E>& a, const Poly<F, E>& b) {
  Poly<F, E> r(std::max(a.size(), b.size()), e_zero<F, E>());
  for (size_t i = 0; i < a.size(); i++) r[i] = a[i];
  for (size_t i = 0; i < b.size(); i++) r[i] = e_sub<F, E>(r[i], b[i]);
  p_trim<F, E>(r);
  return r;
}
// long division by a MONIC divisor (the reference only divides by (x - z) and
// (x - x1)(x - x2): fri.rs:91,101,165-166); returns the quotient.
template <class F, int E> static Poly<F, E> p_div_monic(const Poly<F, E>& num, const Poly<F, E>& den) {
  Poly<F, E> q;
  if (num.empty() || num.size() < den.size()) return q;
  Poly<F, E> r = num;
  size_t dd = den.size() - 1;
  q.assign(num.size() - dd, e_zero<F, E>());
  for (size_t i = num.size(); i-- > dd;) {
    Ext<F, E> c = r[i];
    q[i - dd] = c;
    if (e_is_zero<F, E>(c)) continue;
    for (size_t k = 0; k <= dd; k++) r[i - dd + k] = e_sub<F, E>(r[i - dd + k], e_mul<F>(c, den[k]));
  }
  p_trim<F, E>(q);
  return q;
}

// evaluate a (possibly shorter) ext polynomial over Radix2(D) limb by limb —
// the domain points are base-field elements embedded in the extension
// ([ark-mem] FftField for QuadExtField), fri.rs:350.
template <class F, int E> static std::vector<u64> ext_evaluate_over_domain(const Poly<F, E>& p, size_t D) {
  std::vector<u64> ev(D * E, 0);
  for (size_t i = 0; i < p.size() && i < D; i++) for (int k = 0; k < E; k++) ev[i * E + k] = p[i].c[k];
  NttPlan<F> pl(D, false);
#pragma omp parallel for num_threads(g_threads)
  for (int k = 0; k < E; k++) ntt_inplace<F>(pl, ev.data() + k, E);
  return ev;
}

// ---------------------------------------------------------------------------
// src/starks.rs:312-332
// ---------------------------------------------------------------------------
static int num_queries_from_config(int modulus_bits, u64 security_bits, u64 blowup, u64 steps, u64* linking, u64* fri) {
  if (security_bits < 20) return OR_ERR_SHAPE;  // starks.rs:317-320 panics
  u64 log_steps = ceil_log2_k(steps, 2);
  u64 den = (u64)modulus_bits - log_steps;
  *linking = (security_bits + den - 1) / den;
  u64 rounds = ceil_log2_k(steps * blowup, 2);
  double rho = 1.0 / (double)blowup;
  double denominator = __builtin_log2(2.0 / (1.0 + rho));
  double total = (double)security_bits / denominator;
  *fri = (u64)__builtin_ceil(total / (double)rounds);
  return OR_OK;
}

// ---------------------------------------------------------------------------
// The prover session: one object per Stark::prove call, stage functions cut at
// the transcript interactions of src/starks.rs:59-169 and src/fri.rs:64-189.
// ---------------------------------------------------------------------------
struct SessionBase {
  virtual ~SessionBase() {}
  virtual int trace_commit(const u64* trace, size_t N, size_t w, size_t lpn, u8 root[32]) = 0;
  virtual int interpolate() = 0;
  virtual int polys_lincomb(const u64* scalars, const int* idx, int k) = 0;
  virtual int polys_count() = 0;
  virtual int poly_read(int i, u64* out) = 0;
  virtual int lde_commit(size_t blowup, u64 shift, size_t lpn, u8 root[32]) = 0;
  virtual int lde_read(u64* out) = 0;
  virtual int lde_nodes_read(u8* out) = 0;
  virtual int mix(u64 r) = 0;
  virtual int validity_read(u64* out) = 0;
  virtual int eval_ext(const u64* z, int q, u64* out) = 0;
  virtual int verify_ood(u64 r, const u64* z, int q, const u64* evals) = 0;
  virtual int fri_begin(size_t blowup, size_t rounds, u8 root0[32]) = 0;
  virtual int fri_deep(const u64* z, u64* B) = 0;
  virtual int fri_fold_commit(const u64* alpha, u8 root[32]) = 0;
  virtual int fri_round_info(int round, u64* ncoef, u64* D) = 0;
  virtual int fri_round_poly_read(int round, u64* out) = 0;
  virtual int fri_round_codeword_read(int round, u64* out) = 0;
  virtual int fri_query(const u64* betas, int nq) = 0;
  virtual size_t fri_proof_size() = 0;
  virtual int fri_proof_read(u8* out) = 0;
  virtual int ext_degree() = 0;
};

template <class F, int E> struct Session : SessionBase {
  int zae;
  size_t N = 0, w = 0, L = 0, blowup = 0;
  std::vector<u64> trace;              // N*w row-major (air.rs:15-19)
  MerkleTree trace_tree, lde_tree;
  std::vector<std::vector<u64>> polys;  // c polys, N base coeffs each (untrimmed storage; trimmed on use)
  std::vector<u64> lde;                 // L*c row-major (starks.rs:87-91)
  std::vector<u64> validity;            // N base coeffs
  bool have_validity = false;
  struct Round { Poly<F, E> poly; size_t D; MerkleTree tree; Poly<F, E> split[2]; };
  std::vector<Round> rounds;
  size_t fri_rounds = 0, fri_blowup = 0;
  Ext<F, E> cur_z; u64 cur_B[2 * 4]; bool have_deep = false;
  std::vector<u8> fri_proof;

  explicit Session(int z) : zae(z) {}
  int ext_degree() override { return E; }

  // starks.rs:68-73 — commit to the RAW trace matrix (quirk Q4: groups of lpn
  // over the row-major flattening)
  int trace_commit(const u64* t, size_t N_, size_t w_, size_t lpn, u8 root[32]) override {
    if (!N_ || !w_ || !is_power_of_two(N_)) return OR_ERR_SHAPE;  // air.rs:23
    for (size_t i = 0; i < N_ * w_; i++) if (t[i] >= F::P) return OR_ERR_ARG;
    N = N_; w = w_;
    trace.assign(t, t + N * w);
    int rc = trace_tree.build(trace.data(), N * w, 1, lpn, 2, zae);
    if (rc) return rc;
    memcpy(root, trace_tree.root().b, 32);
    polys.clear(); have_validity = false; rounds.clear();
    return OR_OK;
  }
  // air.rs:147-160
  int interpolate() override {
    if (!N) return OR_ERR_STATE;
    polys.assign(w, std::vector<u64>(N));
    NttPlan<F> pl(N, true);
    u64 ninv = f_inv<F>(N % F::P);
#pragma omp parallel for num_threads(g_threads)
    for (size_t c = 0; c < w; c++) {
      for (size_t j = 0; j < N; j++) polys[c][j] = trace[j * w + c];  // air.rs:151-153
      ntt_inplace<F>(pl, polys[c].data(), 1);                         // air.rs:154
      for (size_t j = 0; j < N; j++) polys[c][j] = F::mul(polys[c][j], ninv);
    }
    return OR_OK;
  }
  // user closures of tests/e2e_goldilocks.rs:48-59 are linear combinations of
  // trace polys with scalar coefficients; appended as a new constraint poly.
  int polys_lincomb(const u64* scalars, const int* idx, int k) override {
    if (polys.empty()) return OR_ERR_STATE;
    std::vector<u64> r(N, 0);
    for (int t = 0; t < k; t++) {
      if (idx[t] < 0 || (size_t)idx[t] >= polys.size() || scalars[t] >= F::P) return OR_ERR_ARG;
      const std::vector<u64>& p = polys[idx[t]];
      for (size_t j = 0; j < N; j++) r[j] = F::add(r[j], F::mul(scalars[t], p[j]));
    }
    polys.push_back(r);
    return OR_OK;
  }
  int polys_count() override { return (int)polys.size(); }
  int poly_read(int i, u64* out) override {
    if (i < 0 || (size_t)i >= polys.size()) return OR_ERR_ARG;
    memcpy(out, polys[i].data(), N * 8); return OR_OK;
  }
  // starks.rs:80-95
  int lde_commit(size_t blowup_, u64 shift, size_t lpn, u8 root[32]) override {
    if (polys.empty()) return OR_ERR_STATE;
    if (!blowup_ || !is_power_of_two(blowup_) || shift == 0 || shift >= F::P) return OR_ERR_ARG;
    blowup = blowup_; L = N * blowup;
    if (ctz64(L) > F::TWO_ADICITY) return OR_ERR_SHAPE;  // Radix2EvaluationDomain::new(..).unwrap() (starks.rs:82-83)
    size_t c = polys.size();
    lde.assign(L * c, 0);
    NttPlan<F> pl(L, false);
#pragma omp parallel for num_threads(g_threads)
    for (size_t i = 0; i < c; i++) {
      std::vector<u64> col(L);
      u64 s = 1;
      for (size_t k = 0; k < L; k++) {  // coset fft: coeff_k * shift^k, zero padded
        col[k] = k < N ? F::mul(polys[i][k], s) : 0;
        s = F::mul(s, shift);
      }
      ntt_inplace<F>(pl, col.data(), 1);                        // starks.rs:89
      for (size_t k = 0; k < L; k++) lde[k * c + i] = col[k];   // starks.rs:90 / air.rs:52-58
    }
    int rc = lde_tree.build(lde.data(), L * c, 1, lpn, 2, zae);  // starks.rs:92-93
    if (rc) return rc;
    memcpy(root, lde_tree.root().b, 32);
    return OR_OK;
  }
  int lde_read(u64* out) override { if (lde.empty()) return OR_ERR_STATE; memcpy(out, lde.data(), lde.size() * 8); return OR_OK; }
  int lde_nodes_read(u8* out) override { if (lde.empty()) return OR_ERR_STATE; memcpy(out, lde_tree.nodes.data(), lde_tree.nodes.size() * 32); return OR_OK; }
  // starks.rs:108-119 — quirk Q1: the "validity polynomial" is the REMAINDER of
  // the division by the vanishing polynomial, i.e. the mixed polynomial itself
  // (every constraint poly has <= N coefficients, so the quotient is zero).
  int mix(u64 r) override {
    if (polys.empty()) return OR_ERR_STATE;
    if (r >= F::P) return OR_ERR_ARG;
    validity.assign(N, 0);
    u64 ri = 1;
    for (size_t i = 0; i < polys.size(); i++) {
      for (size_t j = 0; j < N; j++) validity[j] = F::add(validity[j], F::mul(ri, polys[i][j]));
      ri = F::mul(ri, r);
    }
    have_validity = true;
    return OR_OK;
  }
  int validity_read(u64* out) override { if (!have_validity) return OR_ERR_STATE; memcpy(out, validity.data(), N * 8); return OR_OK; }

  Poly<F, E> extend_poly(const std::vector<u64>& b) {  // field.rs:23-32
    Poly<F, E> p(b.size());
    for (size_t i = 0; i < b.size(); i++) p[i] = e_from_base<F, E>(b[i]);
    p_trim<F, E>(p);
    return p;
  }
  // starks.rs:124-151: out[q][0..c) = constrain_queries, out[q][c] = validity_query
  int eval_ext(const u64* z, int q, u64* out) override {
    if (!have_validity) return OR_ERR_STATE;
    size_t c = polys.size();
    for (int t = 0; t < q; t++) {
      Ext<F, E> x; for (int k = 0; k < E; k++) { if (z[t * E + k] >= F::P) return OR_ERR_ARG; x.c[k] = z[t * E + k]; }
      for (size_t i = 0; i <= c; i++) {
        Poly<F, E> p = extend_poly(i < c ? polys[i] : validity);
        Ext<F, E> v = p_eval<F, E>(p, x);
        for (int k = 0; k < E; k++) out[(t * (c + 1) + i) * E + k] = v.c[k];
      }
    }
    return OR_OK;
  }

  // src/starks.rs:204-225 — the DEEP-ALI half of Stark::verify, against THIS session's constraint polynomials
  // (the verifier derives them itself: tests/e2e_goldilocks.rs:101-102).  evals[t] = c constrain_queries then the
  // validity_query.  Returns 1 on accept, 0 on reject.
  int verify_ood(u64 r, const u64* z, int q, const u64* evals) override {
    if (polys.empty()) return OR_ERR_STATE;
    const size_t c = polys.size();
    for (int t = 0; t < q; t++) {
      Ext<F, E> x; for (int k = 0; k < E; k++) x.c[k] = z[t * E + k];
      Poly<F, E> c_x;  // starks.rs:212
      u64 ri = 1;
      for (size_t i = 0; i < c; i++) {
        Poly<F, E> p = extend_poly(polys[i]);
        Ext<F, E> v = p_eval<F, E>(p, x);
        for (int k = 0; k < E; k++) if (v.c[k] != evals[(t * (c + 1) + i) * E + k]) return 0;  // starks.rs:216
        if (c_x.size() < p.size()) c_x.resize(p.size(), e_zero<F, E>());
        for (size_t j = 0; j < p.size(); j++) c_x[j] = e_add<F, E>(c_x[j], e_mul_base<F, E>(p[j], ri));  // starks.rs:217
        ri = F::mul(ri, r);
      }
      p_trim<F, E>(c_x);
      // starks.rs:220-221: divide_by_vanishing_poly returns (quotient, remainder); the name `rest` binds the QUOTIENT,
      // which must be zero, and `quotient` binds the remainder (quirk Q1)
      if (c_x.size() > N) return 0;
      Ext<F, E> ev = p_eval<F, E>(c_x, x);  // starks.rs:223
      for (int k = 0; k < E; k++) if (ev.c[k] != evals[(t * (c + 1) + c) * E + k]) return 0;  // starks.rs:224
    }
    return 1;
  }

  // fri.rs:314-352
  int new_round(Poly<F, E> poly, size_t domain_size) {
    Round r;
    size_t D = 1; while (D < domain_size) D <<= 1;  // Radix2EvaluationDomain::new rounds up
    if (ctz64(D) > F::TWO_ADICITY) return OR_ERR_SHAPE;
    r.D = D;
    for (size_t i = 0; i < poly.size(); i++) r.split[i % 2].push_back(poly[i]);  // fri.rs:329-343
    p_trim<F, E>(r.split[0]); p_trim<F, E>(r.split[1]);
    if (poly.size() > D) return OR_ERR_SHAPE;
    std::vector<u64> ev = ext_evaluate_over_domain<F, E>(poly, D);  // fri.rs:350
    int rc = r.tree.build(ev.data(), D, E, 2, 2, zae);              // fri.rs:351, starks.rs:290-295
    if (rc) return rc;
    r.poly = std::move(poly);
    rounds.push_back(std::move(r));
    return OR_OK;
  }
  // fri.rs:73-82 — round 0 (its root is NOT added to the reference transcript)
  int fri_begin(size_t blowup_, size_t nrounds, u8 root0[32]) override {
    if (!have_validity) return OR_ERR_STATE;
    if (nrounds < 1) return OR_ERR_ARG;
    rounds.clear(); fri_proof.clear(); have_deep = false;
    fri_rounds = nrounds; fri_blowup = blowup_;
    Poly<F, E> p = extend_poly(validity);               // starks.rs:132
    size_t dsize = (p_degree<F, E>(p) + 1) * blowup_;   // fri.rs:74 (quirk Q11)
    int rc = new_round(std::move(p), dsize);
    if (rc) return rc;
    memcpy(root0, rounds[0].tree.root().b, 32);
    return OR_OK;
  }
  // fri.rs:89-94
  int fri_deep(const u64* z, u64* B) override {
    if (rounds.empty() || rounds.size() >= fri_rounds) return OR_ERR_STATE;
    for (int k = 0; k < E; k++) { if (z[k] >= F::P) return OR_ERR_ARG; cur_z.c[k] = z[k]; }
    Round& pr = rounds.back();
    for (int s = 0; s < 2; s++) {  // fri.rs:354-359
      Ext<F, E> v = p_eval<F, E>(pr.split[s], cur_z);
      for (int k = 0; k < E; k++) { cur_B[s * E + k] = v.c[k]; B[s * E + k] = v.c[k]; }
    }
    have_deep = true;
    return OR_OK;
  }
  // fri.rs:96-109
  int fri_fold_commit(const u64* alpha, u8 root[32]) override {
    if (!have_deep) return OR_ERR_STATE;
    Ext<F, E> a; for (int k = 0; k < E; k++) { if (alpha[k] >= F::P) return OR_ERR_ARG; a.c[k] = alpha[k]; }
    Round& pr = rounds.back();
    // fri.rs:361-372 fold_poly = even + alpha * odd
    Poly<F, E> folded(std::max(pr.split[0].size(), pr.split[1].size()), e_zero<F, E>());
    for (size_t i = 0; i < pr.split[0].size(); i++) folded[i] = pr.split[0][i];
    for (size_t i = 0; i < pr.split[1].size(); i++) folded[i] = e_add<F, E>(folded[i], e_mul<F>(a, pr.split[1][i]));
    p_trim<F, E>(folded);
    // fri.rs:99-100 deep_value = B(alpha)
    Ext<F, E> b0, b1; for (int k = 0; k < E; k++) { b0.c[k] = cur_B[k]; b1.c[k] = cur_B[E + k]; }
    Ext<F, E> deep_value = e_add<F, E>(b0, e_mul<F>(b1, a));
    Poly<F, E> dv{deep_value}; p_trim<F, E>(dv);
    Poly<F, E> den{e_neg<F, E>(cur_z), e_one<F, E>()};  // fri.rs:91
    Poly<F, E> round_poly = p_div_monic<F, E>(p_sub<F, E>(folded, dv), den);  // fri.rs:101
    size_t dsize = pr.D / 2;  // fri.rs:104, 374-376
    int rc = new_round(std::move(round_poly), dsize);
    if (rc) return rc;
    memcpy(root, rounds.back().tree.root().b, 32);
    have_deep = false;
    return OR_OK;
  }
  int fri_round_info(int r, u64* ncoef, u64* D) override {
    if (r < 0 || (size_t)r >= rounds.size()) return OR_ERR_ARG;
    *ncoef = rounds[r].poly.size(); *D = rounds[r].D; return OR_OK;
  }
  int fri_round_poly_read(int r, u64* out) override {
    if (r < 0 || (size_t)r >= rounds.size()) return OR_ERR_ARG;
    for (size_t i = 0; i < rounds[r].poly.size(); i++) for (int k = 0; k < E; k++) out[i * E + k] = rounds[r].poly[i].c[k];
    return OR_OK;
  }
  int fri_round_codeword_read(int r, u64* out) override {
    if (r < 0 || (size_t)r >= rounds.size()) return OR_ERR_ARG;
    memcpy(out, rounds[r].tree.leafs.data(), rounds[r].tree.leafs.size() * 8); return OR_OK;
  }

  // fri.rs:115-189.  Serialised FriProof ("MSFP" layout, see include/ministark.h):
  //   for each window (previous, round), for each beta:
  //     6*E limbs  x1 y1 x2 y2 x3 y3          (fri.rs:148-154)
  //     u64 qlen | qlen*E limbs              (fri.rs:159-167)
  //     MerklePath(y1) | MerklePath(y2)      (fri.rs:170-172; MerkleTree::open_index layout)
  int fri_query(const u64* betas, int nq) override {
    if (rounds.size() != fri_rounds) return OR_ERR_STATE;
    fri_proof.clear();
    auto put64 = [&](u64 v) { for (int k = 0; k < 8; k++) fri_proof.push_back((u8)(v >> (8 * k))); };
    auto putE = [&](const Ext<F, E>& v) { for (int k = 0; k < E; k++) put64(v.c[k]); };
    for (size_t ri = 0; ri + 1 < rounds.size(); ri++) {
      Round& prev = rounds[ri]; Round& cur = rounds[ri + 1];
      if (prev.D / 2 != cur.D) return OR_ERR_SHAPE;  // fri.rs:134-137
      u64 gp = root_of_unity<F>(prev.D), gc = root_of_unity<F>(cur.D);
      for (int j = 0; j < nq; j++) {
        u64 beta = betas[j];
        if (beta > prev.D) beta %= prev.D;  // fri.rs:144-146 (quirk Q6: `>` not `>=`)
        Ext<F, E> x1 = e_from_base<F, E>(f_pow<F>(gp, beta));           // fri.rs:148
        Ext<F, E> x2 = e_from_base<F, E>(f_pow<F>(gp, cur.D + beta));   // fri.rs:149
        Ext<F, E> x3 = e_from_base<F, E>(f_pow<F>(gc, beta));           // fri.rs:150
        Ext<F, E> y1 = p_eval<F, E>(prev.poly, x1), y2 = p_eval<F, E>(prev.poly, x2), y3 = p_eval<F, E>(cur.poly, x3);
        putE(x1); putE(y1); putE(x2); putE(y2); putE(x3); putE(y3);
        // fri.rs:159-161  g(x) = a x + b through (x1,y1),(x2,y2)
        Ext<F, E> a = e_mul<F>(e_sub<F, E>(y2, y1), e_inv<F>(e_sub<F, E>(x2, x1)));
        Ext<F, E> b = e_sub<F, E>(y1, e_mul<F>(a, x1));
        Poly<F, E> g{b, a}; p_trim<F, E>(g);
        Poly<F, E> numerator = p_sub<F, E>(prev.poly, g);  // fri.rs:164
        // fri.rs:165, 283-289  (x - x1)(x - x2)
        Poly<F, E> van{e_mul<F>(x1, x2), e_neg<F, E>(e_add<F, E>(x1, x2)), e_one<F, E>()};
        Poly<F, E> q = p_div_monic<F, E>(numerator, van);  // fri.rs:166
        put64(q.size());
        for (auto& cf : q) putE(cf);
        int rc = prev.tree.generate_proof(y1.c, fri_proof); if (rc) return rc;  // fri.rs:170
        rc = prev.tree.generate_proof(y2.c, fri_proof); if (rc) return rc;      // fri.rs:171
      }
    }
    return OR_OK;
  }
  size_t fri_proof_size() override { return fri_proof.size(); }
  int fri_proof_read(u8* out) override { memcpy(out, fri_proof.data(), fri_proof.size()); return OR_OK; }
};

// ---------------------------------------------------------------------------
// FRI verifier restatement (fri.rs:191-245) over the serialised proof; the
// transcript values (z, B, alpha, roots) are passed in.  Returns 1 on accept.
// ---------------------------------------------------------------------------
template <class F, int E>
static int fri_verify(int zae, size_t rounds, size_t nq, const u64* betas_in, const u64* zs, const u64* Bs, const u64* alphas,
                      const u8* roots /* rounds*32, round 0 first */, const u8* proof, size_t proof_len) {
  size_t domain_size = (size_t)1 << rounds;  // fri.rs:209,256
  u64 g0 = root_of_unity<F>(domain_size);
  std::vector<Ext<F, E>> prev_x3(nq);
  for (size_t j = 0; j < nq; j++) {
    u64 b = betas_in[j]; if (b > domain_size) b %= domain_size;  // fri.rs:277
    prev_x3[j] = e_from_base<F, E>(f_pow<F>(g0, b));              // fri.rs:210
  }
  const u8* p = proof; const u8* end = proof + proof_len;
  auto get64 = [&](u64* v) { if (p + 8 > end) return false; memcpy(v, p, 8); p += 8; return true; };
  auto getE = [&](Ext<F, E>* v) { for (int k = 0; k < E; k++) if (!get64(&v->c[k])) return false; return true; };
  for (size_t i = 0; i + 1 < rounds; i++) {
    Ext<F, E> z, B0, B1, al;
    for (int k = 0; k < E; k++) { z.c[k] = zs[i * E + k]; B0.c[k] = Bs[i * 2 * E + k]; B1.c[k] = Bs[i * 2 * E + E + k]; al.c[k] = alphas[i * E + k]; }
    for (size_t j = 0; j < nq; j++) {
      Ext<F, E> x1, y1, x2, y2, x3, y3;
      if (!getE(&x1) || !getE(&y1) || !getE(&x2) || !getE(&y2) || !getE(&x3) || !getE(&y3)) return 0;
      if (!e_eq<F, E>(x1, prev_x3[j])) return 0;               // fri.rs:217
      if (!e_eq<F, E>(e_neg<F, E>(x1), x2)) return 0;          // fri.rs:218
      if (!e_eq<F, E>(e_mul<F>(x1, x1), x3)) return 0;         // fri.rs:219
      u64 qlen; if (!get64(&qlen)) return 0;
      Poly<F, E> q(qlen);
      for (u64 t = 0; t < qlen; t++) if (!getE(&q[t])) return 0;
      p_trim<F, E>(q);
      size_t total_degree = p_degree<F, E>(q) + 3;             // fri.rs:223-224
      if (total_degree < 2) return 0;
      if (total_degree > ((size_t)1 << (rounds - i))) return 0;  // fri.rs:226
      // fri.rs:229-234 linearity + DEEP adjustment
      Ext<F, E> a = e_mul<F>(e_sub<F, E>(y2, y1), e_inv<F>(e_sub<F, E>(x2, x1)));
      Ext<F, E> b = e_sub<F, E>(y1, e_mul<F>(a, x1));
      Ext<F, E> deep_adj = e_add<F, E>(e_mul<F>(y3, e_sub<F, E>(x3, z)), e_add<F, E>(B0, e_mul<F>(B1, al)));
      Ext<F, E> gal = e_add<F, E>(b, e_mul<F>(a, al));
      if (!e_eq<F, E>(gal, deep_adj)) return 0;
      for (int which = 0; which < 2; which++) {  // fri.rs:236-239
        const u8* path = p;
        size_t hdr = 8 + 2 * E * 8;
        if (p + hdr + 8 > end) return 0;
        u64 nlev; memcpy(&nlev, p + hdr, 8);
        size_t plen = hdr + 8 + nlev * 2 * 32;
        if (p + plen > end) return 0;
        const Ext<F, E>& y = which ? y2 : y1;
        const u64* neigh = (const u64*)(path + 8);
        bool contains = false;
        for (int t = 0; t < 2; t++) { bool eq = true; for (int k = 0; k < E; k++) if (neigh[t * E + k] != y.c[k]) eq = false; if (eq) contains = true; }
        if (!contains) return 0;
        if (!merkle_check_proof(roots + i * 32, path, plen, E, 2, 2, zae)) return 0;
        p += plen;
      }
      prev_x3[j] = x3;  // fri.rs:240
    }
  }
  return p == end;
}

// ---------------------------------------------------------------------------
// C entry points (ctypes)
// ---------------------------------------------------------------------------
#define FIELD_COMMA ,
#define FIELD_DISPATCH(field, ext, EXPR_GL1, EXPR_GL2, EXPR_BB1, EXPR_BB2, EXPR_BB4) \
  do {                                                                            \
    if ((field) == 0 && (ext) == 1) { EXPR_GL1; }                                 \
    else if ((field) == 0 && (ext) == 2) { EXPR_GL2; }                            \
    else if ((field) == 1 && (ext) == 1) { EXPR_BB1; }                            \
    else if ((field) == 1 && (ext) == 2) { EXPR_BB2; }                            \
    else if ((field) == 1 && (ext) == 4) { EXPR_BB4; }                            \
  } while (0)

extern "C" {

void or_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
  if (g_threads > 1) {  // create the thread team now: its first use costs ~1 s in a process that already hosts another OpenMP runtime
    volatile int sink = 0;
#pragma omp parallel num_threads(g_threads)
    { sink = sink + 0; }
  }
}
int or_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
int or_is_power_of_two(u64 n) { return is_power_of_two(n); }
long or_logarithm_of_two_k(u64 n, u64 base) { return logarithm_of_two_k(n, base); }
u64 or_ceil_log2_k(u64 n, u64 base) { return ceil_log2_k(n, base); }
int or_num_queries(int field, u64 security_bits, u64 blowup, u64 steps, u64* linking, u64* fri) {
  return num_queries_from_config(field == 0 ? 64 : 31, security_bits, blowup, steps, linking, fri);
}
u64 or_modulus(int field) { return field == 0 ? GL::P : BB::P; }
u64 or_root_of_unity(int field, u64 n) { return field == 0 ? root_of_unity<GL>(n) : root_of_unity<BB>(n); }
u64 or_mul(int field, u64 a, u64 b) { return field == 0 ? GL::mul(a, b) : BB::mul(a, b); }
u64 or_inv(int field, u64 a) { return field == 0 ? f_inv<GL>(a) : f_inv<BB>(a); }
u64 or_pow(int field, u64 a, u64 e) { return field == 0 ? f_pow<GL>(a, e) : f_pow<BB>(a, e); }
void or_sha256(const u8* msg, size_t len, u8 out[32]) { sha256(msg, len, out); }

// ext arithmetic on limb arrays (tests cross-check against big-int python)
int or_ext_mul(int field, int ext, const u64* a, const u64* b, u64* out) {
  int ok = 0;
#define DO(FF, EE) { Ext<FF, EE> x, y; memcpy(x.c, a, EE * 8); memcpy(y.c, b, EE * 8); Ext<FF, EE> r = e_mul<FF>(x, y); memcpy(out, r.c, EE * 8); ok = 1; }
  FIELD_DISPATCH(field, ext, DO(GL, 1), DO(GL, 2), DO(BB, 1), DO(BB, 2), DO(BB, 4));
#undef DO
  return ok ? OR_OK : OR_ERR_ARG;
}
int or_ext_inv(int field, int ext, const u64* a, u64* out) {
  int ok = 0;
#define DO(FF, EE) { Ext<FF, EE> x; memcpy(x.c, a, EE * 8); Ext<FF, EE> r = e_inv<FF>(x); memcpy(out, r.c, EE * 8); ok = 1; }
  FIELD_DISPATCH(field, ext, DO(GL, 1), DO(GL, 2), DO(BB, 1), DO(BB, 2), DO(BB, 4));
#undef DO
  return ok ? OR_OK : OR_ERR_ARG;
}
// Display string of `count` elements concatenated (merkle.rs:163-166); returns length
size_t or_display(int ext, const u64* limbs, size_t count, int zae, char* out, size_t cap) {
  std::string s;
  for (size_t i = 0; i < count; i++) append_display(s, limbs + i * ext, ext, zae);
  if (out && cap) { size_t n = s.size() < cap ? s.size() : cap; memcpy(out, s.data(), n); }
  return s.size();
}

// standalone transforms: natural order in/out (air.rs:154 / starks.rs:89 / fri.rs:350)
int or_intt(int field, u64* a, size_t n) {
  if (!n || !is_power_of_two(n)) return OR_ERR_SHAPE;
  if (field == 0) { NttPlan<GL> pl(n, true); ntt_inplace<GL>(pl, a, 1); u64 ni = f_inv<GL>(n % GL::P); for (size_t i = 0; i < n; i++) a[i] = GL::mul(a[i], ni); }
  else { NttPlan<BB> pl(n, true); ntt_inplace<BB>(pl, a, 1); u64 ni = f_inv<BB>(n % BB::P); for (size_t i = 0; i < n; i++) a[i] = BB::mul(a[i], ni); }
  return OR_OK;
}
int or_ntt(int field, u64* a, size_t n) {
  if (!n || !is_power_of_two(n)) return OR_ERR_SHAPE;
  if (field == 0) { NttPlan<GL> pl(n, false); ntt_inplace<GL>(pl, a, 1); }
  else { NttPlan<BB> pl(n, false); ntt_inplace<BB>(pl, a, 1); }
  return OR_OK;
}
// coset LDE of one column: out[i] = P(shift * g_L^i), i < L
int or_coset_lde(int field, const u64* coeffs, size_t ncoef, u64 shift, u64* out, size_t L) {
  if (!L || !is_power_of_two(L) || ncoef > L) return OR_ERR_SHAPE;
  u64 s = 1;
  for (size_t k = 0; k < L; k++) {
    if (field == 0) { out[k] = k < ncoef ? GL::mul(coeffs[k], s) : 0; s = GL::mul(s, shift); }
    else { out[k] = k < ncoef ? BB::mul(coeffs[k], s) : 0; s = BB::mul(s, shift); }
  }
  return or_ntt(field, out, L);
}

// Merkle tree over `leaf_num` elements of `ext` limbs each (merkle.rs:81-148).
// nodes_out (may be NULL) receives all nodes level-major, root last.
int or_merkle_build(int ext, const u64* leafs, size_t leaf_num, size_t lpn, size_t ic, int zae, u8* nodes_out, size_t nodes_cap, size_t* nnodes, u8 root[32]) {
  MerkleTree t;
  int rc = t.build(leafs, leaf_num, ext, lpn, ic, zae);
  if (rc) return rc;
  if (nnodes) *nnodes = t.nodes.size();
  if (nodes_out) { if (nodes_cap < t.nodes.size()) return OR_ERR_ARG; memcpy(nodes_out, t.nodes.data(), t.nodes.size() * 32); }
  if (root) memcpy(root, t.root().b, 32);
  return OR_OK;
}
// merkle.rs:188-207 over a freshly built tree shape (no hashing needed)
int or_merkle_parent_idx(size_t leaf_num, size_t lpn, size_t ic, size_t index, size_t* out) {
  MerkleTree t; t.E = 1; t.lpn = lpn; t.ic = ic;
  t.leafs.assign(leaf_num, 0);
  size_t m = leaf_num / lpn, total = 0; for (;;) { total += m; if (m <= 1) break; m /= ic; }
  t.nodes.resize(total);
  return t.get_parent_idx(index, out);
}
// generate_proof by leaf VALUE (merkle.rs:272-288); returns serialised path length or <0
long or_merkle_prove(int ext, const u64* leafs, size_t leaf_num, size_t lpn, size_t ic, int zae, const u64* leaf, u8* out, size_t cap) {
  MerkleTree t;
  int rc = t.build(leafs, leaf_num, ext, lpn, ic, zae);
  if (rc) return rc;
  std::vector<u8> buf;
  rc = t.generate_proof(leaf, buf);
  if (rc) return rc;
  if (out) { if (cap < buf.size()) return OR_ERR_ARG; memcpy(out, buf.data(), buf.size()); }
  return (long)buf.size();
}
int or_merkle_check_proof(const u8 root[32], const u8* path, size_t len, int ext, size_t lpn, size_t ic, int zae) {
  return merkle_check_proof(root, path, len, ext, lpn, ic, zae);
}

// ---- prover session (Goldilocks -> Fp2, BabyBear -> Fp4: field.rs:38-41,67-70) ----
void* or_create(int field, int zero_as_empty) {
  if (field == 0) return new Session<GL, 2>(zero_as_empty);
  if (field == 1) return new Session<BB, 4>(zero_as_empty);
  return nullptr;
}
// FRI over an arbitrary tower (the reference's fri.rs:397-454 unit tests run
// FRI directly on GoldilocksFp / GoldilocksFp2)
void* or_create_ext(int field, int ext, int zero_as_empty) {
  void* r = nullptr;
  FIELD_DISPATCH(field, ext, r = new Session<GL FIELD_COMMA 1>(zero_as_empty), r = new Session<GL FIELD_COMMA 2>(zero_as_empty),
                 r = new Session<BB FIELD_COMMA 1>(zero_as_empty), r = new Session<BB FIELD_COMMA 2>(zero_as_empty), r = new Session<BB FIELD_COMMA 4>(zero_as_empty));
  return r;
}
void or_destroy(void* s) { delete (SessionBase*)s; }
int or_ext_degree(void* s) { return ((SessionBase*)s)->ext_degree(); }
int or_trace_commit(void* s, const u64* trace, size_t N, size_t w, size_t lpn, u8 root[32]) { return ((SessionBase*)s)->trace_commit(trace, N, w, lpn, root); }
int or_interpolate(void* s) { return ((SessionBase*)s)->interpolate(); }
int or_polys_lincomb(void* s, const u64* scalars, const int* idx, int k) { return ((SessionBase*)s)->polys_lincomb(scalars, idx, k); }
int or_polys_count(void* s) { return ((SessionBase*)s)->polys_count(); }
int or_poly_read(void* s, int i, u64* out) { return ((SessionBase*)s)->poly_read(i, out); }
int or_lde_commit(void* s, size_t blowup, u64 shift, size_t lpn, u8 root[32]) { return ((SessionBase*)s)->lde_commit(blowup, shift, lpn, root); }
int or_lde_read(void* s, u64* out) { return ((SessionBase*)s)->lde_read(out); }
int or_lde_nodes_read(void* s, u8* out) { return ((SessionBase*)s)->lde_nodes_read(out); }
int or_mix(void* s, u64 r) { return ((SessionBase*)s)->mix(r); }
int or_validity_read(void* s, u64* out) { return ((SessionBase*)s)->validity_read(out); }
int or_eval_ext(void* s, const u64* z, int q, u64* out) { return ((SessionBase*)s)->eval_ext(z, q, out); }
int or_verify_ood(void* s, u64 r, const u64* z, int q, const u64* evals) { return ((SessionBase*)s)->verify_ood(r, z, q, evals); }
int or_fri_begin(void* s, size_t blowup, size_t rounds, u8 root0[32]) { return ((SessionBase*)s)->fri_begin(blowup, rounds, root0); }
int or_fri_deep(void* s, const u64* z, u64* B) { return ((SessionBase*)s)->fri_deep(z, B); }
int or_fri_fold_commit(void* s, const u64* alpha, u8 root[32]) { return ((SessionBase*)s)->fri_fold_commit(alpha, root); }
int or_fri_round_info(void* s, int r, u64* ncoef, u64* D) { return ((SessionBase*)s)->fri_round_info(r, ncoef, D); }
int or_fri_round_poly_read(void* s, int r, u64* out) { return ((SessionBase*)s)->fri_round_poly_read(r, out); }
int or_fri_round_codeword_read(void* s, int r, u64* out) { return ((SessionBase*)s)->fri_round_codeword_read(r, out); }
int or_fri_query(void* s, const u64* betas, int nq) { return ((SessionBase*)s)->fri_query(betas, nq); }
size_t or_fri_proof_size(void* s) { return ((SessionBase*)s)->fri_proof_size(); }
int or_fri_proof_read(void* s, u8* out) { return ((SessionBase*)s)->fri_proof_read(out); }

int or_fri_verify(int field, int ext, int zae, size_t rounds, size_t nq, const u64* betas, const u64* zs, const u64* Bs, const u64* alphas,
                  const u8* roots, const u8* proof, size_t proof_len) {
  int r = -100;
#define DO(FF, EE) r = fri_verify<FF, EE>(zae, rounds, nq, betas, zs, Bs, alphas, roots, proof, proof_len)
  FIELD_DISPATCH(field, ext, DO(GL, 1), DO(GL, 2), DO(BB, 1), DO(BB, 2), DO(BB, 4));
#undef DO
  return r;
}

}  // extern "C"
