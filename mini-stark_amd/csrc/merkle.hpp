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
// big-endian ASCII words (no per-digit work), shifted into place by funnel shifts
// and stored into a per-thread buffer of message words (one or two blocks + slack)
// that lives in LDS word-interleaved across the workgroup (bank-conflict free);
// complete 64-byte blocks are compressed with SHA-256 state and schedule in
// registers; final blocks without message bytes are finished by a compacted
// follow-up kernel (PadOnlyBlockKernel).  Both passes are
// integer-VALU-issue bound, not HBM bound: algorithmic traffic is
// lpn*E*sizeof(T) + 32 bytes per leaf group and 96 bytes per inner node against
// 2-3 compressions (~1440 instructions each).
#pragma once
#include "field.hpp"

namespace msmerkle {

constexpr int THREADS = 256;
constexpr int OVF_WORDS = 2;    // deferred pad-only block entry: leaf group, message bits (the SHA-256 state waits in the digest slot)
constexpr int OVF_LISTS = 256;  // the deferred entries go to OVF_LISTS lists (workgroup bx appends to list bx % OVF_LISTS): one hot counter would serialise the atomics
constexpr int PAD_GRID_Y = 4;    // workgroups of PadOnlyBlockKernel per list (grid-stride over its entries)
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
// the other 3-source VOP3 ops take ~4.5 (tools/valu_rate.hip, profiles/r04_valu_issue_rate.txt).
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

constexpr u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// K[i] + W[i] of the FIPS 180-4 padding block that follows a message of MSG_BITS bits ending on a block
// boundary (0x80, zeros, bit length).  Evaluated by the compiler: the bitop3 builtins of the run-time path
// are opaque to constant propagation, so the fold is spelled out here instead of being left to the optimiser.
struct PadBlockKW { u32 kw[64]; };
constexpr u32 c_rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
constexpr PadBlockKW make_pad_block_kw(u32 msg_bits) {
  PadBlockKW r{};
  u32 w[64] = {};
  w[0] = 0x80000000u; w[15] = msg_bits;
  for (int i = 16; i < 64; i++) {
    const u32 s0 = c_rotr(w[i - 15], 7) ^ c_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const u32 s1 = c_rotr(w[i - 2], 17) ^ c_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  for (int i = 0; i < 64; i++) r.kw[i] = SHA_K[i] + w[i];
  return r;
}
template <u32 MSG_BITS> struct PadBlock { static constexpr PadBlockKW T = make_pad_block_kw(MSG_BITS); };

struct Sha256 {
  u32 st[8];
  MS_HD void init() {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
  }
  // one compression of the 16 big-endian words w[0..15] (clobbers w)
  MS_HD void compress(u32 (&w)[16]) {
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
      u32 t1 = h + S1 + ch + SHA_K[i] + w[i & 15];
      u32 S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
      u32 mj = maj3(a, b, c);
      u32 t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  }
  // the padding block after MSG_BITS message bits: no message schedule at run time, K+W is one literal per round
  template <u32 MSG_BITS> MS_HD void compress_pad_block() {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      u32 S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
      u32 ch = ch3(e, f, g);
      u32 t1 = h + S1 + ch + PadBlock<MSG_BITS>::T.kw[i];
      u32 S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
      u32 mj = maj3(a, b, c);
      u32 t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  }
};

#ifndef MS_SUBTREE_PAIR_LEVELS
#define MS_SUBTREE_PAIR_LEVELS 1   // 0 (A/B builds): every level of InnerSubtreeKernel one parent per lane
#endif
// One inner node of the binary tree (64-byte message + its constant padding block) hashed by a PAIR of lanes (r05): the e-side (e, f, g, h) of a round in one lane,
// the a-side (a, b, c, d) in its partner (msrt::pair_swap: lane ^ 7), as ONE instruction stream - per-lane rotation amounts, Maj(a, b, c) = Ch(a ^ c, b, c) so that one
// bitop3 pair serves both, T1 and d exchanged by two bank-masked DPP adds, sigma0 / sigma1 of the message schedule split over the two lanes.  1800 VALU instructions per
// node where Sha256 needs 2319: for the levels of a tree whose nodes are fewer than the lanes that could hash them a LEVEL is one node's dependent chain in a lone wave,
// 3.39 instead of 4.09 us (tools/sha_pair_lab.hip, profiles/r05_sha_pair_lab.log).  Every lane of a pair calls every function below (the exchanges are wave operations).
struct Sha256Pair {
  u32 M, R1, R2, R3, Sa, Sb, Sc;   // a-lane mask; Sigma rotations (a-lane: Sigma0, e-lane: Sigma1); schedule rotations / shift (a-lane: sigma1, e-lane: sigma0)
  bool a_lane;
  u32 cv[4];                       // this lane's half of the chaining value: a-lane H0..H3, e-lane H4..H7
  static MS_HD bool is_a_lane(int lane) { return (lane >> 2) & 1; }
  // node index of a lane among the 4 nodes its group of eight lanes hashes (partners get the same index)
  static MS_HD int node_in_group(int lane) { return is_a_lane(lane) ? 3 - (lane & 3) : (lane & 3); }
  MS_DEV void init(int lane) {
    a_lane = is_a_lane(lane);
    M = a_lane ? ~0u : 0u;
    R1 = a_lane ? 2 : 6; R2 = a_lane ? 13 : 11; R3 = a_lane ? 22 : 25;
    Sa = a_lane ? 17 : 7; Sb = a_lane ? 19 : 18; Sc = a_lane ? 10 : 3;
  }
  MS_DEV void reset() {
    cv[0] = a_lane ? 0x6a09e667u : 0x510e527fu; cv[1] = a_lane ? 0xbb67ae85u : 0x9b05688cu; cv[2] = a_lane ? 0x3c6ef372u : 0x1f83d9abu; cv[3] = a_lane ? 0xa54ff53au : 0x5be0cd19u;
  }
  // kw = K + W of the round (used by the e-lane only)
  MS_DEV void round(u32& x, u32& y, u32& z, u32& v, u32 kw) const {
    const u32 S = xor3(msrt::rotr_var(x, R1), msrt::rotr_var(x, R2), msrt::rotr_var(x, R3));
    const u32 xp = ms_bitop3<0x78>(x, z, M);          // a ^ (b & c): the a-lane's a ^ c, the e-lane's e
    const u32 F = ch3(xp, y, z);
    const u32 s = S + F;                              // e-lane: Sigma1 + Ch; a-lane: T2
    const u32 t = s + v + kw;                         // e-lane: T1
    const u32 n = msrt::pair_exchange_add(a_lane, v, t, s);   // e-lane: d + T1 = the new e; a-lane: T1 + T2 = the new a
    v = z; z = y; y = x; x = n;
  }
  // the 16 message words w (both lanes hold them all; clobbered)
  MS_DEV void compress(u32 (&w)[16]) {
    u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      if (i >= 16) {
        const u32 r = ms_bitop3<0xE4>(w[(i + 14) & 15], w[(i + 1) & 15], M);   // c ? a : b - a-lane: W[i-2] (sigma1); e-lane: W[i-15] (sigma0)
        const u32 o = ms_bitop3<0xE4>(w[(i + 9) & 15], w[i & 15], M);          // a-lane: W[i-7]; e-lane: W[i-16]
        const u32 sig = xor3(msrt::rotr_var(r, Sa), msrt::rotr_var(r, Sb), r >> Sc);
        const u32 p = sig + o;
        w[i & 15] = p + msrt::pair_swap(p);
      }
      round(x, y, z, v, w[i & 15] + SHA_K[i]);
    }
    cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
  }
  template <u32 MSG_BITS> MS_DEV void compress_pad_block() {
    u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
    for (int i = 0; i < 64; i++) round(x, y, z, v, PadBlock<MSG_BITS>::T.kw[i]);
    cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
  }
};

// {hi, lo} >> sh (low word), sh in {0, 8, 16, 24}
MS_HD u32 funnel_r(u32 hi, u32 lo, u32 sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return sh ? ((hi << (32 - sh)) | (lo >> sh)) : lo;
#endif
}
// ({hi, lo} << 8*l) >> 32, l in 0..3: the four bytes of the pair that start l bytes into `hi`.
// `sel` = drop_sel(l) is loop invariant (v_perm_b32 byte selector).
MS_HD u32 drop_sel(u32 l) { return 0x07060504u - l * 0x01010101u; }
MS_HD u32 drop_bytes(u32 hi, u32 lo, u32 l, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)l;
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  (void)sel;
  return l ? ((hi << (8 * l)) | (lo >> (32 - 8 * l))) : hi;
#endif
}

// byte stream -> SHA-256.  The not-yet-compressed tail of the message (< 64 bytes between appends) lives as
// big-endian words in a per-thread buffer of NWORDS words in LDS, word-interleaved across the workgroup
// (word i of thread t at buf[i*NT + t]: conflict free); buf[0] is always the first word of the next block.
// Appends are whole-register operations: a string of NW packed words is shifted into place by one funnel
// shift per word and stored at compile-time offsets from the write pointer; the word that holds the
// stream's tail is always stored with zero fill, so LDS is a valid image of the pending bytes at any time.
// `drain` is the only compression site (the 64-round compression is instantiated ONCE per kernel: inlining
// it at every append made the kernel I-cache bound); after a block is compressed the leftover words move
// down to the buffer start.  Between two drains at most 4*(NWORDS-17)-3 bytes may be appended.
template <int NWORDS, int NT, int MAXW, bool LAZY> struct ShaStream {
  static_assert(NWORDS >= 16 + MAXW + 1 && NWORDS <= 48, "buffer = one or two blocks + room for one element");
  Sha256 h;
  u32* buf;    // this thread's word 0
  u32 acc;     // the last four bytes appended (low byte = most recent)
  u32 total;   // bytes appended
  u32 done;    // words compressed (multiple of 16); before the final drain it is the stream index of buf[0]
  u32 fbase;   // `done` when the final drain began (buf[0] stays put from then on)
  MS_HD void init(u32* lds_words, int tid_) { h.init(); buf = lds_words + tid_; acc = 0; total = 0; done = 0; fbase = 0; }
  // append the first nbytes (1 <= nbytes <= 4*NW) bytes of W (big-endian, zero beyond nbytes);
  // last4 = the last four bytes of the stream after the append (only its low min(4, bytes so far) bytes matter)
  template <int NW> MS_HD void append_words(const u32 (&W)[NW], u32 nbytes, u32 last4) {
    const u32 sh = 8u * (total & 3u);
    u32* wp = buf + ((total >> 2) - done) * NT;
    u32 prev = acc;
#pragma unroll
    for (int k = 0; k < NW; k++) { wp[k * NT] = funnel_r(prev, W[k], sh); prev = W[k]; }
    wp[NW * NT] = funnel_r(prev, 0u, sh);
    total += nbytes;
    acc = last4;
  }
  // 1..4 bytes, left-aligned in w (zero below)
  MS_HD void append_small(u32 w, u32 nbytes) {
    const u32 W[1] = {w};
    const u32 l4 = nbytes >= 4 ? w : ((acc << (8 * nbytes)) | (w >> (32 - 8 * nbytes)));
    append_words<1>(W, nbytes, l4);
  }
  MS_HD void begin_final() { fbase = done; }
  // Compresses at most ONE block per call (the only compression site of the kernel).
  // Before the final drain: the block at buf[0..16) once it is complete, after which the leftover moves down.  LAZY buffers
  // hold two blocks and compress only when some lane of the wave runs out of room: the lanes of a wave cross block
  // boundaries within an element or two of each other, and compressing "whenever any lane has a block" ran the 64 rounds
  // ~1.4 times per block on wide rows (lpn = 128), mostly masked off.
  // Final drain (uniform over the workgroup; the caller has appended the 0x80 byte of a msg_bytes-byte message and called
  // begin_final): called until it stops returning MORE; the FIPS 180-4 padding is applied on the fly — words past the 0x80
  // byte read as zero, the last block carries the bit length.  When the length does not fit behind the message, the message
  // ends with one more block that holds NO message bytes (zeros + length).  Few lanes of a wave need it (0.9 % of the
  // Fibonacci LDE rows, but 44 % of its waves), so it is not compressed here: DEFER tells the caller to hand the state to
  // PadOnlyBlockKernel, which runs those lanes compacted.
  enum { DONE = 0, MORE = 1, DEFER = 2 };
  MS_HD int drain(bool final, u32 msg_bytes) {
    if (!final) {
      const u32 pending = (total >> 2) - done;
      const bool go = LAZY ? msrt::wave_any(pending + (u32)MAXW + 1u > (u32)NWORDS) : true;
      if (go && pending >= 16) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = buf[i * NT];
        h.compress(w);
        done += 16;
        u32 t[NWORDS - 16];
#pragma unroll
        for (int k = 0; k < NWORDS - 16; k++) t[k] = buf[(k + 16) * NT];
#pragma unroll
        for (int k = 0; k < NWORDS - 16; k++) buf[k * NT] = t[k];
      }
      return DONE;
    }
    const u32 valid_end = (total + 3) >> 2, limit = ((msg_bytes + 9 + 63) >> 6) << 4;
    if constexpr (!LAZY) {
      // fewer than 16 words were pending: one block at buf[0..16), possibly followed by the pad-only block
      const u32 pending = limit - done;
      u32 w[16];
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = buf[i * NT];
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = (done + i < valid_end) ? w[i] : 0u;
      if (pending == 16) { w[14] = 0; w[15] = msg_bytes * 8u; }  // messages are far below 2^29 bytes
      h.compress(w);
      done += 16;
      return pending == 32 ? DEFER : DONE;
    } else {
      const bool last = limit - done == 16;
      if (limit - 16 >= valid_end && last) return DEFER;   // only the pad-only block is left
      u32 w[16];
      const u32* blk = buf + (done - fbase) * NT;
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = (done + i < valid_end) ? blk[i * NT] : 0u;
      if (last) { w[14] = 0; w[15] = msg_bytes * 8u; }
      h.compress(w);
      done += 16;
      if (limit == done) return DONE;
      return (limit - 16 >= valid_end && limit - done == 16) ? DEFER : MORE;
    }
  }
};

// four decimal digits of c < 10^4 as big-endian BCD bytes (add 0x30303030 for ASCII)
MS_HD u32 bcd4(u32 c) {
  const u32 q = (c * 5243u) >> 19;                     // c / 100
  const u32 t = c + q * 65436u;                         // (q << 16) | (c % 100)
  const u32 tens = ((t * 103u) >> 10) & 0x000F000Fu;    // q / 10, (c % 100) / 10
  const u32 ones = t - tens * 10u;
  return (tens << 8) + ones;
}
MS_HD u32 pack4(u32 c) { return bcd4(c) + 0x30303030u; }
MS_HD u32 clz32(u32 x) { return (u32)__builtin_clz(x); }

// v -> NCH chunks of four decimal digits, most significant first
MS_HD void dec_chunks(const GL&, u64 v, u32 (&c)[5]) {
  // q1 = v / 10^16 from the high word: floor(vh * floor(2^84 / 10^16) / 2^52) is q1 or q1 - 1
  const u32 vh = (u32)(v >> 32);
  u32 q1 = (u32)(((u64)vh * 1934281311u) >> 32) >> 20;
  u64 r = v - (u64)q1 * 10000000000000000ULL;
  if (r >= 10000000000000000ULL) { r -= 10000000000000000ULL; q1++; }
  // q2 = r / 10^8 (r < 10^16 < 2^54) from r >> 22: floor(x * floor(2^54 / 10^8) / 2^32) is q2 or q2 - 1
  const u32 x = (u32)(r >> 22);
  u32 q2 = (u32)(((u64)x * 180143985u) >> 32);
  u32 r2 = (u32)r - q2 * 100000000u;  // < 2 * 10^8: the low word is enough
  if (r2 >= 100000000u) { r2 -= 100000000u; q2++; }
  c[0] = q1;
  c[1] = q2 / 10000u; c[2] = q2 - c[1] * 10000u;
  c[3] = r2 / 10000u; c[4] = r2 - c[3] * 10000u;
}
MS_HD void dec_chunks(const BB&, u64 v, u32 (&c)[3]) {
  const u32 x = (u32)v;
  c[0] = x / 100000000u; const u32 r = x - c[0] * 100000000u;
  c[1] = r / 10000u; c[2] = r - c[1] * 10000u;
}

// canonical decimal of a base element, most significant digit first, no leading zeros;
// ZERO -> "" (zero_as_empty) or "0".
template <class F, class S> MS_HD void put_dec(S& s, typename F::T v_, int zero_as_empty) {
  const u64 v = F::to_u64(v_);
  constexpr int NCH = (F::MAX_DIGITS + 3) / 4;  // 5 (Goldilocks, 20 digits) / 3 (BabyBear, 10 digits -> 12 chars)
  u32 c[NCH];
  dec_chunks(F(), v, c);
  if (c[0] != 0) {
    // common case (>= 10^16 resp. >= 10^8): at most 3 leading zero characters, all in the first word.
    // The 4*NCH characters are packed, shifted left by `lead` bytes and appended as one string.
    u32 W[NCH];
    const u32 b0 = bcd4(c[0]);
    const u32 lead = clz32(b0) >> 3;
    W[0] = b0 + 0x30303030u;
#pragma unroll
    for (int j = 1; j < NCH; j++) W[j] = pack4(c[j]);
    const u32 sel = drop_sel(lead);
    u32 X[NCH];
#pragma unroll
    for (int j = 0; j + 1 < NCH; j++) X[j] = drop_bytes(W[j], W[j + 1], lead, sel);
    X[NCH - 1] = drop_bytes(W[NCH - 1], 0u, lead, sel);
    s.template append_words<NCH>(X, 4u * NCH - lead, W[NCH - 1]);
    return;
  }
  if (v == 0) { if (!zero_as_empty) s.append_small(0x30000000u, 1); return; }
  // short values: chunk by chunk
  u32 lead = 0; bool found = false;  // leading zero characters
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const u32 z = c[j] == 0 ? 4u : (c[j] < 10u ? 3u : (c[j] < 100u ? 2u : (c[j] < 1000u ? 1u : 0u)));
    if (!found) lead += z;
    found = found || (c[j] != 0);
  }
  for (int j = 1; j < NCH; j++) {  // c[0] == 0 here
    const int k = 4 * (j + 1) - (int)lead;
    if (k >= 4) s.append_small(pack4(c[j]), 4);
    else if (k > 0) s.append_small(pack4(c[j]) << (8 * (4 - k)), (u32)k);  // leading chunk: 1..3 digits
  }
}
// arkworks' Display of an extension element, cut at its base limbs: the characters before and after the
// decimal of limb k (0 <= k < E).  E = 2: "QuadExtField(" c0 " + " c1 " * u)"; E = 4 nests the same form.
template <int E> struct Affix;
template <> struct Affix<1> {
  static constexpr int MAX_BYTES = 0;
  template <class S> static MS_HD void before(S&, u32) {}
  template <class S> static MS_HD void after(S&, u32) {}
};
template <class S> MS_HD void put_open(S& s) { const u32 W[4] = {0x51756164u, 0x45787446u, 0x69656c64u, 0x28000000u}; s.template append_words<4>(W, 13, 0x656c6428u); }  // "QuadExtField("
template <class S> MS_HD void put_plus(S& s) { s.append_small(0x202b2000u, 3); }                                                                                  // " + "
template <class S> MS_HD void put_close(S& s) { const u32 W[2] = {0x202a2075u, 0x29000000u}; s.template append_words<2>(W, 5, 0x2a207529u); }                     // " * u)"
template <> struct Affix<2> {
  static constexpr int MAX_BYTES = 13;  // per limb, before + after
  template <class S> static MS_HD void before(S& s, u32 k) { if (k == 0) put_open(s); else put_plus(s); }
  template <class S> static MS_HD void after(S& s, u32 k) { if (k == 1) put_close(s); }
};
template <> struct Affix<4> {
  static constexpr int MAX_BYTES = 26;
  template <class S> static MS_HD void before(S& s, u32 k) {
    if (k == 0) {  // "QuadExtField(QuadExtField("
      const u32 W[7] = {0x51756164u, 0x45787446u, 0x69656c64u, 0x28517561u, 0x64457874u, 0x4669656cu, 0x64280000u};
      s.template append_words<7>(W, 26, 0x656c6428u);
    } else if (k == 2) {  // " + QuadExtField("
      const u32 W[4] = {0x202b2051u, 0x75616445u, 0x78744669u, 0x656c6428u};
      s.template append_words<4>(W, 16, 0x656c6428u);
    } else put_plus(s);
  }
  template <class S> static MS_HD void after(S& s, u32 k) {
    if (k == 1) put_close(s);
    else if (k == 3) {  // " * u) * u)"
      const u32 W[3] = {0x202a2075u, 0x29202a20u, 0x75290000u};
      s.template append_words<3>(W, 10, 0x2a207529u);
    }
  }
};

// Leaf-group hashing.  Element f of the committed vector lives at
//   base + (f % width) * col_stride + (f / width) * row_stride + limb * limb_stride
// (row-major trace: width=1,row_stride=1; column-major LDE: width=c,col_stride=L;
//  FRI codeword: width=1, limb_stride=D).
// One thread owns one digest and walks the base limbs of its group: affix, decimal, affix, drain.
// A VIRTUAL column of the committed matrix (r03): column `col` = sum_t s[t] * column src[t] of the same row, computed while the row is hashed - the linear
// LDE columns (transition polynomials that ms_polys_lincomb defined, starks.rs:80-91 by linearity) are only ever hashed, so they need not exist in HBM.
constexpr int LIN_MAXT = 4;
struct LinColSpec { u32 n /* 0: a stored column */; u32 src[LIN_MAXT]; u64 s[LIN_MAXT]; };
template <class F, int E, bool LAZY = false> struct LeafHashKernel {
  typedef typename F::T T;
  static constexpr int THREADS = msmerkle::THREADS;
  struct Params {
    const T* base; size_t col_stride, row_stride, limb_stride;
    u32 width, lpn; int zero_as_empty;
    size_t ngroups;
    u32* nodes;  // 8 words per digest, standard byte order in memory
    u32* ovf_count; u32* ovf; u32 ovf_cap;  // deferred pad-only blocks: OVF_LISTS counters, and lists of ovf_cap entries of OVF_WORDS words (group, message bits); ovf == nullptr: no lists, such blocks are compressed in place
    // run_len != 0: the launch hashes `ngroups` groups that are RUNS of run_len consecutive groups, run_stride apart, from g_first on (one slice of
    // every peer's chunk of a sharded commitment: its digests can travel while the next slice is hashed)
    size_t g_first; u32 run_len, run_stride;
    const LinColSpec* lin;   // per column of the matrix (width entries), or nullptr: no virtual columns
    size_t out_g0;           // the digest of group g goes to nodes[g - out_g0] (a rank hashing the contiguous groups [out_g0, out_g0 + ngroups) of data every rank holds)
  };
  static constexpr int MAX_BYTES = F::MAX_DIGITS + Affix<E>::MAX_BYTES;  // appended between two drains
  static constexpr int MAXW = (3 + MAX_BYTES + 3) / 4 + 1;               // words one iteration can touch past the write position
  static constexpr int NWORDS = (LAZY ? 32 : 16) + MAXW + 1;
  typedef ShaStream<NWORDS, THREADS, MAXW, LAZY> Stream;
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_HD size_t lds_bytes() { return (size_t)NWORDS * THREADS * sizeof(u32); }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char* lds) {
    size_t g = (size_t)bx * nthreads + tid;
    if (g >= p.ngroups) return;
    if (p.run_len) g = p.g_first + (g / p.run_len) * (size_t)p.run_stride + g % p.run_len;
    Stream s; s.init(reinterpret_cast<u32*>(lds), tid);
    size_t f = g * p.lpn;
    size_t row = f / p.width; u32 col = (u32)(f - row * p.width);
    const u32 nlimbs = p.lpn * (u32)E;
    u32 msg_bytes = 0;
    bool deferred = false;
    for (u32 j = 0; LAZY || j <= nlimbs; j++) {
      if (j < nlimbs) {
        const u32 k = j & (u32)(E - 1);
        const T* rowp = p.base + row * p.row_stride + (size_t)k * p.limb_stride;
        T v;
        if (p.lin && p.lin[col].n) {   // (col is the same on every lane when a group is one row: scalar loads of the spec)
          const LinColSpec& lc = p.lin[col];
          v = 0;
          for (u32 t = 0; t < lc.n; t++) {   // scalars 1 and -1 (every Fibonacci transition has them) cost an add / sub, not a multiplication
            const T x = rowp[(size_t)lc.src[t] * p.col_stride];
            if (lc.s[t] == 1) v = F::add(v, x);
            else if (lc.s[t] == F::P - 1) v = F::sub(v, x);
            else v = F::add(v, F::mul(x, F::from_u64(lc.s[t])));
          }
        } else v = rowp[(size_t)col * p.col_stride];
        Affix<E>::before(s, k);
        put_dec<F>(s, v, p.zero_as_empty);
        Affix<E>::after(s, k);
        if (k == (u32)(E - 1) && ++col == p.width) { col = 0; row++; }
      } else if (j == nlimbs) {
        msg_bytes = s.total;
        s.append_small(0x80000000u, 1);
        s.begin_final();
      }
      const int st = s.drain(j >= nlimbs, msg_bytes);  // the only compression site
      if constexpr (LAZY) { if (j >= nlimbs && st != Stream::MORE) { deferred = st == Stream::DEFER; break; } }
      else deferred = st == Stream::DEFER;
    }
    if (deferred && !p.ovf) {   // no deferred-block lists (the fused FRI round, fri_tail.hpp: a launch of a few waves has nothing to compact): the pad-only block right here
      u32 w[16];
#pragma unroll
      for (int i = 0; i < 15; i++) w[i] = 0;
      w[15] = msg_bytes * 8u;
      s.h.compress(w);
      deferred = false;
    }
    const u32 list = (u32)bx % (u32)OVF_LISTS;
    const u32 slot = msrt::wave_alloc_slot(p.ovf_count + list, deferred);
    const size_t go = g - p.out_g0;   // digest slot
    if (deferred) {
      u32* e = p.ovf + ((size_t)list * p.ovf_cap + slot) * OVF_WORDS;
      e[0] = (u32)go; e[1] = msg_bytes * 8u;
      u32* st = p.nodes + go * 8;  // the state is parked in the digest slot
#pragma unroll
      for (int i = 0; i < 8; i++) st[i] = s.h.st[i];
      return;
    }
    uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + go * 8);
    uint4_t o0, o1;
    o0.x = bswap32(s.h.st[0]); o0.y = bswap32(s.h.st[1]); o0.z = bswap32(s.h.st[2]); o0.w = bswap32(s.h.st[3]);
    o1.x = bswap32(s.h.st[4]); o1.y = bswap32(s.h.st[5]); o1.z = bswap32(s.h.st[6]); o1.w = bswap32(s.h.st[7]);
    out[0] = o0; out[1] = o1;
  }
};
// The last block of the messages LeafHashKernel deferred: zeros + bit length, one compacted lane per entry.
struct PadOnlyBlockKernel {
  static constexpr int THREADS = msmerkle::THREADS;
  struct Params { const u32* ovf_count; const u32* ovf; u32 ovf_cap; u32* nodes; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int nthreads, unsigned char*) {
    const u32 n = p.ovf_count[bx];   // list bx
    const u32 stride = (u32)nthreads * PAD_GRID_Y;
    for (u32 t = (u32)by * nthreads + tid; t < n; t += stride) {
      const u32* e = p.ovf + ((size_t)bx * p.ovf_cap + t) * OVF_WORDS;
      Sha256 h;
      const u32* st = p.nodes + (size_t)e[0] * 8;
#pragma unroll
      for (int i = 0; i < 8; i++) h.st[i] = st[i];
      u32 w[16];
#pragma unroll
      for (int i = 0; i < 15; i++) w[i] = 0;
      w[15] = e[1];
      h.compress(w);
      uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + (size_t)e[0] * 8);
      uint4_t o0, o1;
      o0.x = bswap32(h.st[0]); o0.y = bswap32(h.st[1]); o0.z = bswap32(h.st[2]); o0.w = bswap32(h.st[3]);
      o1.x = bswap32(h.st[4]); o1.y = bswap32(h.st[5]); o1.z = bswap32(h.st[6]); o1.w = bswap32(h.st[7]);
      out[0] = o0; out[1] = o1;
    }
  }
};

// Inner levels.  `nlevels` consecutive levels are processed by the launch: with
// nlevels > 1 the grid must be a single workgroup (fused tree top).
// IC > 0: inner_children fixed at compile time (IC = 2 is the prover's tree, starks.rs:290-301: the padding block's
// message schedule then folds to literals); IC = 0: taken from Params.
struct InnerHashParams { u32* nodes; size_t child_off, nchildren; u32 ic; u32 nlevels; u32* host_root; /* optional: page-locked host memory that also receives the root (nparents == 1) */
                         unsigned long long* aux_src; unsigned long long* aux_dst; /* optional: one more 8-byte result (the round polynomial's trimmed length) forwarded to page-locked host memory by the same thread */
                         msrt::HostFlag flag; /* optional: raised behind root and aux word, for a host that polls instead of synchronising (rt.hpp) */ };
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
      // message schedule of this block is a compile-time table (PadBlock).
      if constexpr (IC > 0) h.template compress_pad_block<(u32)IC * 256u>();
      else {
        w[0] = 0x80000000u;
#pragma unroll
        for (int i = 1; i < 15; i++) w[i] = 0;
        w[15] = ic * 256u;  // message bits (ic * 32 bytes)
        h.compress(w);
      }
      uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + (child_off + nchildren + g) * 8);
      uint4_t o0, o1;
      o0.x = bswap32(h.st[0]); o0.y = bswap32(h.st[1]); o0.z = bswap32(h.st[2]); o0.w = bswap32(h.st[3]);
      o1.x = bswap32(h.st[4]); o1.y = bswap32(h.st[5]); o1.z = bswap32(h.st[6]); o1.w = bswap32(h.st[7]);
      out[0] = o0; out[1] = o1;
      if (p.host_root && nparents == 1) {   // saves the caller the copy launches (32-byte root, 8-byte degree word) in front of its stream synchronisation
        uint4_t* hr = reinterpret_cast<uint4_t*>(p.host_root); hr[0] = o0; hr[1] = o1;
        if (p.aux_src) { *p.aux_dst = *p.aux_src; *p.aux_src = 0; }   // ... and the device word is zero again for the next round's scan
        msrt::raise_host_flag(p.flag);
      }
      if (stride == 0) break;
    }
  }
};

typedef InnerHashKernelT<0> InnerHashKernel;
typedef InnerHashKernelT<2> InnerHashKernel2;

// Binary-tree levels whose launches are latency, not throughput (a few thousand parents: far less than one wave per SIMD): workgroup b takes the 2^nlevels
// children [b << nlevels, (b + 1) << nlevels) of the level at child_off and hashes the nlevels levels above them, 2^(nlevels-1), ..., 1 threads at work.  Every
// node is written to its place in the level-major node array (the openings read them), but a level reads its children from LDS, as the state words the level
// below left there (no byte swaps either), behind a barrier that orders LDS traffic only: a level costs 4.5 us (the two compressions of a lone wave: 4.1 us)
// instead of the 6.0-6.3 us that one launch per level, or one workgroup walking the levels through global memory, took (r04 trace of one proof:
// profiles/r04_small_round_kernels_ab.log).  r05: from the level on where the workgroup has at most half as many parents as lanes, a parent is hashed by a PAIR of
// lanes (Sha256Pair above): 3.4 us of compressions per level instead of 4.1.
// nlevels <= MAX_LEVELS; grid = nchildren >> nlevels; host_root / aux as InnerHashKernelT.
struct InnerSubtreeKernel {
  static constexpr int THREADS = msmerkle::THREADS;
  static constexpr int MAX_LEVELS = 9;
  static_assert((1 << (MAX_LEVELS - 1)) == THREADS, "level 0 of a workgroup: one parent per thread");
  typedef InnerHashParams Params;
  static MS_HD size_t lds_bytes() { return (size_t)(THREADS + THREADS / 2) * 32; }
  static MS_DEV void run(const Params& p, int bx, int, int, int tid, unsigned char* lds) {
    u32* const buf0 = reinterpret_cast<u32*>(lds);          // levels 0, 2, 4, ...: <= THREADS digests
    u32* const buf1 = buf0 + (size_t)THREADS * 8;           // levels 1, 3, ...: <= THREADS / 2 digests
    const u32 nl = p.nlevels;
    size_t child_off = p.child_off, nchildren = p.nchildren;
    u32 pp = 1u << nl;
    Sha256Pair hp; hp.init(tid);
    for (u32 l = 0; l < nl; l++) {
      pp >>= 1;   // parents of this workgroup at this level
      const size_t nparents = nchildren >> 1;
      if (MS_SUBTREE_PAIR_LEVELS && 2 * pp <= (u32)THREADS) {
        // at most half as many parents as lanes: one parent per PAIR of lanes (Sha256Pair), whole groups of eight lanes at work (4 parents each; with fewer than 4
        // parents the spare pairs hash parent 0's children again and store nothing)
        const u32 nact = 2 * pp < 8 ? 8 : 2 * pp;
        if ((u32)tid < nact) {
          const u32 nd0 = ((u32)tid >> 3) * 4 + (u32)Sha256Pair::node_in_group(tid);
          const bool live = nd0 < pp;
          const u32 nd = live ? nd0 : 0;
          const size_t g = (size_t)bx * pp + nd;
          u32 w[16];
          if (l == 0) {
            const uint4_t* c4 = reinterpret_cast<const uint4_t*>(p.nodes + (child_off + 2 * g) * 8);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const uint4_t v = c4[q];
              w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
            }
          } else {
            const uint4_t* c4 = reinterpret_cast<const uint4_t*>(((l & 1) ? buf0 : buf1) + (size_t)nd * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint4_t v = c4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
          }
          hp.reset();
          hp.compress(w);
          hp.template compress_pad_block<512u>();
          const int half = hp.a_lane ? 0 : 1;   // the a-lane holds words 0..3 of the digest, the e-lane words 4..7
          uint4_t sv; sv.x = hp.cv[0]; sv.y = hp.cv[1]; sv.z = hp.cv[2]; sv.w = hp.cv[3];
          uint4_t ov; ov.x = bswap32(hp.cv[0]); ov.y = bswap32(hp.cv[1]); ov.z = bswap32(hp.cv[2]); ov.w = bswap32(hp.cv[3]);
          const bool fwd = p.host_root && nparents == 1;   // (uniform) the root also goes to page-locked host memory, stored by ONE lane: the e-lane fetches the other half
          uint4_t oth = ov;
          if (fwd) { oth.x = msrt::pair_swap(ov.x); oth.y = msrt::pair_swap(ov.y); oth.z = msrt::pair_swap(ov.z); oth.w = msrt::pair_swap(ov.w); }
          if (live) {
            if (l + 1 < nl) reinterpret_cast<uint4_t*>(((l & 1) ? buf1 : buf0) + (size_t)nd * 8)[half] = sv;
            reinterpret_cast<uint4_t*>(p.nodes + (child_off + nchildren + g) * 8)[half] = ov;
            if (fwd && !hp.a_lane) {
              uint4_t* hr = reinterpret_cast<uint4_t*>(p.host_root); hr[0] = oth; hr[1] = ov;
              if (p.aux_src) { *p.aux_dst = *p.aux_src; *p.aux_src = 0; }
              msrt::raise_host_flag(p.flag);
            }
          }
        }
      } else if ((u32)tid < pp) {
        const size_t g = (size_t)bx * pp + (u32)tid;
        Sha256 h; h.init();
        u32 w[16];
        if (l == 0) {
          const uint4_t* c4 = reinterpret_cast<const uint4_t*>(p.nodes + (child_off + 2 * g) * 8);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint4_t v = c4[q];
            w[4 * q] = bswap32(v.x); w[4 * q + 1] = bswap32(v.y); w[4 * q + 2] = bswap32(v.z); w[4 * q + 3] = bswap32(v.w);
          }
        } else {
          const uint4_t* c4 = reinterpret_cast<const uint4_t*>(((l & 1) ? buf0 : buf1) + (size_t)tid * 16);
#pragma unroll
          for (int q = 0; q < 4; q++) { const uint4_t v = c4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
        }
        h.compress(w);
        h.template compress_pad_block<512u>();
        if (l + 1 < nl) {
          uint4_t* o = reinterpret_cast<uint4_t*>(((l & 1) ? buf1 : buf0) + (size_t)tid * 8);
          uint4_t s0, s1;
          s0.x = h.st[0]; s0.y = h.st[1]; s0.z = h.st[2]; s0.w = h.st[3]; s1.x = h.st[4]; s1.y = h.st[5]; s1.z = h.st[6]; s1.w = h.st[7];
          o[0] = s0; o[1] = s1;
        }
        uint4_t* out = reinterpret_cast<uint4_t*>(p.nodes + (child_off + nchildren + g) * 8);
        uint4_t o0, o1;
        o0.x = bswap32(h.st[0]); o0.y = bswap32(h.st[1]); o0.z = bswap32(h.st[2]); o0.w = bswap32(h.st[3]);
        o1.x = bswap32(h.st[4]); o1.y = bswap32(h.st[5]); o1.z = bswap32(h.st[6]); o1.w = bswap32(h.st[7]);
        out[0] = o0; out[1] = o1;
        if (p.host_root && nparents == 1) {
          uint4_t* hr = reinterpret_cast<uint4_t*>(p.host_root); hr[0] = o0; hr[1] = o1;
          if (p.aux_src) { *p.aux_dst = *p.aux_src; *p.aux_src = 0; }
          msrt::raise_host_flag(p.flag);
        }
      }
      if (l + 1 < nl) msrt::wg_barrier();
      child_off += nchildren; nchildren = nparents;
    }
  }
};

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
  // One WAVE per opened leaf (grid = njobs workgroups), a lane per 4-byte word of the path: the sibling addresses depend on the leaf index only, so all loads of a
  // path are in flight together (r04; one THREAD per leaf walked the levels one after the other, each level's loads waiting behind the previous level's
  // stores - 105 us for the 128 openings of a 2^20-row proof, all of it latency).
  static MS_DEV void phase(int, const Params& pp, int bx, int, int tid, int nthreads, unsigned char*) {
    if ((u32)bx >= pp.njobs) return;
    const PathJob<F, E>& p = pp.jobs[bx];
    const size_t li = (size_t)*p.idx;
    if (li >= p.leaf_num) return;  // value not found: the host reports MS_ERR_LEAF_NOT_FOUND
    u64* o = reinterpret_cast<u64*>(p.out);
    const u32 nval = p.lpn * (u32)E;
    if (tid == 0) { o[0] = li; o[1 + nval] = p.nlevels; }
    const size_t start = li - li % p.lpn;  // merkle.rs:230-236
    for (u32 v = (u32)tid; v < nval; v += (u32)nthreads) o[1 + v] = F::to_u64(p.leafs[(size_t)(v % E) * p.limb_stride + start + v / E]);
    u32* o32 = reinterpret_cast<u32*>(o + 2 + nval);
    const u32 per = p.ic * 8, total = p.nlevels * per;
    const size_t grp = li / p.lpn, ngrp = p.leaf_num / p.lpn;
    for (u32 w = (u32)tid; w < total; w += (u32)nthreads) {  // merkle.rs:241-265: level l holds the ic siblings of the path's node at that level
      const u32 l = w / per, i = w % per;
      size_t cur, level_off;
      if (p.ic == 2) { cur = grp >> l; level_off = 2 * ngrp - ((2 * ngrp) >> l); }   // sum_{q<l} ngrp / 2^q
      else {
        cur = grp; level_off = 0;
        size_t level_n = ngrp;
        for (u32 q = 0; q < l; q++) { level_off += level_n; level_n /= p.ic; cur /= p.ic; }
      }
      const size_t s = cur - cur % p.ic;
      o32[w] = p.nodes[(level_off + s) * 8 + i];
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Sharded trees (ms_set_shard).  After the all-to-all, recv chunk k holds the digests of the leaves
// p = k + W*q (q < per) of this rank's contiguous range: interleave them into subtree leaf order.
struct InterleaveDigestsKernel {
  static constexpr int THREADS = 256;
  struct Params { const uint4_t* recv; uint4_t* leaves; size_t per /* digests per chunk */; u32 W; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    const size_t t = (size_t)bx * nthreads + tid;   // half-digest index in subtree leaf order
    const size_t pidx = t >> 1, half = t & 1;
    if (pidx >= p.per * p.W) return;
    const size_t k = pidx % p.W, q = pidx / p.W;
    p.leaves[t] = p.recv[(k * p.per + q) * 2 + half];
  }
};
// byte copies described by a device job table (gathering Merkle paths from the exchange buffer into the proof blob)
struct CopyJob { const unsigned char* src; unsigned char* dst; size_t bytes; };
struct CopyJobsKernel {
  static constexpr int THREADS = 64;
  struct Params { const CopyJob* jobs; u32 njobs; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int, int tid, int nthreads, unsigned char*) {
    if ((u32)bx >= p.njobs) return;
    const CopyJob j = p.jobs[bx];
    for (size_t i = (size_t)tid; i < j.bytes; i += (size_t)nthreads) j.dst[i] = j.src[i];
  }
};
// the same for LARGE ranges (the ranks' slices of a sharded proof's quotient polynomials, tens of MiB each): blockIdx.y = job, blockIdx.x = a 32 KiB piece of it;
// sources, destinations and lengths are multiples of 8 bytes (u64 elements of the MSFP blob)
struct CopyRangesKernel {
  static constexpr int THREADS = 256;
  static constexpr size_t WORDS = 4096;   // 8-byte words per workgroup
  struct Params { const CopyJob* jobs; u32 njobs; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& p, int bx, int by, int tid, int nthreads, unsigned char*) {
    if ((u32)by >= p.njobs) return;
    const CopyJob j = p.jobs[by];
    const size_t nw = j.bytes / 8, w0 = (size_t)bx * WORDS;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(j.src);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(j.dst);
    for (size_t i = w0 + (size_t)tid; i < w0 + WORDS && i < nw; i += (size_t)nthreads) dst[i] = src[i];
  }
};
// Top of a sharded tree: the W all-gathered (root | aux) records -> W contiguous roots, the maximum of the ranks' aux words (the trimmed length of a distributed round
// polynomial rides on the root all-gather: no collective of its own), and the levels above the W roots in the SAME launch (r05: two launches until then; every launch
// of this chain is latency on every rank): one workgroup unpacks the records, then walks the log2(W) levels with the children in LDS (InnerSubtreeKernel's device code),
// the root and the (maximum) aux word forwarded as `tree` says.  W <= 2^MAX_LEVELS.
struct ShardTopTreeKernel {
  static constexpr int THREADS = InnerSubtreeKernel::THREADS;
  static constexpr int REC = 64;   // bytes per rank: 32-byte root, 8-byte aux, padding
  struct Params { const unsigned char* recs; u32 W; unsigned long long* aux_max; InnerHashParams tree; /* nodes = the top's node array, child_off 0, nchildren W, nlevels log2 W */ };
  static MS_HD size_t lds_bytes() { return InnerSubtreeKernel::lds_bytes(); }
  static MS_DEV void run(const Params& p, int, int, int, int tid, unsigned char* lds) {
    for (u32 i = (u32)tid; i < p.W * 8; i += (u32)THREADS) p.tree.nodes[i] = reinterpret_cast<const u32*>(p.recs + (size_t)(i / 8) * REC)[i % 8];
    if (tid == 0 && p.aux_max) {
      unsigned long long m = 0;
      for (u32 r = 0; r < p.W; r++) { const unsigned long long v = *reinterpret_cast<const unsigned long long*>(p.recs + (size_t)r * REC + 32); if (v > m) m = v; }
      *p.aux_max = m;
    }
    msrt::wg_barrier_global();   // level 0 reads the roots other threads just stored
    if (p.tree.nlevels) { InnerSubtreeKernel::run(p.tree, 0, 0, 1, tid, lds); return; }
    // a one-rank world: the lone subtree root is the root
    if (p.tree.host_root && tid == 0) {
      for (int k = 0; k < 8; k++) p.tree.host_root[k] = p.tree.nodes[k];
      if (p.tree.aux_src) { *p.tree.aux_dst = *p.tree.aux_src; *p.tree.aux_src = 0; }
      msrt::raise_host_flag(p.tree.flag);
    }
  }
};
// MerklePath of a sharded binary tree, same layout as PathKernel; every byte is written by exactly ONE rank
// (the others leave zeros; the ranks' buffers are then summed):
//   leaf index, leaf values, level count : the rank that evaluated the leaf group  (group % W)
//   sibling pairs inside a subtree       : the rank that owns the contiguous range (group / Mloc)
//   sibling pairs of the replicated top  : rank 0
template <class F, int E> struct ShardPathJob {
  const typename F::T* cw; size_t limb_stride, m;   // local codeword: limb l, coset t, index q at cw[l*limb_stride + t*m + q]
  const u32* sub; const u32* top; size_t Mloc;      // subtree nodes (level-major, Mloc leaves) / top nodes (level-major, W subtree roots)
  u32 lpn, W, rank, nlevels /* log2(M) */;
  const unsigned long long* idx;                    // global element index of the opened leaf
  unsigned char* out;
};
template <class F, int E> struct ShardPathKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 64;
  struct Params { const ShardPathJob<F, E>* jobs; u32 njobs; };
  static MS_HD int nphases(const Params&) { return 1; }
  static MS_DEV void phase(int, const Params& pp, int bx, int, int tid, int nthreads, unsigned char*) {
    const u32 t = (u32)bx * nthreads + tid;
    if (t >= pp.njobs) return;
    const ShardPathJob<F, E>& p = pp.jobs[t];
    const size_t li = (size_t)*p.idx;
    if (li == (size_t)~0ULL) return;
    u64* o = reinterpret_cast<u64*>(p.out);
    const size_t grp = li / p.lpn;
    if (grp % p.W == p.rank) {
      const size_t q = grp / p.W;
      o[0] = li;
      for (u32 i = 0; i < p.lpn; i++)
        for (int k = 0; k < E; k++) o[1 + i * E + k] = F::to_u64(p.cw[(size_t)k * p.limb_stride + (size_t)i * p.m + q]);
      o[1 + p.lpn * E] = p.nlevels;
    }
    u32* o32 = reinterpret_cast<u32*>(o + 2 + p.lpn * E);
    u32 sub_levels = 0; while (((size_t)1 << sub_levels) < p.Mloc) sub_levels++;
    const size_t owner = grp / p.Mloc;
    size_t level_off = 0, level_n = p.Mloc;
    for (u32 l = 0; l < sub_levels; l++) {
      if (owner == p.rank) {
        const size_t cur = grp >> l, s = cur - (cur & 1) - owner * level_n;
        const u32* src = p.sub + (level_off + s) * 8;
        for (u32 i = 0; i < 16; i++) o32[(size_t)l * 16 + i] = src[i];
      }
      level_off += level_n; level_n >>= 1;
    }
    level_off = 0; level_n = p.W;
    for (u32 l = sub_levels; l < p.nlevels; l++) {
      if (p.rank == 0) {
        const size_t cur = grp >> l, s = cur - (cur & 1);
        const u32* src = p.top + (level_off + s) * 8;
        for (u32 i = 0; i < 16; i++) o32[(size_t)l * 16 + i] = src[i];
      }
      level_off += level_n; level_n >>= 1;
    }
  }
};

}  // namespace msmerkle
