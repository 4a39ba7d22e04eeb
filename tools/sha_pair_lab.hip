// Lab (MI355X): one Merkle inner node (SHA-256 of 64 bytes + its constant padding block) hashed by a PAIR of lanes - the e-side (e, f, g, h) of a round in the even
// lane, the a-side (a, b, c, d) in the odd one, as the same instruction stream with per-lane rotation amounts, Maj(a,b,c) = Ch(a ^ c, b, c), the message schedule's
// sigma0 / sigma1 split over the two lanes, and one DPP add per exchange - against the one-lane-per-node formulation of csrc/merkle.hpp, as a DEPENDENT CHAIN of nodes
// in a lone wave (the situation of the upper tree levels: a level costs what one node costs).  Prints us per node for both, and checks they agree.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mini-stark_amd/csrc tools/sha_pair_lab.hip -o /tmp/sha_pair_lab && /tmp/sha_pair_lab
#include "merkle.hpp"
#include <cstdio>
using msmerkle::Sha256;
using msmerkle::SHA_K;

__device__ __forceinline__ u32 swap1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]: the pair's other lane
__device__ __forceinline__ u32 rotv(u32 x, u32 r) { return __builtin_amdgcn_alignbit(x, x, r); }
__device__ __forceinline__ u32 sel(u32 a, u32 b, u32 m) { return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4); }               // m ? a : b, bitwise
struct PairConsts { u32 M, R1, R2, R3, Sa, Sb, Sc; };
__device__ __forceinline__ PairConsts pair_consts(int lane) {
  const bool A = lane & 1;
  PairConsts c; c.M = A ? ~0u : 0u;
  c.R1 = A ? 2 : 6; c.R2 = A ? 13 : 11; c.R3 = A ? 22 : 25;
  c.Sa = A ? 17 : 7; c.Sb = A ? 19 : 18; c.Sc = A ? 10 : 3;
  return c;
}
// one round; kw = K + W of the round (used by the e-lane only)
__device__ __forceinline__ void pair_round(u32& x, u32& y, u32& z, u32& v, u32 kw, const PairConsts& c) {
  const u32 S = msmerkle::xor3(rotv(x, c.R1), rotv(x, c.R2), rotv(x, c.R3));
  const u32 xp = __builtin_amdgcn_bitop3_b32(x, z, c.M, 0x78);    // a-lane: a ^ c (Maj(a,b,c) = Ch(a ^ c, b, c)); e-lane: e
  const u32 F = msmerkle::ch3(xp, y, z);
  const u32 s = S + F;                                            // e-lane: Sigma1 + Ch; a-lane: T2
  const u32 t = s + v + kw;                                       // e-lane: T1 (a-lane: unused)
  const u32 snd = sel(v, t, c.M), own = sel(s, t, c.M);           // a-lane sends d and keeps T2; e-lane sends T1 and keeps T1
  const u32 n = swap1(snd) + own;                                 // a-lane: T1 + T2 = new a; e-lane: d + T1 = new e
  v = z; z = y; y = x; x = n;
}
// cv: this lane's half of the chaining value (a-lane H0..3, e-lane H4..7); w: the 16 message words (both lanes hold them all; clobbered)
__device__ __forceinline__ void pair_compress(u32 (&cv)[4], u32 (&w)[16], const PairConsts& c) {
  u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const u32 r = sel(w[(i + 14) & 15], w[(i + 1) & 15], c.M);   // a-lane: W[i-2] (sigma1); e-lane: W[i-15] (sigma0)
      const u32 o = sel(w[(i + 9) & 15], w[i & 15], c.M);          // a-lane: W[i-7]; e-lane: W[i-16]
      const u32 sig = msmerkle::xor3(rotv(r, c.Sa), rotv(r, c.Sb), r >> c.Sc);
      const u32 p = sig + o;
      w[i & 15] = p + swap1(p);
    }
    pair_round(x, y, z, v, w[i & 15] + SHA_K[i], c);
  }
  cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
}
template <u32 MSG_BITS> __device__ __forceinline__ void pair_compress_pad(u32 (&cv)[4], const PairConsts& c) {
  u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
  for (int i = 0; i < 64; i++) pair_round(x, y, z, v, msmerkle::PadBlock<MSG_BITS>::T.kw[i], c);
  cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
}

// chain: digest_{k+1} = H(digest_k || right), through LDS as the subtree kernel hands a level to the next
template <bool PAIR> __global__ __launch_bounds__(64) void chain(u32* out, const u32* seed, int iters) {
  __shared__ u32 lds[64 * 16];
  const int lane = threadIdx.x;
  const int node = PAIR ? lane >> 1 : lane;
  u32* slot = lds + node * 16;
  if (!PAIR || !(lane & 1)) { for (int i = 0; i < 8; i++) { slot[i] = seed[i] + node; slot[8 + i] = seed[8 + i] * 3u + node; } }
  __syncthreads();
  const PairConsts c = pair_consts(lane);
  for (int it = 0; it < iters; it++) {
    u32 w[16];
    const msmerkle::uint4_t* c4 = reinterpret_cast<const msmerkle::uint4_t*>(slot);
#pragma unroll
    for (int q = 0; q < 4; q++) { const msmerkle::uint4_t v = c4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    if (PAIR) {
      u32 cv[4];
      const bool A = lane & 1;
      cv[0] = A ? 0x6a09e667u : 0x510e527fu; cv[1] = A ? 0xbb67ae85u : 0x9b05688cu; cv[2] = A ? 0x3c6ef372u : 0x1f83d9abu; cv[3] = A ? 0xa54ff53au : 0x5be0cd19u;
      pair_compress(cv, w, c);
      pair_compress_pad<512u>(cv, c);
      __syncthreads();
      msmerkle::uint4_t o; o.x = cv[0]; o.y = cv[1]; o.z = cv[2]; o.w = cv[3];
      reinterpret_cast<msmerkle::uint4_t*>(slot)[A ? 0 : 1] = o;     // the digest = the next node's left child
    } else {
      Sha256 h; h.init();
      h.compress(w);
      h.template compress_pad_block<512u>();
      __syncthreads();
      msmerkle::uint4_t o0, o1; o0.x = h.st[0]; o0.y = h.st[1]; o0.z = h.st[2]; o0.w = h.st[3]; o1.x = h.st[4]; o1.y = h.st[5]; o1.z = h.st[6]; o1.w = h.st[7];
      reinterpret_cast<msmerkle::uint4_t*>(slot)[0] = o0; reinterpret_cast<msmerkle::uint4_t*>(slot)[1] = o1;
    }
    __syncthreads();
  }
  if (!PAIR || !(lane & 1)) for (int i = 0; i < 8; i++) out[node * 8 + i] = slot[i];
}
int main() {
  u32 *d0, *d1, *ds; hipMalloc(&d0, 64 * 8 * 4); hipMalloc(&d1, 64 * 8 * 4); hipMalloc(&ds, 64);
  u32 hs[16]; for (int i = 0; i < 16; i++) hs[i] = 0x9e3779b9u * (i + 1);
  hipMemcpy(ds, hs, 64, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  float ms[2];
  for (int pair = 0; pair < 2; pair++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pair) chain<true><<<1, 64>>>(d1, ds, iters); else chain<false><<<1, 64>>>(d0, ds, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[pair], e0, e1);
    }
    printf("%s: %.3f us per node of a dependent chain in a lone wave\n", pair ? "lane pair per node " : "one lane per node  ", ms[pair] * 1e3 / iters);
  }
  u32 h0[64 * 8], h1[64 * 8];
  hipMemcpy(h0, d0, sizeof h0, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, sizeof h1, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 32 * 8; i++) bad += h0[i] != h1[i];   // the pair kernel has 32 nodes: they equal nodes 0..31 of the other
  printf("digests %s (%d of 256 words differ); pair / single = %.3f\n", bad ? "DIFFER" : "agree", bad, ms[1] / ms[0]);
  return bad != 0;
}
