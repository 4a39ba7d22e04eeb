// Lab (MI355X): one Merkle inner node (SHA-256 of 64 bytes + its constant padding block) hashed by a PAIR of lanes - the e-side (e, f, g, h) of a round in the even
// lane, the a-side (a, b, c, d) in the odd one, as the same instruction stream with per-lane rotation amounts, Maj(a,b,c) = Ch(a ^ c, b, c), the message schedule's
// sigma0 / sigma1 split over the two lanes, and one DPP add per exchange (a local neighbour-lane form, and merkle.hpp's Sha256Pair) - against the one-lane-per-node
// formulation of csrc/merkle.hpp, as a DEPENDENT CHAIN of nodes
// in a lone wave (the situation of the upper tree levels: a level costs what one node costs).  Prints us per node for both, and checks they agree.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mini-stark_amd/csrc tools/sha_pair_lab.hip -o /tmp/sha_pair_lab && /tmp/sha_pair_lab
#include "merkle.hpp"
#include <cstdio>
using msmerkle::Sha256;
using msmerkle::SHA_K;

__device__ __forceinline__ u32 swap1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]: the pair's other lane
__device__ __forceinline__ u32 rotv(u32 x, u32 r) { return __builtin_amdgcn_alignbit(x, x, r); }
__device__ __forceinline__ u32 sel(u32 a, u32 b, u32 m) { return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4); }               // m ? a : b, bitwise
__device__ __forceinline__ u32 swapm(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); }  // row_half_mirror: lane i <-> 7 - i of its 8 lanes
struct PairConsts { u32 M, R1, R2, R3, Sa, Sb, Sc; };
template <int MODE> __device__ __forceinline__ bool is_a_lane(int lane) { return MODE == 2 ? ((lane >> 2) & 1) : (lane & 1); }   // MODE 2: banks 1 and 3 of a row (bank_mask 0xa)
template <int MODE> __device__ __forceinline__ PairConsts pair_consts(int lane) {
  const bool A = is_a_lane<MODE>(lane);
  PairConsts c; c.M = A ? ~0u : 0u;
  c.R1 = A ? 2 : 6; c.R2 = A ? 13 : 11; c.R3 = A ? 22 : 25;
  c.Sa = A ? 17 : 7; c.Sb = A ? 19 : 18; c.Sc = A ? 10 : 3;
  return c;
}
// one round; kw = K + W of the round (used by the e-lane only)
template <int MODE> __device__ __forceinline__ void pair_round(u32& x, u32& y, u32& z, u32& v, u32 kw, const PairConsts& c) {
  const u32 S = msmerkle::xor3(rotv(x, c.R1), rotv(x, c.R2), rotv(x, c.R3));
  const u32 xp = __builtin_amdgcn_bitop3_b32(x, z, c.M, 0x78);    // a-lane: a ^ c (Maj(a,b,c) = Ch(a ^ c, b, c)); e-lane: e
  const u32 F = msmerkle::ch3(xp, y, z);
  const u32 s = S + F;                                            // e-lane: Sigma1 + Ch; a-lane: T2
  const u32 t = s + v + kw;                                       // e-lane: T1 (a-lane: unused)
  u32 n;
  if (MODE == 2) {   // e-lanes (banks 0, 2): n = partner's d + T1; a-lanes (banks 1, 3): n = partner's T1 + T2.  (2 wait states between the write of t and its DPP read)
    asm("v_add_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\ts_nop 0\n\tv_add_u32_dpp %0, %2, %3 row_half_mirror row_mask:0xf bank_mask:0xa" : "=&v"(n) : "v"(v), "v"(t), "v"(s));
  } else {
    const u32 snd = sel(v, t, c.M), own = sel(s, t, c.M);         // a-lane sends d and keeps T2; e-lane sends T1 and keeps T1
    n = swap1(snd) + own;                                         // a-lane: T1 + T2 = new a; e-lane: d + T1 = new e
  }
  v = z; z = y; y = x; x = n;
}
// cv: this lane's half of the chaining value (a-lane H0..3, e-lane H4..7); w: the 16 message words (both lanes hold them all; clobbered)
template <int MODE> __device__ __forceinline__ void pair_compress(u32 (&cv)[4], u32 (&w)[16], const PairConsts& c) {
  u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const u32 r = sel(w[(i + 14) & 15], w[(i + 1) & 15], c.M);   // a-lane: W[i-2] (sigma1); e-lane: W[i-15] (sigma0)
      const u32 o = sel(w[(i + 9) & 15], w[i & 15], c.M);          // a-lane: W[i-7]; e-lane: W[i-16]
      const u32 sig = msmerkle::xor3(rotv(r, c.Sa), rotv(r, c.Sb), r >> c.Sc);
      const u32 p = sig + o;
      w[i & 15] = p + (MODE == 2 ? swapm(p) : swap1(p));
    }
    pair_round<MODE>(x, y, z, v, w[i & 15] + SHA_K[i], c);
  }
  cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
}
template <int MODE, u32 MSG_BITS> __device__ __forceinline__ void pair_compress_pad(u32 (&cv)[4], const PairConsts& c) {
  u32 x = cv[0], y = cv[1], z = cv[2], v = cv[3];
#pragma unroll
  for (int i = 0; i < 64; i++) pair_round<MODE>(x, y, z, v, msmerkle::PadBlock<MSG_BITS>::T.kw[i], c);
  cv[0] += x; cv[1] += y; cv[2] += z; cv[3] += v;
}

// chain: digest_{k+1} = H(digest_k || right), through LDS as the subtree kernel hands a level to the next
template <int MODE> __global__ __launch_bounds__(64) void chain(u32* out, const u32* seed, int iters) {
  constexpr bool PAIR = MODE != 0;
  __shared__ u32 lds[64 * 16];
  const int lane = threadIdx.x;
  const bool A = PAIR && is_a_lane<MODE>(lane);
  const int node = MODE == 0 ? lane : (MODE == 1 ? lane >> 1 : (lane >> 3) * 4 + (A ? 3 - (lane & 3) : (lane & 3)));
  u32* slot = lds + node * 16;
  if (!A) { for (int i = 0; i < 8; i++) { slot[i] = seed[i] + node; slot[8 + i] = seed[8 + i] * 3u + node; } }
  __syncthreads();
  const PairConsts c = pair_consts<MODE>(lane);
  for (int it = 0; it < iters; it++) {
    u32 w[16];
    const msmerkle::uint4_t* c4 = reinterpret_cast<const msmerkle::uint4_t*>(slot);
#pragma unroll
    for (int q = 0; q < 4; q++) { const msmerkle::uint4_t v = c4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    if (MODE == 2) {   // the product's formulation itself (merkle.hpp Sha256Pair: partners lane ^ 7, bank-masked DPP adds)
      msmerkle::Sha256Pair hp; hp.init(lane);
      hp.reset();
      hp.compress(w);
      hp.template compress_pad_block<512u>();
      __syncthreads();
      msmerkle::uint4_t o; o.x = hp.cv[0]; o.y = hp.cv[1]; o.z = hp.cv[2]; o.w = hp.cv[3];
      reinterpret_cast<msmerkle::uint4_t*>(slot)[A ? 0 : 1] = o;
    } else if (PAIR) {
      u32 cv[4];
      cv[0] = A ? 0x6a09e667u : 0x510e527fu; cv[1] = A ? 0xbb67ae85u : 0x9b05688cu; cv[2] = A ? 0x3c6ef372u : 0x1f83d9abu; cv[3] = A ? 0xa54ff53au : 0x5be0cd19u;
      pair_compress<MODE>(cv, w, c);
      pair_compress_pad<MODE, 512u>(cv, c);
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
  if (!A) for (int i = 0; i < 8; i++) out[node * 8 + i] = slot[i];
}
int main() {
  u32 *d0, *d1, *ds; hipMalloc(&d0, 64 * 8 * 4); hipMalloc(&d1, 64 * 8 * 4); hipMalloc(&ds, 64);
  u32 hs[16]; for (int i = 0; i < 16; i++) hs[i] = 0x9e3779b9u * (i + 1);
  hipMemcpy(ds, hs, 64, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  float ms[3];
  u32* d2; hipMalloc(&d2, 64 * 8 * 4);
  const char* names[3] = {"one lane per node                                     ", "lane pair (neighbours; two selects + one DPP add)     ", "lane pair (merkle.hpp Sha256Pair: half-mirror, asm)   "};
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) chain<0><<<1, 64>>>(d0, ds, iters); else if (mode == 1) chain<1><<<1, 64>>>(d1, ds, iters); else chain<2><<<1, 64>>>(d2, ds, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("%s: %.3f us per node of a dependent chain in a lone wave (%.3f of one lane per node)\n", names[mode], ms[mode] * 1e3 / iters, ms[mode] / ms[0]);
  }
  u32 h0[64 * 8], h1[64 * 8], h2[64 * 8];
  hipMemcpy(h0, d0, sizeof h0, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, sizeof h2, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int i = 0; i < 32 * 8; i++) { bad += h0[i] != h1[i]; bad2 += h0[i] != h2[i]; }   // the pair kernels have 32 nodes: they equal nodes 0..31 of the other
  printf("digests: neighbours %s (%d of 256 words differ), half-mirror %s (%d)\n", bad ? "DIFFER" : "agree", bad, bad2 ? "DIFFER" : "agree", bad2);
  bad += bad2;
  return bad != 0;
}
