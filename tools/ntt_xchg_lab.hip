// ntt_xchg_lab.hip - VERDICT r3 #6: what does an INTRA-WAVE exchange in front of the last radix-8 sub-round of the NTT tiles cost against the LDS round trip + s_barrier
// it would replace?  One workgroup = the later pass's tile: 512 threads, 2^10 rows x 8 columns of Goldilocks elements (64 KiB of LDS + the 8 KiB w_r table), two workgroups
// per CU.  Every thread holds two radix-8 items (16 elements) in registers and repeats  [exchange -> radix-8 DIF -> 7 boundary twiddle multiplications per item]:
//   lds      the product's scheme: the 16 elements go to the tile (row swizzle of ntt.hpp), one workgroup barrier, the thread reads the 16 elements of its next items
//   swizzle  the 8 partners of an item sit in 8 adjacent lanes: 8 x 8 transpose of 64-bit values in three ds_swizzle stages (xor 4, 2, 1) - no LDS allocation, no barrier
//   none     no exchange at all (the arithmetic alone: what both schemes sit on top of)
// Reports ns per launch and clocks per element and sub-round; the instruction counts are read off the ISA.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../mini-stark_amd/csrc/ntt.hpp"

typedef unsigned long long u64_;
constexpr int TH = 512, K = 10, C = 8, R = 1 << K;

__device__ __forceinline__ int prow(int row) { return row ^ ((row >> 3) & 3); }

template <int MASK> __device__ __forceinline__ u64 swz(u64 v) {
  const int lo = __builtin_amdgcn_ds_swizzle((int)(u32)v, (MASK << 10) | 0x1F), hi = __builtin_amdgcn_ds_swizzle((int)(u32)(v >> 32), (MASK << 10) | 0x1F);
  return ((u64)(u32)hi << 32) | (u32)lo;
}
template <int MASK> __device__ __forceinline__ void xstage(u64 (&x)[8], bool bit) {
#pragma unroll
  for (int r = 0; r < 8; r++) {
    if (r & MASK) continue;
    const u64 send = bit ? x[r] : x[r | MASK];
    const u64 recv = swz<MASK>(send);
    if (bit) x[r] = recv; else x[r | MASK] = recv;
  }
}
__device__ __forceinline__ void transpose8(u64 (&x)[8], int lane) {
  xstage<4>(x, (lane & 4) != 0); xstage<2>(x, (lane & 2) != 0); xstage<1>(x, (lane & 1) != 0);
}

template <int MODE> __global__ void __launch_bounds__(TH, 4) lab(u64* out, const u64* in, const u64* wtab, int iters) {
  extern __shared__ __align__(16) unsigned char lds[];
  u64* tile = reinterpret_cast<u64*>(lds);
  u64* w = tile + R * C;
  const int tid = threadIdx.x;
  for (int i = tid; i < R; i += TH) w[i] = wtab[i];
  u64 x[2][8];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int t = 0; t < 8; t++) x[j][t] = in[((size_t)blockIdx.x * TH + tid) * 16 + j * 8 + t] % GL::P;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {   // the product's exchange: sub-round 1 layout (rows R0 + t*8) -> tile -> barrier -> sub-round 2 layout (rows g*8 + t)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int item = tid + j * TH, c = item & 7, g = item >> 3, lo = g & 7, R0 = ((g >> 3) << 6) | lo;
#pragma unroll
        for (int t = 0; t < 8; t++) tile[prow(R0 + t * 8) * C + c] = x[j][t];
      }
      msrt::wg_barrier();
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int item = tid + j * TH, c = item & 7, g = item >> 3;
#pragma unroll
        for (int t = 0; t < 8; t++) x[j][t] = tile[prow(g * 8 + t) * C + c];
      }
      msrt::wg_barrier();   // (the product needs this one too: the tile is rewritten by the next exchange / the next item's load)
    } else if (MODE == 1) {
      transpose8(x[0], tid & 63); transpose8(x[1], tid & 63);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      msntt::dif_regs<GLM, false, 3>(x[j], w, K);
      const int lo = (tid >> 3) & 7;
#pragma unroll
      for (int e = 1; e < 8; e++) x[j][e] = GLM::mul_tw(x[j][e], w[(e * lo) << 4]);   // the boundary's twiddle multiplications (general form: table multiply)
    }
  }
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int t = 0; t < 8; t++) out[((size_t)blockIdx.x * TH + tid) * 16 + j * 8 + t] = x[j][t];
}
// the swizzle transpose really is a transpose: lane L of a group of 8 ends up with register L of lanes 0..7
__global__ void check(int* bad) {
  u64 x[8];
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < 8; r++) x[r] = (u64)lane * 8 + r;
  transpose8(x, lane);
  for (int r = 0; r < 8; r++) if (x[r] != (u64)((lane & ~7) + r) * 8 + (lane & 7)) atomicAdd(bad, 1);
}

template <int MODE> double run(const char* name, u64* d_out, u64* d_in, u64* d_w, int blocks, int iters) {
  const size_t ldsb = (size_t)(R * C + R) * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&lab<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  lab<MODE><<<blocks, TH, ldsb>>>(d_out, d_in, d_w, iters);
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) lab<MODE><<<blocks, TH, ldsb>>>(d_out, d_in, d_w, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double elems = (double)blocks * TH * 16 * iters;
  // clocks per element and sub-round on one SIMD: 1024 SIMDs x 2.4 GHz
  printf("%-44s blocks %5d iters %4d  %8.3f ms   %6.2f ps per element-subround   %6.2f SIMD-clocks per wave-element-subround\n", name, blocks, iters, ms, ms * 1e9 / elems,
         ms * 1e-3 * 2.4e9 * 1024 / (elems / 64));
  return ms;
}

int main() {
  const int blocks = 512 * 4, iters = 64;
  std::vector<u64> h((size_t)blocks * TH * 16), hw(R);
  u64 s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
  for (auto& v : hw) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % GL::P; }
  u64 *d_in, *d_out, *d_w; int* d_bad;
  hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, h.size() * 8); hipMalloc(&d_w, R * 8); hipMalloc(&d_bad, 4);
  hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_w, hw.data(), R * 8, hipMemcpyHostToDevice); hipMemset(d_bad, 0, 4);
  check<<<4, 256>>>(d_bad);
  int bad = -1; hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
  printf("swizzle transpose check: %s\n", bad == 0 ? "ok" : "WRONG");
  for (int rep = 0; rep < 2; rep++) {
    const double t2 = run<2>("none    (radix-8 + 7 twiddle multiplications)", d_out, d_in, d_w, blocks, iters);
    const double t0 = run<0>("lds     (tile round trip + 2 barriers)", d_out, d_in, d_w, blocks, iters);
    const double t1 = run<1>("swizzle (8x8 transpose, ds_swizzle x 3 stages)", d_out, d_in, d_w, blocks, iters);
    printf("exchange alone: lds %.3f ms, swizzle %.3f ms per launch (arithmetic %.3f ms)\n", t0 - t2, t1 - t2, t2);
  }
  printf("hip status %d\n", (int)hipGetLastError());
  return 0;
}
