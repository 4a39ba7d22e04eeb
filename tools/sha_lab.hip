// SHA-256 issue-rate lab (MI355X): the compression function of csrc/merkle.hpp on register-resident data, no memory traffic, at
// 1..8 waves per SIMD - what a compression costs when nothing but the VALU is in the way, to compare with the hash kernels
// (LeafHashKernel: 3601 instructions in 4.14 clocks each).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mini-stark_amd/csrc tools/sha_lab.hip -o /tmp/sha_lab && /tmp/sha_lab
#include "merkle.hpp"
#include <cstdio>
template <bool PAD> __global__ __launch_bounds__(256) void k(u32* out, u32 seed, int iters) {
  extern __shared__ unsigned char lds[];   // only to limit the workgroups per CU
  msmerkle::Sha256 h; h.init();
  u32 w[16];
  for (int i = 0; i < 16; i++) w[i] = seed * 2654435761u + threadIdx.x * 16 + i + blockIdx.x;
  for (int it = 0; it < iters; it++) {
    h.compress(w);                                       // clobbers w with the last 16 schedule words: the next message
    if (PAD) h.compress_pad_block<512>();                // an inner node: 64-byte message + its constant padding block
    w[0] ^= h.st[0]; w[5] ^= h.st[3]; w[9] ^= h.st[6];
  }
  u32 s = 0; for (int i = 0; i < 8; i++) s ^= h.st[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 9999) out[0] = lds[0];
}
int main() {
  u32* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  for (int pad = 0; pad < 2; pad++)
    for (int wps = 1; wps <= 8; wps++) {   // waves per SIMD = workgroups (4 waves) per CU
      const int blocks = 256 * wps;
      const size_t lds = (160 * 1024 / wps) - 1024;       // at most wps workgroups fit a CU
      auto launch = [&](u32 seed) {
        if (pad) { hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<true><<<blocks, 256, lds>>>(d, seed, iters); }
        else { hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<false><<<blocks, 256, lds>>>(d, seed, iters); }
      };
      launch(1); hipDeviceSynchronize();
      hipEventRecord(e0); launch(2); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double comps_per_simd = (double)wps * iters * (pad ? 2 : 1);       // wave-compressions per SIMD
      const double us_per = ms * 1e3 / comps_per_simd;
      printf("%s  %d waves/SIMD  %8.3f ms  %.3f us per wave-compression per SIMD = %.0f clocks at 2.2 GHz  (chip: %.1f G compressions/s)\n",
             pad ? "message + padding block" : "message block only     ", wps, ms, us_per, us_per * 2200.0, 1024.0 * 64 / us_per / 1e3);
    }
  return 0;
}
