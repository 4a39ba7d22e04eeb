// VALU issue-rate micro-benchmark (MI355X): wave-instructions per cycle per SIMD for the ops SHA-256 and
// Goldilocks arithmetic are built from.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
template <int OP> __global__ void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
  uint32_t b = seed ^ 0x9e3779b9u, c = seed * 3 + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {  // 8 independent chains per thread
      if (OP == 0) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 1) asm volatile("v_alignbit_b32 %0, %1, %1, 7" : "=v"(a[i]) : "v"(a[i]));
      if (OP == 2) asm volatile("v_lshl_or_b32 %0, %1, 25, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));  // v_lshl_or_b32
      if (OP == 3) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
      if (OP == 5) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
      if (OP == 6) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 7) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 8) { uint64_t t = (uint64_t)a[i] * b + c; a[i] = (uint32_t)(t >> 32) ^ (uint32_t)t; }  // v_mad_u64_u32 (+xor)
      if (OP == 9) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 10) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
      if (OP == 11) asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 12) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : );
      if (OP == 13) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 14) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (OP == 16) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[i]));
      if (OP == 17) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 18) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 19) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
      if (OP == 20) asm volatile("v_sub_co_u32 %0, s[10:11], %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "s10", "s11");
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> __global__ void k64(uint32_t* out, uint32_t seed) {
  uint64_t a[8];
  for (int i = 0; i < 8; i++) a[i] = ((uint64_t)(seed + threadIdx.x * 8 + i) << 32) | (seed * 7 + i);
  uint64_t b = ((uint64_t)(seed ^ 0x9e3779b9u) << 32) | seed, c = seed * 3 + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
      if (OP == 1) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 2) asm volatile("v_lshlrev_b64 %0, 5, %1" : "=v"(a[i]) : "v"(a[i]));
      if (OP == 3) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %3" : "=v"(a[i]) : "v"((uint32_t)a[i]), "v"((uint32_t)b), "v"(c) : "s10", "s11");
      if (OP == 4) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (OP == 5) asm volatile("v_lshrrev_b64 %0, 7, %1" : "=v"(a[i]) : "v"(a[i]));
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int OP> void run64(const char* name, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256;
  k64<OP><<<blocks, threads>>>(d, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k64<OP><<<blocks, threads>>>(d, 2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * threads / 64 * ITERS * 8;
  double per_simd_per_us = winstr / 1024 / (ms * 1e3);
  printf("%-16s %8.3f ms  %.1f wave-instr/us/SIMD  (cycles per wave-instr at 2.4 GHz: %.2f)\n", name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
}
template <int OP> void run(const char* name, uint32_t* d, int extra) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  k<OP><<<blocks, threads>>>(d, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(d, 2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * threads / 64 * ITERS * 8 * (1 + extra);
  double per_simd_per_us = winstr / 1024 / (ms * 1e3);
  printf("%-16s %8.3f ms  %.1f wave-instr/us/SIMD  (cycles per wave-instr at 2.4 GHz: %.2f)\n", name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_xor", d, 0); run<1>("v_alignbit", d, 0); run<2>("v_lshl_or", d, 0); run<3>("v_add3", d, 0); run<4>("v_bfi", d, 0);
  run<5>("v_bitop3", d, 0); run<6>("v_mul_lo_u32", d, 0); run<7>("v_mul_hi_u32", d, 0); run<8>("v_mad_u64_u32+xor", d, 1);
  run<9>("v_add_u32", d, 0); run<10>("v_perm", d, 0); run<11>("v_alignbyte", d, 0);
  run<12>("v_cndmask(vcc)", d, 0); run<13>("v_add_co_u32", d, 0); run<14>("v_addc_co_u32", d, 0); run<15>("v_mov_b32", d, 0); run<16>("v_lshrrev_b32", d, 0);
  run<17>("v_cmp_lt_u32", d, 0); run<18>("v_mul_u32_u24", d, 0); run<19>("v_mad_u32_u24", d, 0); run<20>("v_sub_co(sgpr)", d, 0);
  run64<0>("v_lshl_add_u64", d); run64<1>("v_cmp_lt_u64", d); run64<2>("v_lshlrev_b64", d); run64<3>("v_mad_u64_u32", d); run64<4>("v_mov_b64", d); run64<5>("v_lshrrev_b64", d);
  return 0;
}
