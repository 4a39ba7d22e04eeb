// Memory-pattern probe for the NTT later pass (MI355X): a persistent grid of 512 x 512 threads copies `cols` columns of 2^23 u64 as
// tiles of 1024 row pieces of 64 bytes, the rows `rs` elements apart on the read side and `ws` elements apart on the write side
// (8192 = the natural 64 KiB pitch of the two-pass 2^23-point plan; 8 = contiguous).  No LDS, no arithmetic: what is timed is the
// access pattern alone, so that a padded pitch / a column-at-a-time order can be judged before the kernels are touched.
//   hipcc --offload-arch=gfx950 -O3 tools/stride_probe.hip -o /tmp/stride_probe && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint64_t u64;
typedef u64 V16 __attribute__((vector_size(16)));
struct P { const u64* src; u64* dst; size_t rs, ws, rpitch_tile, wpitch_tile, col_r, col_w; int tiles, cols, xcd; };
__global__ __launch_bounds__(512) void k(P p) {
  const int tid = threadIdx.x, c0 = (tid & 3) * 2, rb = tid >> 2;
  const size_t total = (size_t)p.tiles * p.cols;
  const int nbx = gridDim.x, bx = blockIdx.x;
  const size_t stride = p.xcd ? (nbx >> 3) : nbx, first = p.xcd ? (bx >> 3) : bx, lim = p.xcd ? (total >> 3) : total, base = p.xcd ? (size_t)(bx & 7) * (total >> 3) : 0;
  for (size_t it = first; it < lim; it += stride) {
    const size_t g = base + it, by = g / p.tiles, tl = g - by * p.tiles;
    const u64* s = p.src + by * p.col_r + tl * p.rpitch_tile + c0 + (size_t)rb * p.rs;
    u64* d = p.dst + by * p.col_w + tl * p.wpitch_tile + c0 + (size_t)rb * p.ws;
    V16 r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = *reinterpret_cast<const V16*>(s + (size_t)(i * 128) * p.rs);
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<V16*>(d + (size_t)(i * 128) * p.ws) = r[i];
  }
}
static float run(P p, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) k<<<512, 512>>>(p);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) k<<<512, 512>>>(p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1000.f;
}
int main() {
  const size_t n = (size_t)1 << 23, maxpad = 4096;
  const int maxcols = 6;
  const size_t cap = maxcols * (n + 1024 * maxpad) + 65536;
  u64 *a, *b;
  if (hipMalloc(&a, cap * 8) != hipSuccess || hipMalloc(&b, cap * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, cap * 8); hipMemset(b, 0, cap * 8);
  auto report = [&](const char* name, P p) {
    float us = run(p, 20);
    double bytes = 2.0 * p.cols * n * 8;
    printf("%-64s cols %d  %8.1f us  %6.2f TB/s  (%.1f us per column)\n", name, p.cols, us, bytes / us / 1e6, us / p.cols);
    fflush(stdout);
  };
  for (int cols : {6, 3, 1}) {
    // strided both sides: tile t = 64-byte pieces at offset 8 t of rows `pitch` apart
    for (size_t pad : {(size_t)0, (size_t)8, (size_t)16, (size_t)32, (size_t)64, (size_t)128, (size_t)512, (size_t)2048}) {
      char nm[128];
      P p{a, b, 8192 + pad, 8192, 8, 8, n + 1024 * pad, n, 1024, cols, 1};
      snprintf(nm, sizeof nm, "read pitch 64 KiB + %zu B, write pitch 64 KiB", pad * 8); report(nm, p);
    }
    { P p{a, b, 8192 + 32, 8192 + 32, 8, 8, n + 1024 * 32, n + 1024 * 32, 1024, cols, 1}; report("read and write pitch 64 KiB + 256 B", p); }
    { P p{a, b, 8, 8192, 8192, 8, n, n, 1024, cols, 1}; report("read contiguous, write pitch 64 KiB", p); }
    { P p{a, b, 8, 8192 + 32, 8192, 8, n, n + 1024 * 32, 1024, cols, 1}; report("read contiguous, write pitch 64 KiB + 256 B", p); }
    { P p{a, b, 8192, 8, 8, 8192, n, n, 1024, cols, 1}; report("read pitch 64 KiB, write contiguous", p); }
    { P p{a, b, 8192 + 32, 8, 8, 8192, n + 1024 * 32, n, 1024, cols, 1}; report("read pitch 64 KiB + 256 B, write contiguous", p); }
    { P p{a, b, 8, 8, 8192, 8192, n, n, 1024, cols, 1}; report("both contiguous", p); }
    { P p{a, b, 8192, 8192, 8, 8, n, n, 1024, cols, 0}; report("both 64 KiB pitch, round-robin tile walk (no XCD ranges)", p); }
  }
  // in place (the write goes where the read came from): the working set of a column halves
  { P p{a, a, 8192, 8192, 8, 8, n, n, 1024, 6, 1}; report("in place, both 64 KiB pitch", p); }
  { P p{a, a, 8192, 8192, 8, 8, n, n, 1024, 1, 1}; report("in place, both 64 KiB pitch", p); }
  return 0;
}
