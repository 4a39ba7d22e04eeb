// Host-visible latency of the small steps a FRI round is made of (single proof in flight): what one stage call costs
// before any arithmetic.  hipcc --offload-arch=gfx950 -O2 tools/latency_probe.hip -o tools/latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty(uint32_t* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_write(uint32_t* dst, uint32_t v) { if (threadIdx.x < 8) dst[threadIdx.x] = v + threadIdx.x; }
__global__ void k_flag(volatile uint32_t* dst, uint32_t v) { if (threadIdx.x < 8) dst[threadIdx.x] = v + threadIdx.x; __threadfence_system(); if (threadIdx.x == 0) dst[16] = v; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *d, *h;
  CK(hipMalloc(&d, 4096)); CK(hipHostMalloc(&h, 4096, hipHostMallocDefault));
  const int N = 2000;
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 50; i++) body(i);
    double t0 = now();
    for (int i = 0; i < N; i++) body(i);
    printf("%-70s %7.2f us\n", name, (now() - t0) / N);
    return 0;
  };
  run("1 kernel + hipStreamSynchronize", [&](int) { k_empty<<<1, 64, 0, s>>>(d); (void)hipStreamSynchronize(s); });
  run("5 dependent kernels + sync", [&](int) { for (int j = 0; j < 5; j++) k_empty<<<1, 64, 0, s>>>(d); (void)hipStreamSynchronize(s); });
  run("12 dependent kernels + sync", [&](int) { for (int j = 0; j < 12; j++) k_empty<<<1, 64, 0, s>>>(d); (void)hipStreamSynchronize(s); });
  run("1 kernel + 32-byte hipMemcpyAsync D2H (pinned) + sync", [&](int i) { k_write<<<1, 64, 0, s>>>(d, i); (void)hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
  run("1 kernel storing 32 bytes to pinned host memory + sync", [&](int i) { k_write<<<1, 64, 0, s>>>(h, i); (void)hipStreamSynchronize(s); });
  run("1 kernel storing to pinned host memory, host polls a flag (no sync)", [&](int i) {
    k_flag<<<1, 64, 0, s>>>(h, (uint32_t)i + 1);
    volatile uint32_t* f = h + 16; while (*f != (uint32_t)i + 1) {}
  });
  run("1 kernel + spin on hipStreamQuery", [&](int) { k_empty<<<1, 64, 0, s>>>(d); while (hipStreamQuery(s) == hipErrorNotReady) {} });
  run("5 dependent kernels, the last stores a flag the host polls (no sync)", [&](int i) {
    for (int j = 0; j < 4; j++) k_empty<<<1, 64, 0, s>>>(d);
    k_flag<<<1, 64, 0, s>>>(h, (uint32_t)i + 1);
    volatile uint32_t* f = h + 16; while (*f != (uint32_t)i + 1) {}
  });
  run("5 dependent kernels + spin on hipStreamQuery", [&](int) { for (int j = 0; j < 5; j++) k_empty<<<1, 64, 0, s>>>(d); while (hipStreamQuery(s) == hipErrorNotReady) {} });
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  run("1 kernel + event record + hipEventSynchronize", [&](int) { k_empty<<<1, 64, 0, s>>>(d); (void)hipEventRecord(ev, s); (void)hipEventSynchronize(ev); });
  run("32-byte hipMemcpyAsync H2D (pinned) + 1 kernel + sync", [&](int) { (void)hipMemcpyAsync(d, h, 32, hipMemcpyHostToDevice, s); k_empty<<<1, 64, 0, s>>>(d); (void)hipStreamSynchronize(s); });
  run("kernel with 64 B of by-value arguments only (no H2D) + sync", [&](int i) { k_write<<<1, 64, 0, s>>>(d, i); (void)hipStreamSynchronize(s); });
  CK(hipStreamSynchronize(s));
  return 0;
}
