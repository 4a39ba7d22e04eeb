// rt.hpp — thin runtime layer under the kernels.
//
// Kernels are written as structs with a Params block and a `phase()` function:
// every workgroup barrier sits BETWEEN phases and no per-thread state survives
// a phase.  On the GPU `ms_kmain<K>` runs the phases with `__syncthreads()`
// between them.  With -DMS_EMU (tests/emu only — never shipped, never loaded by
// the product) the same phase code is executed thread by thread on the CPU so
// kernel logic can be checked against the oracle without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(MS_HOST_ONLY)
// host code that only needs the field arithmetic of field.hpp (the CPU verifier of mini-stark_amd/host/stark_host.cpp)
#define MS_HD inline
#define MS_DEV inline
#define MS_RESTRICT
MS_HD uint64_t ms_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

#elif defined(MS_EMU)
#include <cstdlib>
#include <vector>
#define MS_HD inline
#define MS_DEV inline
#define MS_RESTRICT
namespace msrt {
struct Stream { int dummy; };
inline int malloc_dev(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
inline int free_dev(void* p) { std::free(p); return 0; }
inline int malloc_host(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
inline int malloc_host_coherent(void** p, size_t n) { return malloc_host(p, n); }
inline int free_host(void* p) { std::free(p); return 0; }
inline int h2d(void* d, const void* h, size_t n, Stream*) { std::memcpy(d, h, n); return 0; }
inline int d2h(void* h, const void* d, size_t n, Stream*) { std::memcpy(h, d, n); return 0; }
inline int d2d(void* d, const void* s, size_t n, Stream*) { std::memmove(d, s, n); return 0; }
inline int memset_dev(void* d, int v, size_t n, Stream*) { std::memset(d, v, n); return 0; }
inline int sync(Stream*) { return 0; }
inline int sync_flag(Stream*, const unsigned long long* f, unsigned long long v) { return *f == v ? 0 : 999; }   // emulated launches are synchronous: the flag is up or it never will be
inline int stream_create(Stream** s) { *s = new Stream(); return 0; }
inline int stream_destroy(Stream* s) { delete s; return 0; }
inline int set_device(int) { return 0; }
inline const char* last_error_string() { return "emu"; }
struct Event { int dummy; };
inline int event_create(Event** e) { *e = new Event(); return 0; }
inline int event_destroy(Event* e) { delete e; return 0; }
inline int event_record(Event*, Stream*) { return 0; }
inline int event_elapsed_ms(float* ms, Event*, Event*) { *ms = 0.f; return 0; }
inline int stream_wait_event(Stream*, Event*) { return 0; }
inline int event_sync(Event*) { return 0; }
inline bool is_pinned_host(const void*) { const char* e = std::getenv("MS_EMU_SDMA"); return e && *e; }   // (the scripted engine of the failure-handling tests takes any host memory)
template <class K>
inline int launch(Stream*, unsigned gx, unsigned gy, int threads, size_t lds_bytes, const typename K::Params& p) {
  std::vector<unsigned char> lds(lds_bytes + 16);
  int nph = K::nphases(p);
  for (unsigned by = 0; by < gy; by++)
    for (unsigned bx = 0; bx < gx; bx++)
      for (int ph = 0; ph < nph; ph++)
        for (int t = 0; t < threads; t++) K::phase(ph, p, (int)bx, (int)by, t, threads, lds.data());
  return 0;
}
// no RCCL in the emulation build: ms_set_shard_rccl reports MS_ERR_HIP, the gloo tests use the exchange callback
struct Rccl {
  struct UniqueId { char internal[128]; };
  void* lib = nullptr;
  int (*get_unique_id)(void*) = nullptr; int (*comm_init_rank)(void**, int, UniqueId, int) = nullptr; int (*comm_destroy)(void*) = nullptr;
  int (*group_start)() = nullptr; int (*group_end)() = nullptr;
  int (*send)(void*, size_t, int, int, void*, Stream*) = nullptr; int (*recv)(void*, size_t, int, int, void*, Stream*) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, Stream*) = nullptr; int (*all_reduce)(const void*, void*, size_t, int, int, void*, Stream*) = nullptr;
  const char* (*err_string)(int) = nullptr;
  static Rccl& get() { static Rccl r; return r; }
  int load() { return 1; }
};
// No copy engines in the emulation build: the read-back is the plain copy - unless MS_EMU_SDMA is set (tests of the boundary's failure handling): a scripted
// engine whose copies are memcpy's.  "ok": every copy completes; "hang": copies are accepted and never complete (wait() reports a timeout at once; an unlimited
// wait - ms_destroy - returns, or the test itself would hang); "fail": copies are accepted and complete with a NEGATIVE signal (what hsa_ext_amd.h documents
// for a failed asynchronous copy), nothing is written; "hang-upload" / "fail-upload": the same for host-to-device copies only.
struct Sdma {
  struct Signal { unsigned long long handle = 0; };
  long long sig_val[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned nsig = 0;
  static Sdma& get() { static Sdma s; return s; }
  static const char* mode() { const char* e = std::getenv("MS_EMU_SDMA"); return (e && *e) ? e : nullptr; }
  static bool is(const char* m) { const char* e = mode(); return e && !std::strcmp(e, m); }
  int load() { return mode() ? 0 : 1; }
  int bind_device(int, int* g) { if (!mode()) return 1; *g = 0; return 0; }
  int signal_create(Signal* s) { if (!mode()) return 1; s->handle = 1 + (nsig++ & 7); return 0; }
  void signal_destroy(Signal) {}
  int copy_d2h(int, void* d, const void* s_, size_t n, Signal sg, unsigned) {
    if (!mode()) return 1;
    if (is("hang")) { sig_val[sg.handle - 1] = 1; return 0; }
    if (is("fail")) { sig_val[sg.handle - 1] = -1; return 0; }
    std::memcpy(d, s_, n); sig_val[sg.handle - 1] = 0; return 0;
  }
  int copy_h2d(int, void* d, const void* s_, size_t n, Signal sg, unsigned) {
    if (!mode()) return 1;
    if (is("hang") || is("hang-upload")) { sig_val[sg.handle - 1] = 1; return 0; }
    if (is("fail") || is("fail-upload")) { sig_val[sg.handle - 1] = -1; return 0; }
    std::memcpy(d, s_, n); sig_val[sg.handle - 1] = 0; return 0;
  }
  unsigned h2d_engine(int) { return mode() ? 1 : 0; }
  unsigned d2h_engine(int) { return mode() ? 1 : 0; }
  // 0: complete; 1: not within `seconds` (seconds < 0: no limit); -1: the copy FAILED
  int wait(Signal sg, double seconds) { const long long v = sig_val[sg.handle - 1]; if (v == 0) return 0; if (v < 0) return -1; return seconds < 0 ? 0 : 1; }
  bool unavailable() const { return !mode(); }
  const char* runtime_path() const { return mode() ? "emulated copy engine (MS_EMU_SDMA)" : ""; }
};
}  // namespace msrt
// ---- cooperative kernels (K::run with workgroup barriers INSIDE the function, per-thread state alive across them): every thread of a
// workgroup is a fiber (its own stack; fiber_switch below), resumed round-robin by the scheduler.  wg_barrier() is a real barrier over the fibers that have not
// returned (as s_barrier is over the waves that have not ended); wave_shfl_xor() & co. are rendezvous of a lane with its PARTNER lane inside
// each group of 64 consecutive threads (the emulated wave is 64 lanes wide, like the hardware's), so lanes that sit out a stretch of
// exchanges - the idle lanes of the upper tree levels - do not have to mirror their partners' calls (r05; until then both were plain
// yields, correct only while every fiber made every call).
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif
namespace msrt {
// Stack switch of the fibers (x86-64 SysV): callee-saved registers on the old stack, stack pointers exchanged, `ret` into the new one.  (glibc's swapcontext does the
// same plus one sigprocmask system call per switch; since the lane-pair exchanges of the tree levels a proof switches fibers millions of times and the system calls were
// a third of the CPU suite's time.)
#if !defined(__x86_64__)
#error "the kernel-emulation build switches fiber stacks with x86-64 assembly"
#endif
__attribute__((naked, noinline, unused)) static void fiber_switch(void** /*save_sp: rdi*/, void* /*to_sp: rsi*/) {
  asm volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
               "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
               "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
struct FiberBlock {
  static constexpr size_t STACK = 96 << 10;
  std::vector<void*> sp; std::vector<unsigned char> stacks; std::vector<char> done; std::vector<unsigned long long> xchg; std::vector<unsigned> shseq, rdseq; std::vector<int> shpart;
  void* sched_sp = nullptr; int cur = -1, nthreads = 0;
  int alive = 0, bar_arrived = 0; unsigned bar_gen = 0;
  void (*entry)(void*, int) = nullptr; void* arg = nullptr;
  static FiberBlock*& active() { static thread_local FiberBlock* a = nullptr; return a; }
  // (AddressSanitizer keeps per-thread stack bounds: it is told about every switch)
  void to_fiber(int t) {
#if defined(__SANITIZE_ADDRESS__)
    void* fake = nullptr; __sanitizer_start_switch_fiber(&fake, stacks.data() + (size_t)t * STACK, STACK);
    fiber_switch(&sched_sp, sp[t]);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    fiber_switch(&sched_sp, sp[t]);
#endif
  }
  void to_sched(int t, bool last) {
#if defined(__SANITIZE_ADDRESS__)
    void* fake = nullptr; __sanitizer_start_switch_fiber(last ? nullptr : &fake, sched_stack_bottom, sched_stack_size);
    fiber_switch(&sp[t], sched_sp);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    (void)last; fiber_switch(&sp[t], sched_sp);
#endif
  }
#if defined(__SANITIZE_ADDRESS__)
  const void* sched_stack_bottom = nullptr; size_t sched_stack_size = 0;
#endif
  static void trampoline() {
    FiberBlock* b = active(); const int t = b->cur;
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(nullptr, &b->sched_stack_bottom, &b->sched_stack_size);
#endif
    b->entry(b->arg, t); b->done[t] = 1; b->alive--;
    b->to_sched(t, true);
    __builtin_unreachable();
  }
  void yield() { to_sched(cur, false); }
  void barrier() {
    const unsigned g = bar_gen;
    bar_arrived++;
    for (;;) {
      if (bar_gen != g) return;
      if (bar_arrived >= alive) { bar_arrived = 0; bar_gen++; return; }   // the last to arrive (or the others have returned) opens it
      yield();
    }
  }
  void run_block(int threads, void (*fn)(void*, int), void* a) {
    nthreads = threads; entry = fn; arg = a;
    if ((int)sp.size() < threads) { sp.resize(threads); stacks.resize((size_t)threads * STACK + 64); }
    done.assign(threads, 0); xchg.assign((size_t)threads * 2, 0); shseq.assign(threads, 0); rdseq.assign(threads, 0); shpart.assign((size_t)threads * 2, 0);
    alive = threads; bar_arrived = 0; bar_gen = 0;
    for (int t = 0; t < threads; t++) {   // a fresh stack: six zero registers under the address the first switch returns to; the entry sees rsp = 8 (mod 16), as after a call
      uintptr_t top = (reinterpret_cast<uintptr_t>(stacks.data()) + (size_t)(t + 1) * STACK) & ~(uintptr_t)15;
      void** w = reinterpret_cast<void**>(top);
      *--w = nullptr;
      *--w = reinterpret_cast<void*>(&trampoline);
      for (int k = 0; k < 6; k++) *--w = nullptr;
      sp[t] = w;
    }
    FiberBlock* prev = active(); active() = this;
    for (;;) {
      bool any = false;
      for (int t = 0; t < threads; t++) if (!done[t]) { any = true; cur = t; to_fiber(t); }
      if (!any) break;
    }
    cur = -1; active() = prev;
  }
};
inline void wg_barrier() { FiberBlock::active()->barrier(); }
inline void wg_barrier_global() { FiberBlock::active()->barrier(); }   // (the emulated workgroup's "global" memory is plain host memory: always coherent)
inline void fence_device() {}
inline int wave_uniform(int v) { return v; }
// value of lane (lane ^ mask) of the caller's 64-lane wave; the partner lane must make the matching call (its k-th exchange with the caller's k-th: the lanes of a
// wave run the same instruction stream).  Two slots per lane, written alternately: before a lane reuses a slot it waits until the partner of the exchange that used
// it (two exchanges back; the partner changes from call to call in a butterfly) has read it.  A partner that has returned gives the caller's own value.
inline unsigned long long wave_shfl_xor(unsigned long long v, int mask) {
  FiberBlock* b = FiberBlock::active(); const int t = b->cur;
  const int pt = (t & ~63) | ((t ^ mask) & 63);
  if (pt >= b->nthreads) return v;
  const unsigned k = b->shseq[t];
  const size_t slot = (size_t)t * 2 + (k & 1);
  if (k >= 2) { const int old = b->shpart[slot]; while (b->rdseq[old] < k - 1 && !b->done[old]) b->yield(); }
  b->xchg[slot] = v; b->shpart[slot] = pt; b->shseq[t] = k + 1;
  while (b->shseq[pt] < k + 1) { if (b->done[pt]) { b->rdseq[t] = k + 1; return v; } b->yield(); }
  const unsigned long long r = b->xchg[(size_t)pt * 2 + (k & 1)];
  b->rdseq[t] = k + 1;
  return r;
}
template <class K> struct CoopArgs { const typename K::Params* p; int bx, by, nbx, threads; unsigned char* lds; };
template <class K> inline void coop_entry(void* a, int tid) { auto* c = (CoopArgs<K>*)a; K::run(*c->p, c->bx, c->by, c->nbx, tid, c->lds); }
template <class K>
inline int launch_coop(Stream*, unsigned gx, unsigned gy, int threads, size_t lds_bytes, const typename K::Params& p) {
  static thread_local FiberBlock fb;
  std::vector<unsigned char> lds(lds_bytes + 16);
  for (unsigned by = 0; by < gy; by++)
    for (unsigned bx = 0; bx < gx; bx++) {
      CoopArgs<K> a{&p, (int)bx, (int)by, (int)gx, threads, lds.data()};
      fb.run_block(threads, &coop_entry<K>, &a);
    }
  return 0;
}
MS_DEV void atomic_min_u64(unsigned long long* a, unsigned long long v) { if (v < *a) *a = v; }
MS_DEV void atomic_max_u64(unsigned long long* a, unsigned long long v) { if (v > *a) *a = v; }
MS_DEV unsigned atomic_add_u32(unsigned* a, unsigned v) { unsigned o = *a; *a = o + v; return o; }
// one slot of a device list per lane that wants one (call from all live lanes)
MS_DEV unsigned wave_alloc_slot(unsigned* counter, bool want) { if (!want) return 0; unsigned o = *counter; *counter = o + 1; return o; }
// true if the predicate holds on any live lane of the wave (emulation: a wave of one lane — results must not depend on it)
MS_DEV bool wave_any(bool pred) { return pred; }
// lane-pair exchanges of msmerkle::Sha256Pair (lane i and lane i ^ 7 of a group of eight; see the HIP forms below)
inline unsigned pair_swap(unsigned v) { return (unsigned)wave_shfl_xor(v, 7); }
inline unsigned pair_exchange_add(bool a_lane, unsigned v, unsigned t, unsigned s) {
  const unsigned long long r = wave_shfl_xor((unsigned long long)v | ((unsigned long long)t << 32), 7);
  return a_lane ? (unsigned)(r >> 32) + s : (unsigned)r + t;
}
inline unsigned rotr_var(unsigned x, unsigned r) { return (x >> r) | (x << ((32 - r) & 31)); }
struct HostFlag { unsigned long long* dst; unsigned long long val; };
MS_DEV void raise_host_flag(const HostFlag& f) { if (f.dst) *f.dst = f.val; }   // (an emulated launch has run to its end before the host looks)
inline void raise_host_flag_wg(const HostFlag& f, int tid) { if (f.dst && tid == 0) *f.dst = f.val; }
}  // namespace msrt
MS_HD uint64_t ms_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

#else  // ---------------------------------------------------------------- HIP (gfx950)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <ctime>
#include <vector>
#define MS_HD __host__ __device__ __forceinline__
#define MS_DEV __device__ __forceinline__
#define MS_RESTRICT __restrict__
namespace msrt {
typedef ihipStream_t Stream;
inline int malloc_dev(void** p, size_t n) { return (int)hipMalloc(p, n ? n : 1); }
inline int free_dev(void* p) { return (int)hipFree(p); }
inline int malloc_host(void** p, size_t n) { return (int)hipHostMalloc(p, n ? n : 1, hipHostMallocDefault); }
// ... fine-grained whatever HIP_HOST_COHERENT says: a kernel's stores are visible to the host while the kernel still runs (the context's staging area: polled results)
inline int malloc_host_coherent(void** p, size_t n) { return (int)hipHostMalloc(p, n ? n : 1, hipHostMallocCoherent); }
inline int free_host(void* p) { return (int)hipHostFree(p); }
inline int h2d(void* d, const void* h, size_t n, Stream* s) { return (int)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); }
inline int d2h(void* h, const void* d, size_t n, Stream* s) { return (int)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); }
inline int d2d(void* d, const void* s_, size_t n, Stream* s) { return (int)hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s); }
inline int memset_dev(void* d, int v, size_t n, Stream* s) { return (int)hipMemsetAsync(d, v, n, s); }
inline int sync(Stream* s) { return (int)hipStreamSynchronize(s); }
// host side of HostFlag (below): spin on the page-locked word for up to 2 ms, then fall back to the stream synchronisation (a long stage gains nothing from polling; a
// failed launch never raises the flag and reports through the stream)
inline int sync_flag(Stream* s, const unsigned long long* f, unsigned long long v) {
  const volatile unsigned long long* vf = f;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned i = 1;; i++) {
    if (*vf == v) { std::atomic_thread_fence(std::memory_order_acquire); return 0; }
    __builtin_ia32_pause();
    if (!(i & 1023u) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  const int e = (int)hipStreamSynchronize(s);
  if (e) return e;
  std::atomic_thread_fence(std::memory_order_acquire);
  return *vf == v ? 0 : (int)hipErrorUnknown;
}
inline int stream_create(Stream** s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
inline int stream_destroy(Stream* s) { return (int)hipStreamDestroy(s); }
inline int set_device(int d) { return (int)hipSetDevice(d); }
inline const char* last_error_string() { return hipGetErrorString(hipGetLastError()); }
typedef ihipEvent_t Event;
inline int event_create(Event** e) { return (int)hipEventCreate(e); }
inline int event_destroy(Event* e) { return (int)hipEventDestroy(e); }
inline int event_record(Event* e, Stream* s) { return (int)hipEventRecord(e, s); }
inline int event_elapsed_ms(float* ms, Event* a, Event* b) { return (int)hipEventElapsedTime(ms, a, b); }
inline int stream_wait_event(Stream* s, Event* e) { return (int)hipStreamWaitEvent(s, e, 0); }
inline int event_sync(Event* e) { return (int)hipEventSynchronize(e); }
// page-locked host memory the device can address (hipHostMalloc / hipHostRegister): the only kind of destination a copy engine may be pointed at directly
inline bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable memory: "invalid value" - clear the sticky error
  return a.type == hipMemoryTypeHost;
}

// ---- RCCL, bound at run time (dlopen: the library neither links against nor requires librccl unless ms_set_shard_rccl is used).
// Only the handful of entry points the sharded proof needs; types restated from <rccl/rccl.h> (ncclUniqueId = 128 opaque bytes,
// ncclUint8 = 1, ncclUint64 = 5, ncclSum = 0, ncclMin = 3, ncclSuccess = 0).
struct Rccl {
  typedef int (*GetUniqueIdFn)(void*);
  typedef int (*CommInitRankFn)(void**, int, const void* /* ncclUniqueId by value, see init_rank */, int);
  typedef int (*CommDestroyFn)(void*);
  typedef int (*GroupFn)();
  typedef int (*SendRecvFn)(void*, size_t, int, int, void*, Stream*);
  typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, Stream*);
  typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, Stream*);
  typedef const char* (*ErrStrFn)(int);
  struct UniqueId { char internal[128]; };
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr; int (*comm_init_rank)(void**, int, UniqueId, int) = nullptr; CommDestroyFn comm_destroy = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr; SendRecvFn send = nullptr, recv = nullptr; AllGatherFn all_gather = nullptr; AllReduceFn all_reduce = nullptr;
  ErrStrFn err_string = nullptr;
  static Rccl& get() { static Rccl r; return r; }
  // 0 on success.  MS_RCCL_LIB names the library (default: the RCCL already mapped into the process, e.g. PyTorch's, else librccl.so)
  int load();
};
}  // namespace msrt
#include <dlfcn.h>
namespace msrt {
inline int Rccl::load() {
  if (lib) return 0;
  const char* names[] = {getenv("MS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  if (const char* e = getenv("MS_RCCL_LIB")) { if (*e) h = dlopen(e, RTLD_NOW | RTLD_LOCAL); }
  // the RCCL already mapped into the process first (PyTorch's, built against the HIP runtime the process runs on): a second copy from another ROCm release would work
  // on a runtime it was not built for
  if (!h) for (const char* n : names) { if (n && *n) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; } }
  if (!h) for (const char* n : names) { if (n && *n) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; } }
  if (!h) return 1;
  auto sym = [&](const char* n) { return dlsym(h, n); };
  get_unique_id = (GetUniqueIdFn)sym("ncclGetUniqueId");
  comm_init_rank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
  comm_destroy = (CommDestroyFn)sym("ncclCommDestroy");
  group_start = (GroupFn)sym("ncclGroupStart"); group_end = (GroupFn)sym("ncclGroupEnd");
  send = (SendRecvFn)sym("ncclSend"); recv = (SendRecvFn)sym("ncclRecv");
  all_gather = (AllGatherFn)sym("ncclAllGather"); all_reduce = (AllReduceFn)sym("ncclAllReduce");
  err_string = (ErrStrFn)sym("ncclGetErrorString");
  if (!get_unique_id || !comm_init_rank || !comm_destroy || !group_start || !group_end || !send || !recv || !all_gather || !all_reduce) { dlclose(h); return 2; }
  lib = h;
  return 0;
}

}  // namespace msrt
// ---- SDMA copies through the HSA runtime, bound at run time (the runtime HIP itself sits on: already mapped into the process).
// hipMemcpyAsync(device -> page-locked host) leaves the choice between a copy engine and a shader blit kernel (`__amd_rocclr_copyBuffer`) to the runtime; with
// eight provers in flight part of the 64 MiB read-backs ran as blit kernels that take issue slots from the VALU-bound provers (r02/r03 I/O legs).
// hsa_amd_memory_async_copy_on_engine with force_copy_on_sdma queues the copy on a chosen SDMA engine; completion is an HSA signal the host waits on.
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <link.h>
#include <mutex>
#include <string>
namespace msrt {
struct Sdma {
  typedef hsa_signal_t Signal;
  void* lib = nullptr; int state = 0;   // 0 not tried, 1 ready, 2 unavailable
  std::mutex mu;
  decltype(&hsa_init) f_init = nullptr;
  decltype(&hsa_iterate_agents) f_iterate = nullptr;
  decltype(&hsa_agent_get_info) f_agent_info = nullptr;
  decltype(&hsa_signal_create) f_sig_create = nullptr;
  decltype(&hsa_signal_destroy) f_sig_destroy = nullptr;
  decltype(&hsa_signal_store_relaxed) f_sig_store = nullptr;
  decltype(&hsa_signal_wait_scacquire) f_sig_wait = nullptr;
  decltype(&hsa_amd_memory_async_copy_on_engine) f_copy = nullptr;
  decltype(&hsa_amd_memory_copy_engine_status) f_status = nullptr;
  decltype(&hsa_amd_memory_get_preferred_copy_engine) f_pref = nullptr;
  decltype(&hsa_amd_pointer_info) f_ptr_info = nullptr;
  hsa_agent_t cpu{0}; bool have_cpu = false;
  struct Gpu { hsa_agent_t agent; uint32_t bdf, domain; unsigned d2h_engine, h2d_engine; };
  std::vector<Gpu> gpus;
  std::vector<int> dev_gpu;   // HIP device ordinal -> index into gpus (-1: not matched)
  static Sdma& get() { static Sdma s; return s; }
  static hsa_status_t agent_cb(hsa_agent_t a, void* self) {
    Sdma* S = reinterpret_cast<Sdma*>(self);
    hsa_device_type_t ty;
    if (S->f_agent_info(a, HSA_AGENT_INFO_DEVICE, &ty) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (ty == HSA_DEVICE_TYPE_CPU) { if (!S->have_cpu) { S->cpu = a; S->have_cpu = true; } }
    else if (ty == HSA_DEVICE_TYPE_GPU) {
      Gpu g; g.agent = a; g.bdf = 0; g.domain = 0; g.d2h_engine = 0; g.h2d_engine = 0;
      S->f_agent_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &g.bdf);
      S->f_agent_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &g.domain);
      S->gpus.push_back(g);
    }
    return HSA_STATUS_SUCCESS;
  }
  // 0 on success.  Binds ONLY the HSA runtime the process already runs on: the loaded objects are enumerated (dl_iterate_phdr), the one named
  // libhsa-runtime64* is re-opened with RTLD_NOLOAD - whatever directory or soname it came from (a ROCm install, the PyTorch wheel's bundled copy) - and with
  // several copies mapped (a profiler preloading its own) the one next to the libamdhip64 this library is linked against wins, else none.  Nothing is ever
  // loaded: a second runtime would open KFD again and know none of HIP's allocations.  No runtime found = state 2: the caller keeps hipMemcpyAsync.
  std::string path;                                        // what was bound (ms_io_runtime_path)
  struct Scan { std::vector<std::string> hsa; std::string hip_dir; };
  static int scan_cb(struct dl_phdr_info* info, size_t, void* data) {
    Scan* sc = reinterpret_cast<Scan*>(data);
    const char* n = info->dlpi_name;
    if (!n || !*n) return 0;
    const char* base = strrchr(n, '/'); base = base ? base + 1 : n;
    if (!strncmp(base, "libhsa-runtime64", 16)) sc->hsa.push_back(n);
    return 0;
  }
  int load() {
    std::lock_guard<std::mutex> lk(mu);
    if (state) return state == 1 ? 0 : 1;
    state = 2;
    Scan sc;
    dl_iterate_phdr(&scan_cb, &sc);
    { Dl_info di; if (dladdr(reinterpret_cast<const void*>(&hipStreamSynchronize), &di) && di.dli_fname) { std::string f = di.dli_fname; const size_t k = f.rfind('/'); sc.hip_dir = k == std::string::npos ? std::string() : f.substr(0, k); } }
    std::string pick;
    if (sc.hsa.size() == 1) pick = sc.hsa[0];
    else for (const std::string& c : sc.hsa) { const size_t k = c.rfind('/'); if (k != std::string::npos && c.substr(0, k) == sc.hip_dir) { pick = c; break; } }
    if (pick.empty()) return 1;
    void* h = dlopen(pick.c_str(), RTLD_NOW | RTLD_NOLOAD);   // a handle on the mapped copy; NOLOAD: never a fresh one
    if (!h) return 1;
#define MS_HSA_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name)); if (!field) return 1
    MS_HSA_SYM(f_init, "hsa_init"); MS_HSA_SYM(f_iterate, "hsa_iterate_agents"); MS_HSA_SYM(f_agent_info, "hsa_agent_get_info");
    MS_HSA_SYM(f_sig_create, "hsa_signal_create"); MS_HSA_SYM(f_sig_destroy, "hsa_signal_destroy"); MS_HSA_SYM(f_sig_store, "hsa_signal_store_relaxed");
    MS_HSA_SYM(f_sig_wait, "hsa_signal_wait_scacquire"); MS_HSA_SYM(f_copy, "hsa_amd_memory_async_copy_on_engine"); MS_HSA_SYM(f_status, "hsa_amd_memory_copy_engine_status");
    MS_HSA_SYM(f_ptr_info, "hsa_amd_pointer_info");
#undef MS_HSA_SYM
    f_pref = reinterpret_cast<decltype(f_pref)>(dlsym(h, "hsa_amd_memory_get_preferred_copy_engine"));   // optional (HSA AMD extension 1.8)
    if (f_init() != HSA_STATUS_SUCCESS) return 1;            // reference-counted: HIP initialised this very copy (ms_create made HIP calls before any copy is queued)
    if (f_iterate(&agent_cb, this) != HSA_STATUS_SUCCESS || !have_cpu || gpus.empty()) return 1;
    lib = h; path = pick; state = 1;
    return 0;
  }
  const char* runtime_path() const { return path.c_str(); }
  // does THIS runtime know the allocation behind p (device memory of hipMalloc, page-locked host memory of hipHostMalloc)?  A pointer it has never seen means
  // the bound runtime is not the one HIP allocates through: no copy is queued on it
  bool knows(const void* p) {
    hsa_amd_pointer_info_t info; memset(&info, 0, sizeof info); info.size = sizeof info;
    if (f_ptr_info(const_cast<void*>(p), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    return info.type != HSA_EXT_POINTER_TYPE_UNKNOWN;
  }
  // HIP device -> HSA agent by PCI address; *gpu_index identifies it in the calls below
  int bind_device(int hip_device, int* gpu_index) {
    if (load()) return 1;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)dev_gpu.size() <= hip_device) dev_gpu.resize(hip_device + 1, -2);
    if (dev_gpu[hip_device] == -2) {
      int dom = 0, bus = 0, dv = 0;
      dev_gpu[hip_device] = -1;
      if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_device) == hipSuccess && hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) == hipSuccess &&
          hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, hip_device) == hipSuccess) {
        for (size_t i = 0; i < gpus.size(); i++)
          if (((gpus[i].bdf >> 8) & 0xFF) == (uint32_t)bus && ((gpus[i].bdf >> 3) & 0x1F) == (uint32_t)dv && gpus[i].domain == (uint32_t)dom) dev_gpu[hip_device] = (int)i;
      }
      if (dev_gpu[hip_device] < 0 && gpus.size() == 1) dev_gpu[hip_device] = 0;   // a single visible GPU needs no matching
      if (dev_gpu[hip_device] >= 0) {
        Gpu& g = gpus[dev_gpu[hip_device]];
        uint32_t avail = 0, pref = 0;
        if (const char* e = getenv("MS_SDMA_ENGINE")) g.d2h_engine = (unsigned)strtoul(e, nullptr, 0);      // an hsa_amd_sdma_engine_id_t bit
        else {
          f_status(cpu, g.agent, &avail);                                  // engines usable for device -> host (bits of hsa_amd_sdma_engine_id_t)
          if (f_pref) f_pref(cpu, g.agent, &pref);
          uint32_t m = (pref & avail) ? (pref & avail) : (avail ? avail : pref);
          g.d2h_engine = m ? (m & (~m + 1)) : 0;                           // lowest set bit
        }
        if (const char* e = getenv("MS_SDMA_ENGINE_H2D")) g.h2d_engine = (unsigned)strtoul(e, nullptr, 0);
        else {
          avail = pref = 0;
          f_status(g.agent, cpu, &avail);
          if (f_pref) f_pref(g.agent, cpu, &pref);
          uint32_t m = (pref & avail) ? (pref & avail) : (avail ? avail : pref);
          g.h2d_engine = m ? (m & (~m + 1)) : 0;
        }
        if (!g.d2h_engine) dev_gpu[hip_device] = -2;   // no engine reported free right now (all busy): not cached - the next caller asks again
      }
    }
    if (dev_gpu[hip_device] < 0) return 1;
    *gpu_index = dev_gpu[hip_device];
    return 0;
  }
  unsigned d2h_engine(int gpu_index) { std::lock_guard<std::mutex> lk(mu); return gpus[gpu_index].d2h_engine; }   // (under the mutex: bind_device of another context may rewrite the record - ADVICE r4)
  bool unavailable() const { return state == 2; }   // the HSA runtime could not be bound at all (as opposed to: no engine free at this moment)
  unsigned h2d_engine(int gpu_index) {               // callers read it ONCE per copy
    std::lock_guard<std::mutex> lk(mu);
    Gpu& g = gpus[gpu_index];
    if (!g.h2d_engine && !getenv("MS_SDMA_ENGINE_H2D")) {   // none was free when the device was bound: ask again
      uint32_t avail = 0, pref = 0;
      f_status(g.agent, cpu, &avail);
      if (f_pref) f_pref(g.agent, cpu, &pref);
      const uint32_t m = (pref & avail) ? (pref & avail) : (avail ? avail : pref);
      g.h2d_engine = m ? (m & (~m + 1)) : 0;
    }
    return g.h2d_engine;
  }
  // page-locked host memory -> device memory of `gpu_index`
  int copy_h2d(int gpu_index, void* dst_dev, const void* src_host, size_t n, Signal sig, unsigned engine) {
    if (!knows(dst_dev) || !knows(src_host)) return 1;
    f_sig_store(sig, 1);
    return (int)f_copy(dst_dev, gpus[gpu_index].agent, src_host, cpu, n, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)engine, true);
  }
  int signal_create(Signal* s) { return f_sig_create(1, 0, nullptr, s) == HSA_STATUS_SUCCESS ? 0 : 1; }
  void signal_destroy(Signal s) { if (s.handle) f_sig_destroy(s); }
  // device memory of `gpu_index` -> page-locked host memory, on SDMA engine `engine`; `sig` reads 0 when the bytes have landed
  int copy_d2h(int gpu_index, void* dst_host, const void* src_dev, size_t n, Signal sig, unsigned engine) {
    if (!knows(dst_host) || !knows(src_dev)) return 1;
    f_sig_store(sig, 1);
    return (int)f_copy(dst_host, cpu, src_dev, gpus[gpu_index].agent, n, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)engine, true);
  }
  // 0: complete (the signal reads 0); 1: not within `seconds` (seconds < 0: no limit); -1: the copy FAILED - the runtime reports a failed asynchronous copy by
  // setting the completion signal to a NEGATIVE value (hsa_ext_amd.h), which must not pass for completion
  int wait(Signal sig, double seconds) {
    const uint64_t slice = 2000000000ull;   // the timeout is in ticks of the system timestamp counter (>= 1e8 Hz here): wake up now and then, bound the total by wall time
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
      const hsa_signal_value_t v = f_sig_wait(sig, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_BLOCKED);
      if (v == 0) return 0;
      if (v < 0) return -1;
      if (seconds < 0) continue;
      timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > seconds) return 1;
    }
  }
};

// optional K::MIN_WAVES = minimum waves per SIMD the register allocator must leave room for
template <class K, class = void> struct MinWaves { static constexpr int v = 1; };
template <class K> struct MinWaves<K, decltype((void)K::MIN_WAVES)> { static constexpr int v = K::MIN_WAVES; };
template <class K>
__global__ void __launch_bounds__(K::THREADS, MinWaves<K>::v) ms_kmain(const typename K::Params p) {
  extern __shared__ __align__(16) unsigned char ms_lds[];
  const int nph = K::nphases(p);
  for (int ph = 0; ph < nph; ph++) {
    K::phase(ph, p, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, K::THREADS, ms_lds);
    if (ph + 1 < nph) __syncthreads();
  }
}
template <class K>
inline int launch(Stream* s, unsigned gx, unsigned gy, int threads, size_t lds_bytes, const typename K::Params& p) {
  if (threads != K::THREADS) return (int)hipErrorInvalidValue;
  if (lds_bytes > 65536) {  // opt in to more than 64 KiB of dynamic LDS (per device: set on every such launch, microseconds)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ms_kmain<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ms_kmain<K>), dim3(gx, gy, 1), dim3(threads, 1, 1), lds_bytes, s, p);
  return (int)hipGetLastError();
}
// cooperative style: K::run holds the workgroup barriers itself (per-thread state stays in registers across them)
template <class K>
__global__ void __launch_bounds__(K::THREADS, MinWaves<K>::v) ms_kmain_coop(const typename K::Params p) {
  extern __shared__ __align__(16) unsigned char ms_lds[];
  K::run(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)threadIdx.x, ms_lds);
}
template <class K>
inline int launch_coop(Stream* s, unsigned gx, unsigned gy, int threads, size_t lds_bytes, const typename K::Params& p) {
  if (threads != K::THREADS) return (int)hipErrorInvalidValue;
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ms_kmain_coop<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ms_kmain_coop<K>), dim3(gx, gy, 1), dim3(threads, 1, 1), lds_bytes, s, p);
  return (int)hipGetLastError();
}
// Workgroup barrier of the cooperative kernels, ordering LDS traffic only: __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), which would end every prefetch (global loads issued for the NEXT tile) at the first barrier behind it.
// Global data written before the barrier is NOT made visible by it (these kernels never hand global data between waves).

MS_DEV void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and the barrier that DOES hand global data from wave to wave of a workgroup (fused multi-step kernels: one step's stores are the next step's loads)
MS_DEV void wg_barrier_global() { __syncthreads(); }
// device-scope release / acquire fence: global stores of this thread before it are visible to every workgroup of the device that synchronises with it through an
// atomic behind it (the "last workgroup finishes the job" hand-over of the fused FRI round, fri_tail.hpp): L2 write-back / invalidate across the XCDs
MS_DEV void fence_device() { __threadfence(); }
MS_DEV unsigned long long wave_shfl_xor(unsigned long long v, int mask) { return __shfl_xor(v, mask, 64); }
// a value every lane of the wave holds alike, moved to a scalar register (so that branches on it are scalar branches)
MS_DEV int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
MS_DEV void atomic_min_u64(unsigned long long* a, unsigned long long v) { atomicMin(a, v); }
MS_DEV void atomic_max_u64(unsigned long long* a, unsigned long long v) { atomicMax(a, v); }
MS_DEV unsigned atomic_add_u32(unsigned* a, unsigned v) { return atomicAdd(a, v); }
// true if the predicate holds on any live lane of the wave
MS_DEV bool wave_any(bool pred) { return __any(pred) != 0; }
// one slot of a device list per lane that wants one (call from all live lanes): ONE atomic per wave, not per lane
MS_DEV unsigned wave_alloc_slot(unsigned* counter, bool want) {
  const unsigned long long mask = __ballot(want);
  if (!want) return 0;
  const unsigned lane = __lane_id();
  const int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
  base = (unsigned)__shfl((int)base, leader);
  return base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
}
// Lane-pair exchanges of msmerkle::Sha256Pair: lane i and lane 7 - i (= i ^ 7) of every group of eight lanes are partners (DPP row_half_mirror); the lanes of banks
// 0 and 2 of a 16-lane row (i & 4 == 0) hold the e-side of a round, those of banks 1 and 3 the a-side.
MS_DEV unsigned pair_swap(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); }
// e-lanes: partner's v + t; a-lanes: partner's t + s - two bank-masked DPP adds into one register.  The wait state between them (with the first add) makes the two a VALU
// write of t needs before a DPP read of it; v is three rounds old.
MS_DEV unsigned pair_exchange_add(bool, unsigned v, unsigned t, unsigned s) {
  unsigned n;
  asm("v_add_u32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\ts_nop 0\n\tv_add_u32_dpp %0, %2, %3 row_half_mirror row_mask:0xf bank_mask:0xa" : "=&v"(n) : "v"(v), "v"(t), "v"(s));
  return n;
}
MS_DEV unsigned rotr_var(unsigned x, unsigned r) { return __builtin_amdgcn_alignbit(x, x, r); }
// "this stage's results are in page-locked host memory": a sequence number the kernel that ends a stage stores BEHIND its results, for a host that polls the word
// instead of synchronising with the stream (MS_FLAG_LATENCY: 6.4 instead of 11.0 us from launch to host-visible result, profiles/r05_latency_probe.txt).  Called by
// the ONE thread that stored the results itself: system-scope fence (its stores have left for the host), then a system-scope release store of the word.
struct HostFlag { unsigned long long* dst; unsigned long long val; };
MS_DEV void raise_host_flag(const HostFlag& f) {
  if (!f.dst) return;
  __threadfence_system();
  __hip_atomic_store(f.dst, f.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and the form for results several threads of ONE workgroup stored: called by every thread of the workgroup (f is uniform)
MS_DEV void raise_host_flag_wg(const HostFlag& f, int tid) {
  if (!f.dst) return;
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(f.dst, f.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace msrt
MS_HD uint64_t ms_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
#endif

// ---- bit-level helpers shared by the three builds ---------------------------------------------
// Arbitrary 3-input bit function, truth table TT indexed by (a << 2 | b << 1 | c).  gfx950: ONE v_bitop3_b32, which issues at
// the full VALU rate (~2.7 cycles per wave-instruction) where compares, v_cndmask and carry ops cost ~4.5
// (profiles/r04_valu_issue_rate.txt) and compares additionally route their result through an SGPR pair (2 wait states before use).
template <int TT> MS_HD uint32_t ms_bitop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
  return ((TT & 1) ? (~a & ~b & ~c) : 0u) | ((TT & 2) ? (~a & ~b & c) : 0u) | ((TT & 4) ? (~a & b & ~c) : 0u) | ((TT & 8) ? (~a & b & c) : 0u) |
         ((TT & 16) ? (a & ~b & ~c) : 0u) | ((TT & 32) ? (a & ~b & c) : 0u) | ((TT & 64) ? (a & b & ~c) : 0u) | ((TT & 128) ? (a & b & c) : 0u);
#endif
}
// all-ones if bit 31 is set, else zero (v_ashrrev_i32)
MS_HD uint32_t ms_sar31(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }
// keeps a 64-bit value materialised as ONE register pair (stops the compiler from splitting "x + (lo | hi << 32)" into two adds)
MS_HD uint64_t ms_pin64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(v));
#endif
  return v;
}

