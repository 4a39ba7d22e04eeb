// ministark.cpp — host side of the C ABI (include/ministark.h): context, NTT
// plans, Merkle builder and the Stark::prove / Fri::prove stage functions that
// sequence the kernels of ntt.hpp / merkle.hpp / poly.hpp on one HIP stream.
// Compiled with hipcc for gfx950 (libministark.so).  There is no CPU fallback.
#include "../../include/ministark.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "field.hpp"
#include "merkle.hpp"
#include "ntt.hpp"
#include "poly.hpp"

namespace {

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) msrt::free_dev(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (msrt::malloc_dev(&p, want)) { p = nullptr; return 1; }
    cap = want;
    return 0;
  }
  void release() { if (p) msrt::free_dev(p); p = nullptr; cap = 0; }
  template <class U> U* as() const { return reinterpret_cast<U*>(p); }
};

// ---- src/util.rs:4-44 ------------------------------------------------------
inline bool is_pow2(u64 n) { return (n & (n - 1)) == 0; }
inline int ctz64(u64 x) { return x ? __builtin_ctzll(x) : 64; }
inline long log_two_k(u64 number, u64 base) {
  int log_n = ctz64(base);
  if (!is_pow2(number)) return -1;
  int p2 = ctz64(number);
  if (p2 % log_n != 0) return -2;
  return p2 / log_n;
}
inline u64 ceil_log2_k(u64 number, u64 base) {
  if (number == 1) return 1;
  u64 lb = ctz64(base), ln = ctz64(number);
  if (is_pow2(number) && ln % lb == 0) return ln;
  u64 np2 = 64 - __builtin_clzll(number);
  return ((np2 + lb - 1) / lb) * lb;
}

struct CtxBase {
  std::string err;
  virtual ~CtxBase() {}
  virtual int ext_degree() const = 0;
  virtual void bind_device() const = 0;  // HIP's current device is per host thread: every entry point binds the context's device
  virtual int set_stream(void* s) = 0;
  virtual int synchronize() = 0;
  virtual int set_shard(int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user) = 0;
  virtual int set_shard_rccl(int rank, int world, const u8* unique_id, size_t cap) = 0;
  virtual int shard_stats(u64* out) = 0;
  virtual int shard_proof_on_root(int on) = 0;
  virtual int shard_round_is_distributed(int r) = 0;
  virtual int shard_proof_is_elsewhere() const = 0;
  virtual int shard_slice_layout(size_t* off, size_t* stride) = 0;
  virtual int rccl_selftest() = 0;
  virtual int trace_commit(const u64* trace, bool on_device, size_t N, size_t w, size_t lpn, u8* root) = 0;
  virtual int interpolate() = 0;
  virtual int polys_lincomb(const u64* s, const int* idx, int k) = 0;
  virtual int polys_append(const u64* coeffs, size_t n) = 0;
  virtual int polys_count() const = 0;
  virtual int poly_read(int i, u64* out) = 0;
  virtual int lde_commit(size_t blowup, u64 shift, size_t lpn, u8* root) = 0;
  virtual int lde_read(u64* out) = 0;
  virtual int mix(u64 r) = 0;
  virtual int mix_cubic(u64 r, const int* spec, const u64* sc, int ncons) = 0;
  virtual size_t validity_len_() const = 0;
  virtual int validity_read(u64* out) = 0;
  virtual int eval_ext(const u64* z, int q, u64* out) = 0;
  virtual int fri_begin(size_t blowup, size_t rounds, u8* root0) = 0;
  virtual int fri_deep(const u64* z, u64* B) = 0;
  virtual int fri_fold_commit(const u64* alpha, u8* root) = 0;
  virtual int fri_round_info(int r, u64* ncoef, u64* D) = 0;
  virtual int fri_round_poly_read(int r, u64* out) = 0;
  virtual int fri_round_codeword_read(int r, u64* out) = 0;
  virtual int fri_query(const u64* betas, int nq, u8* ext_out, size_t ext_cap, size_t* ext_len) = 0;
  virtual size_t fri_proof_size() const = 0;
  virtual int fri_proof_read(u8* out) = 0;
  virtual int fri_proof_read_async(u8* out) = 0;
  virtual int fri_proof_wait() = 0;
  virtual int io_engine() const = 0;
  virtual int merkle_commit(const u64* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, u8* nodes_out, size_t cap, size_t* nn, u8* root) = 0;
  virtual int merkle_prove(const u64* leafs, size_t leaf_num, int ext, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) = 0;
  virtual int ntt(u64* data, size_t n, size_t batch, int inverse) = 0;
  virtual int coset_lde(const u64* coeffs, size_t ncoef, size_t batch, u64 shift, u64* out, size_t L) = 0;
  virtual int bench_lde(size_t blowup, u64 shift) = 0;
  virtual int arith_selftest(int op, const u64* a, const u64* b, u64* out, size_t n) = 0;
  virtual int profile_begin() = 0;
  virtual int profile_end(char* out, size_t cap) = 0;
};

#define CK(...) do { int _e = (__VA_ARGS__); if (_e) return this->fail_rt(_e, #__VA_ARGS__); } while (0)
#define RQ(...) do { int _e = (__VA_ARGS__); if (_e) return _e; } while (0)

template <class F> struct Ctx : CtxBase {
  typedef typename F::T T;
  static constexpr int E = F::EXT;
  typedef Ext<F, E> XE;

  int device = 0, zae = 1, trace_mont = 0;
  int lde_linear = 1;  // MS_LDE_LINEAR=0 disables the linear-provenance shortcut of lde_compute (A/B measurements)
  int lde_multi = -1;  // MS_LDE_MULTI: the linear LDE columns in shared sweeps (LincombMultiKernel).  -1 (default): for AIRs of >= 16 polynomials (the wide AIR: 112
                       // launches become 16 and the LDE commit 108.7 -> 105.6 ms, r03); 0 / 1 force it.  Narrow AIRs keep the one-by-one kernel: measured (r02) 45 us less
                       // per Fibonacci proof alone, but 248 -> 240 proofs/s with 8 proofs in flight (five interleaved runs each)
  int tree_top_parents = msmerkle::THREADS;  // MS_TREE_TOP: levels of at most this many parents are walked by one workgroup in one launch (measured: 256 beats 1024 by 3 % in latency)
  int leaf_lazy_min = 16;  // MS_LEAF_LAZY_MIN: leaf groups of at least this many base limbs use the wave-synchronous two-block leaf kernel
  size_t fold_small_max = 131072;  // MS_FOLD_SMALL_MAX (16384 / 131072 / 2^20: within noise of each other with one proof and with eight in flight; 0 is 1.5 % slower with one)
  size_t eval_small_max = (size_t)1 << 19;   // MS_EVAL_SMALL_MAX: polynomials of at most this many coefficients are evaluated 4 coefficients per thread (latency), longer ones 16
  // MS_TREE_SUBTREE_PARENTS: binary-tree levels of at most this many parents run as subtree launches (msmerkle::InnerSubtreeKernel); 0: one launch per level + the fused top.
  // Same-box A/Bs (profiles/r04_small_round_kernels_ab.log), final build, 4 passes: 8 proofs in flight 4096 / 16384 / 65536 -> 258.9 / 260.5 / 254.2 proofs/s, one proof in
  // flight 127.0 / 129.6 / 130.3; against 0 (one launch per level): 4096 -> +0.8 % in flight, +1.2 % alone
  size_t subtree_parents = 16384;
  int fri_pointwise = 1;  // MS_FRI_POINTWISE=0: codewords of FRI rounds >= 1 by NTT of the round polynomial instead of the evaluation-domain fold
  msrt::Stream* own_stream = nullptr;
  msrt::Stream* stream = nullptr;
  void* pinned = nullptr; size_t pinned_cap = 0;

  // ---- one proof sharded over `sh_world` ranks (ms_set_shard; include/ministark.h)
  int sh_rank = 0, sh_world = 1;
  // sh_on: the sharded code paths are active.  Normally that is world > 1; with MS_SHARD_WORLD1=1 (tests) a ONE-rank "world" runs them too - every exchange degenerates to a
  // transfer to itself, but every kernel, buffer offset, stream ordering and RCCL call of the sharded prover executes, through whole proofs, on one GPU
  bool sh_on = false; int allow_w1 = 0;
  u8* xs = nullptr; u8* xr = nullptr; size_t xcap = 0;   // caller-owned exchange buffers (device)
  ms_exchange_fn xfn = nullptr; void* xuser = nullptr;
  size_t shard_min_leaves = 32768;                        // MS_SHARD_MIN_LEAVES: smaller commitments stay replicated
  void* rccl_comm = nullptr; DevBuf rccl_send, rccl_recv;   // ms_set_shard_rccl: the library owns the communicator and the exchange buffers
  size_t rccl_max_piece = (size_t)1 << 30;                  // MS_RCCL_MAX_PIECE: most bytes of one ncclSend / ncclRecv / ncclAllGather (see exchange)
  u64 xstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // calls per op [0..3], bytes sent per op [4..7] (ms_shard_stats)
  int exchange(int op, size_t bytes) {
    { const int slot = (op == MS_XCHG_GATHER) ? 1 : (op & 3);   // (the gather to rank 0 is counted with the all-gathers)
      xstat[slot]++; xstat[4 + slot] += (op == MS_XCHG_ALL_TO_ALL ? bytes * (size_t)(sh_world - 1) : (op == MS_XCHG_GATHER && sh_rank == 0 ? 0 : bytes)); }
    if (rccl_comm) {
      // RCCL on the context's stream: stream-ordered with the kernels on both sides, no host synchronisation, no host callback
      msrt::Rccl& R = msrt::Rccl::get();
      const int W = sh_world;
      int e = 0;
      // One ncclSend / ncclRecv moves at most rccl_max_piece bytes (default 1 GiB; MS_RCCL_MAX_PIECE): found r05 by the full-size one-rank-world tests - RCCL 2.26.6
      // delivers WRONG BYTES for a single send / recv of >= 2 GiB of ncclUint8 (2^23 rows unsliced on one rank: 2 GiB to itself; 1 GiB and 4 x 512 MiB are right).
      // On W GPUs a peer's chunk of a 2^24-row commitment is 4 GiB / W^2, so this only matters for W <= 2 - but a silent wrong root is not an acceptable failure mode.
      auto sendrecv = [&](const u8* sp, int to, u8* rp, int from, size_t n, msrt::Stream* st) -> int {
        int er = 0;
        for (size_t o = 0; o < n && !er; o += rccl_max_piece) {
          const size_t len = n - o < rccl_max_piece ? n - o : rccl_max_piece;
          if (sp) er = R.send(const_cast<u8*>(sp) + o, len, 1 /* ncclUint8 */, to, rccl_comm, st);
          if (rp && !er) er = R.recv(rp + o, len, 1, from, rccl_comm, st);
        }
        return er;
      };
      if (op == MS_XCHG_ALL_TO_ALL) {
        e = R.group_start();
        for (int r = 0; r < W && !e; r++) e = sendrecv(xs + (size_t)r * bytes, r, xr + (size_t)r * bytes, r, bytes, stream);
        const int e2 = R.group_end();
        if (!e) e = e2;
      } else if (op == MS_XCHG_GATHER) {   // to rank 0 only: chunk r of its receive buffer from rank r
        e = R.group_start();
        if (sh_rank != 0) { if (!e) e = sendrecv(xs, 0, nullptr, 0, bytes, stream); }
        else for (int r = 1; r < W && !e; r++) e = sendrecv(nullptr, 0, xr + (size_t)r * bytes, r, bytes, stream);
        const int e2 = R.group_end();
        if (!e) e = e2;
        if (!e && sh_rank == 0 && msrt::d2d(xr, xs, bytes, stream)) e = -1;
      } else if (op == MS_XCHG_ALL_GATHER) {
        if (bytes <= rccl_max_piece) e = R.all_gather(xs, xr, bytes, 1, rccl_comm, stream);
        else {   // (a gathered round polynomial of a <= 2-rank world: the same pieces as grouped send / recv)
          e = R.group_start();
          for (int r = 0; r < W && !e; r++) e = sendrecv(xs, r, xr + (size_t)r * bytes, r, bytes, stream);
          const int e2 = R.group_end();
          if (!e) e = e2;
        }
      }
      else if (op == MS_XCHG_ALL_REDUCE_MIN_U64) e = R.all_reduce(xs, xs, bytes / 8, 5 /* ncclUint64 */, 3 /* ncclMin */, rccl_comm, stream);
      else e = R.all_reduce(xs, xs, bytes, 1, 0 /* ncclSum */, rccl_comm, stream);
      if (e) { err = std::string("RCCL error ") + std::to_string(e) + (R.err_string ? std::string(": ") + R.err_string(e) : std::string()); return MS_ERR_HIP; }
      return 0;
    }
    CK(msrt::sync(stream));
    if (xfn(xuser, op, bytes)) return fail(MS_ERR_HIP, "exchange callback failed");
    return 0;
  }
  // one slice of a sliced digest all-to-all: peer r's piece at offset `off + r * stride`, `bytes` long, in both exchange buffers
  int shard_slices = 4; bool shard_slices_set = false; size_t shard_slice_min = 1024;   // (RCCL path: 1 unless MS_SHARD_SLICES is set - ADVICE r3: the two-stream overlap has never run on more than one GPU)
  size_t xl_off = 0, xl_stride = 0;
  msrt::Stream* comm_stream = nullptr; std::vector<msrt::Event*> ev_hash, ev_xchg;
  int shard_slice_layout(size_t* off, size_t* stride) override { if (!off || !stride) return fail(MS_ERR_ARG, "shard_slice_layout"); *off = xl_off; *stride = xl_stride; return MS_OK; }
  int exchange_slice(size_t off, size_t stride, size_t bytes, int sl, int S) {
    if (sl == 0) xstat[0]++;                        // one all-to-all per commitment, whatever the number of slices
    xstat[4] += bytes * (size_t)(sh_world - 1);
    if (rccl_comm) {
      msrt::Rccl& R = msrt::Rccl::get();
      if (!comm_stream) CK(msrt::stream_create(&comm_stream));
      while ((int)ev_hash.size() < S) { msrt::Event* a; msrt::Event* b; CK(msrt::event_create(&a)); CK(msrt::event_create(&b)); ev_hash.push_back(a); ev_xchg.push_back(b); }
      CK(msrt::event_record(ev_hash[sl], stream));                 // slice hashed (incl. its deferred pad-only blocks)
      CK(msrt::stream_wait_event(comm_stream, ev_hash[sl]));
      int e = R.group_start();
      for (int r = 0; r < sh_world && !e; r++)
        for (size_t o = 0; o < bytes && !e; o += rccl_max_piece) {   // (pieces: see exchange)
          const size_t len = bytes - o < rccl_max_piece ? bytes - o : rccl_max_piece;
          e = R.send(xs + off + (size_t)r * stride + o, len, 1 /* ncclUint8 */, r, rccl_comm, comm_stream);
          if (!e) e = R.recv(xr + off + (size_t)r * stride + o, len, 1, r, rccl_comm, comm_stream);
        }
      const int e2 = R.group_end();
      if (!e) e = e2;
      if (e) { err = std::string("RCCL error ") + std::to_string(e) + (R.err_string ? std::string(": ") + R.err_string(e) : std::string()); return MS_ERR_HIP; }
      CK(msrt::event_record(ev_xchg[sl], comm_stream));
      if (sl == S - 1) CK(msrt::stream_wait_event(stream, ev_xchg[sl]));   // the communication stream runs in order: the last slice's event covers all of them
      return 0;
    }
    CK(msrt::sync(stream));
    xl_off = off; xl_stride = stride;
    if (xfn(xuser, MS_XCHG_ALL_TO_ALL_SLICE, bytes)) return fail(MS_ERR_HIP, "exchange callback failed");
    return 0;
  }
  void drop_rccl() {
    if (rccl_comm) { msrt::sync(stream); if (comm_stream) msrt::sync(comm_stream); msrt::Rccl::get().comm_destroy(rccl_comm); rccl_comm = nullptr; }   // (work may be queued on either stream)
    rccl_send.release(); rccl_recv.release();
  }
  void unshard() { sh_on = false; sh_rank = 0; sh_world = 1; xs = xr = nullptr; xcap = 0; xfn = nullptr; xuser = nullptr; have_lde = false; nrounds_done = 0; blob_size = 0; }
  int set_shard_rccl(int rank, int world, const u8* unique_id, size_t cap) override {
    if (world < 1 || !is_pow2((u64)world) || rank < 0 || rank >= world) return fail(MS_ERR_ARG, "set_shard_rccl: world must be a power of two and 0 <= rank < world");
    // the old communicator and buffers go first; until the new ones are complete the context is UNSHARDED, so that a failure below
    // (no unique id, librccl missing, out of memory, ncclCommInitRank) cannot leave sh_world > 1 over freed buffers / a null callback
    drop_rccl();
    unshard();
    if (world == 1 && !(allow_w1 && unique_id && cap >= 4096)) return MS_OK;
    if (!unique_id || cap < 4096) return fail(MS_ERR_ARG, "set_shard_rccl: unique id / buffer capacity missing");
    msrt::Rccl& R = msrt::Rccl::get();
    if (R.load()) return fail(MS_ERR_HIP, "RCCL is not available (librccl.so could not be loaded; MS_RCCL_LIB names it)");
    if (rccl_send.ensure(cap) || rccl_recv.ensure(cap)) { drop_rccl(); return fail(MS_ERR_NOMEM, "exchange buffers"); }
    msrt::Rccl::UniqueId id; memcpy(id.internal, unique_id, sizeof id.internal);
    void* comm = nullptr;
    const int e = R.comm_init_rank(&comm, world, id, rank);
    if (e || !comm) { drop_rccl(); err = std::string("ncclCommInitRank failed: ") + (R.err_string ? R.err_string(e) : "?"); return MS_ERR_HIP; }
    rccl_comm = comm;
    sh_rank = rank; sh_world = world; xs = rccl_send.as<u8>(); xr = rccl_recv.as<u8>(); xcap = cap; sh_on = true;
    return MS_OK;
  }
  // The four collectives on a one-rank communicator (send/recv to self, all-gather, both all-reduces) with known payloads:
  // checks the run-time binding of librccl.so (symbols, calling convention of the by-value ncclUniqueId, datatype / op enums)
  // and the stream ordering on a box with a single GPU.
  int rccl_selftest() override {
    msrt::Rccl& R = msrt::Rccl::get();
    if (R.load()) return fail(MS_ERR_HIP, "RCCL is not available (librccl.so could not be loaded; MS_RCCL_LIB names it)");
    msrt::Rccl::UniqueId id;
    if (R.get_unique_id(&id)) return fail(MS_ERR_HIP, "ncclGetUniqueId failed");
    void* comm = nullptr;
    int e = R.comm_init_rank(&comm, 1, id, 0);
    if (e || !comm) return fail(MS_ERR_HIP, "ncclCommInitRank(1 rank) failed");
    DevBuf a, b;
    int rc = MS_OK;
    if (a.ensure(4096) || b.ensure(4096)) rc = fail(MS_ERR_NOMEM, "selftest buffers");
    u64 h[64], g[64];
    for (int i = 0; i < 64; i++) h[i] = 0x0123456789ABCDEFull * (u64)(i + 1);
    if (!rc && (msrt::h2d(a.p, h, sizeof h, stream) || msrt::memset_dev(b.p, 0, 4096, stream))) rc = fail(MS_ERR_HIP, "selftest upload");
    if (!rc) {
      e = R.group_start();
      if (!e) e = R.send(a.p, 256, 1, 0, comm, stream);
      if (!e) e = R.recv(b.p, 256, 1, 0, comm, stream);
      const int e2 = R.group_end(); if (!e) e = e2;
      if (!e) e = R.all_gather(a.as<u8>() + 256, b.as<u8>() + 256, 128, 1, comm, stream);
      if (!e) e = R.all_reduce(a.p, a.p, 8, 5, 3, comm, stream);        // MIN over one rank: unchanged
      if (!e) e = R.all_reduce(a.p, a.p, 64, 1, 0, comm, stream);       // SUM over one rank: unchanged
      if (e) rc = fail(MS_ERR_HIP, "RCCL collective failed in the self test");
    }
    if (!rc && (msrt::d2h(g, b.p, sizeof g, stream) || msrt::sync(stream))) rc = fail(MS_ERR_HIP, "selftest download");
    if (!rc) for (int i = 0; i < 48; i++) if (g[i] != h[i]) { rc = fail(MS_ERR_HIP, "RCCL self test: payload mismatch"); break; }
    // the choreography of a sliced digest exchange (exchange_slice): payload produced on the prover's stream, grouped send / recv on a SECOND stream behind an
    // event, the prover's stream waiting for the exchange's event before it reads the result
    if (!rc) {
      msrt::Stream* cs = nullptr; msrt::Event* e1 = nullptr; msrt::Event* e2 = nullptr;
      u64 h2[32], g2[32];
      for (int i = 0; i < 32; i++) h2[i] = 0xA5A5A5A5DEADBEEFull + (u64)i * 0x1000193ull;
      int er = msrt::stream_create(&cs) || msrt::event_create(&e1) || msrt::event_create(&e2);
      if (!er) er = msrt::h2d(a.as<u8>() + 1024, h2, sizeof h2, stream) || msrt::memset_dev(b.as<u8>() + 1024, 0, sizeof h2, stream);
      if (!er) er = msrt::event_record(e1, stream) || msrt::stream_wait_event(cs, e1);
      if (!er) {
        e = R.group_start();
        if (!e) e = R.send(a.as<u8>() + 1024, 128, 1, 0, comm, cs);
        if (!e) e = R.recv(b.as<u8>() + 1024, 128, 1, 0, comm, cs);
        if (!e) e = R.send(a.as<u8>() + 1024 + 128, 128, 1, 0, comm, cs);     // a second, strided piece in the same group, as a slice of several peers' chunks would be
        if (!e) e = R.recv(b.as<u8>() + 1024 + 128, 128, 1, 0, comm, cs);
        const int e3 = R.group_end(); if (!e) e = e3;
        er = e;
      }
      if (!er) er = msrt::event_record(e2, cs) || msrt::stream_wait_event(stream, e2);
      if (!er) er = msrt::d2h(g2, b.as<u8>() + 1024, sizeof g2, stream) || msrt::sync(stream);
      if (er) rc = fail(MS_ERR_HIP, "RCCL self test: exchange on the communication stream failed");
      else for (int i = 0; i < 32; i++) if (g2[i] != h2[i]) { rc = fail(MS_ERR_HIP, "RCCL self test: payload mismatch on the communication stream"); break; }
      if (cs) { msrt::sync(cs); msrt::stream_destroy(cs); }
      if (e1) msrt::event_destroy(e1);
      if (e2) msrt::event_destroy(e2);
    }
    if (!rc && (msrt::d2h(g, a.p, sizeof g, stream) || msrt::sync(stream))) rc = fail(MS_ERR_HIP, "selftest download");
    if (!rc) for (int i = 0; i < 64; i++) if (g[i] != h[i]) { rc = fail(MS_ERR_HIP, "RCCL self test: all-reduce changed a one-rank payload"); break; }
    msrt::sync(stream);
    R.comm_destroy(comm);
    a.release(); b.release();
    return rc;
  }
  int shard_proof_on_root(int on) override { proof_root_only = on ? 1 : 0; return MS_OK; }
  int shard_proof_is_elsewhere() const override { return (proof_root_only && sh_world > 1 && sh_rank != 0 && nrounds_done == fri_rounds && fri_rounds) ? 1 : 0; }
  int shard_round_is_distributed(int r) override { return (r >= 0 && (size_t)r < nrounds_done && rounds[r]->dist) ? 1 : 0; }
  int shard_stats(u64* out) override { if (!out) return fail(MS_ERR_ARG, "shard_stats"); memcpy(out, xstat, sizeof xstat); return MS_OK; }
  bool shardable(size_t leaf_groups) const { return sh_on && leaf_groups >= shard_min_leaves && leaf_groups >= (size_t)sh_world * (size_t)sh_world; }
  int set_shard(int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user) override {
    if (world < 1 || !is_pow2((u64)world) || rank < 0 || rank >= world) return fail(MS_ERR_ARG, "set_shard: world must be a power of two and 0 <= rank < world");
    if (world > 1 && (!d_send || !d_recv || !fn || cap < 4096)) return fail(MS_ERR_ARG, "set_shard: exchange buffers / callback missing");
    drop_rccl();
    sh_rank = rank; sh_world = world; xs = reinterpret_cast<u8*>(d_send); xr = reinterpret_cast<u8*>(d_recv); xcap = cap; xfn = fn; xuser = user;
    sh_on = world > 1 || (allow_w1 && d_send && d_recv && fn && cap >= 4096);
    have_lde = false; nrounds_done = 0; blob_size = 0;
    return MS_OK;
  }

  int fail_rt(int e, const char* what) { err = std::string("runtime error ") + std::to_string(e) + " in " + what + ": " + msrt::last_error_string(); return MS_ERR_HIP; }
  int fail(int code, const char* msg) { err = msg; return code; }

  // ---- optional per-kernel timing with HIP events on the launching stream (bench.py roofline leg)
  enum { K_NTT_PASS, K_SCALE_POW, K_LEAF_HASH, K_INNER_HASH, K_TRANSPOSE, K_IO, K_LINCOMB, K_MIX, K_EVAL, K_EVAL_REDUCE, K_FOLD,
         K_SUFFIX_HORNER, K_DEGREE, K_FIND_FIRST, K_PATH, K_QUERY_POINTS, K_COUNT };
  struct ProfRec { int kid, sub; msrt::Event* a; msrt::Event* b; double bytes; bool part; };
  // sharded proofs: launches inside a PartScope work on this rank's PART of the proof (1 / world of it); everything else is replicated on every rank.
  // ms_profile_end reports both sums: the replicated one bounds the strong scaling (bench.py: sharded.replicated_ms_estimate)
  int part_depth = 0;
  struct PartScope { Ctx* c; explicit PartScope(Ctx* c_) : c(c_) { c->part_depth++; } ~PartScope() { c->part_depth--; } };
  bool prof_on = false;
  std::vector<ProfRec> prof_recs;
  double next_bytes = 0;  // algorithmic bytes attributed to the next launch
  int next_sub = 0;       // NTT pass template instance of the next launch: index into ntt_names (the kernel names rocprofv3 reports)
  std::vector<std::string> ntt_names{std::string("?")};
  void name_next(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    if (!prof_on) return;
    char buf[200]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    for (size_t i = 0; i < ntt_names.size(); i++) if (ntt_names[i] == buf) { next_sub = (int)i; return; }
    ntt_names.push_back(buf); next_sub = (int)ntt_names.size() - 1;
  }
  static const char* fname() { return F::ID == 0 ? "GL" : "BB"; }
  template <class A> static const char* aname() { return std::is_same<A, GLM>::value ? "GLM" : (std::is_same<A, GLT>::value ? "GLT" : (std::is_same<A, GL>::value ? "GL" : "BB")); }
  template <class K> int run_coop(int kid, unsigned gx, int threads, size_t lds, const typename K::Params& p, unsigned gy = 1) {
    if (gx == 0 || gy == 0) return 0;
    if (!prof_on) return msrt::launch_coop<K>(stream, gx, gy, threads, lds, p);
    ProfRec r; r.kid = kid; r.sub = next_sub; r.bytes = next_bytes; r.part = part_depth > 0; next_bytes = 0; next_sub = 0;
    if (msrt::event_create(&r.a) || msrt::event_create(&r.b)) return 1;
    msrt::event_record(r.a, stream);
    int e = msrt::launch_coop<K>(stream, gx, gy, threads, lds, p);
    msrt::event_record(r.b, stream);
    prof_recs.push_back(r);
    return e;
  }
  template <class K> int run(int kid, unsigned gx, unsigned gy, int threads, size_t lds, const typename K::Params& p) {
    if (gx == 0 || gy == 0) return 0;
    if (!prof_on) return msrt::launch<K>(stream, gx, gy, threads, lds, p);
    ProfRec r; r.kid = kid; r.sub = next_sub; r.bytes = next_bytes; r.part = part_depth > 0; next_bytes = 0; next_sub = 0;
    if (msrt::event_create(&r.a) || msrt::event_create(&r.b)) return 1;
    msrt::event_record(r.a, stream);
    int e = msrt::launch<K>(stream, gx, gy, threads, lds, p);
    msrt::event_record(r.b, stream);
    prof_recs.push_back(r);
    return e;
  }
  int profile_begin() override { prof_on = true; return 0; }
  int profile_end(char* out, size_t cap) override {
    static const char* names[K_COUNT] = {"ntt_pass", "scale_pow", "leaf_hash", "inner_hash", "transpose_in", "io_copy", "lincomb", "mix", "eval", "eval_reduce",
                                         "fold", "suffix_horner", "degree", "find_first", "merkle_path", "query_points"};
    msrt::sync(stream);
    double ms[K_COUNT] = {0}, by[K_COUNT] = {0}, ms_part = 0, ms_repl = 0, repl_by[K_COUNT] = {0}; unsigned long long cnt[K_COUNT] = {0};
    std::map<int, double> sub_ms, sub_by; std::map<int, unsigned long long> sub_cnt;
    for (auto& r : prof_recs) {
      float t = 0.f; msrt::event_elapsed_ms(&t, r.a, r.b);
      ms[r.kid] += t; by[r.kid] += r.bytes; cnt[r.kid]++;
      if (r.part) ms_part += t; else { ms_repl += t; repl_by[r.kid] += t; }
      if (r.kid == K_NTT_PASS) { sub_ms[r.sub] += t; sub_by[r.sub] += r.bytes; sub_cnt[r.sub]++; }
      msrt::event_destroy(r.a); msrt::event_destroy(r.b);
    }
    prof_recs.clear(); prof_on = false;
    std::string j = "{";
    for (int k = 0; k < K_COUNT; k++) {
      char buf[256];
      snprintf(buf, sizeof buf, "%s\"%s\": {\"launches\": %llu, \"ms\": %.6f, \"alg_bytes\": %.0f}", k ? ", " : "", names[k], cnt[k], ms[k], by[k]);
      j += buf;
    }
    // per template instance of the NTT pass kernel (matches the kernel names rocprofv3 reports)
    j += ", \"ntt_pass_variants\": {";
    bool first = true;
    for (auto& kv : sub_ms) {
      const int sub = kv.first; char buf[420];
      const char* name = ntt_names[(size_t)sub < ntt_names.size() ? sub : 0].c_str();
      snprintf(buf, sizeof buf, "%s\"%s\": {\"launches\": %llu, \"ms\": %.6f, \"alg_bytes\": %.0f}", first ? "" : ", ", name, sub_cnt[sub], kv.second, sub_by[sub]);
      j += buf; first = false;
    }
    j += "}";
    { char buf[200]; snprintf(buf, sizeof buf, ", \"shard\": {\"world\": %d, \"partitioned_ms\": %.6f, \"replicated_ms\": %.6f, \"replicated_by_kernel\": {", sh_world, ms_part, ms_repl); j += buf;
      bool first_k = true;
      for (int k = 0; k < K_COUNT; k++) if (repl_by[k] > 0) { snprintf(buf, sizeof buf, "%s\"%s\": %.4f", first_k ? "" : ", ", names[k], repl_by[k]); j += buf; first_k = false; }
      j += "}}}"; }
    if (out && cap) { size_t n = j.size() < cap - 1 ? j.size() : cap - 1; memcpy(out, j.data(), n); out[n] = 0; }
    return 0;
  }

  // ------------------------------------------------------------------ NTT plans
  struct Plan {
    int log_n = 0, log_r0 = 0, log_rho = 0, npass = 0, K[4] = {0, 0, 0, 0}, lo_bits = 0;
    int LC[4] = {msntt::TILE_LOG_C, msntt::TILE_LOG_C, msntt::TILE_LOG_C, msntt::TILE_LOG_C};   // log2 tile columns of every pass
    bool v2[4] = {false, false, false, false};                                                   // pass runs on msntt::PassKernel2
    bool regp[4] = {false, false, false, false};                                                 // pass runs on msntt::RegPassKernel (last pass, radix <= 32, registers only)
    DevBuf tw_lo, tw_hi, w_r[4], vtw, w0;
    T n_inv = 0;
  };
  std::map<int, Plan*> plans;  // key = (log_n*4 + log_pad)*2 + inverse
  DevBuf ntt_scratch;
  int ntt_kmax = 9;            // largest tile (log2 rows) of a multi-pass plan; MS_NTT_KMAX overrides (tuning)
  int ntt_v2 = 1, ntt_v2_min = 14, ntt_v2_maxpass = 2, ntt_v2_sub3 = 1, ntt_v2_regpass = 1;   // MS_NTT_V2=0: round-1 kernels only (A/B); transforms of at least 2^MS_NTT_V2_MIN points use the two-sub-round tiles
  int ntt_colbatch = 0, ntt_mall_mib = 96;    // MS_NTT_COLBATCH: columns per chunk of a multi-pass transform (0: all; -1: as many as keep a chunk within MS_NTT_MALL_MIB); see ntt_run
  int ntt_maxpad = msntt::MAX_LOG_PAD, ntt_maxrho = msntt::MAX_LOG_RHO, ntt_th512 = 1, ntt_fast = 1, ntt_fast_min = 22, ntt_fast_max = 24;  // tuning knobs (MS_NTT_MAXPAD / MS_NTT_MAXRHO / MS_NTT_TH512)

  // log_pad: the input is zero beyond n >> log_pad
  int get_plan(int log_n, int log_pad, bool inverse, Plan** out) {
    int key = (log_n * 4 + log_pad) * 2 + (inverse ? 1 : 0);
    auto it = plans.find(key);
    if (it != plans.end()) { *out = it->second; return 0; }
    if (log_n > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "domain larger than the field's two-adicity");
    Plan* pl = new Plan();
    pl->log_n = log_n;
    const int v2_lc = (F::ID == 0) ? 3 : 4;   // 64-byte tile rows: 8 Goldilocks / 16 BabyBear columns
    bool use_v2 = false;
    // both fields (r03: BabyBear too - LDE 6 x 2^20 -> 2^23 0.505 ms on the round-1 tiles, 0.415 ms on these with two sub-rounds; MS_NTT_V2=0: round-1 tiles)
    if (ntt_v2 >= 1 && log_n >= ntt_v2_min && log_n > msntt::MAX_LOG_R && (log_pad == 0 || log_pad == 3)) {
      // passes of up to 2^10 rows; a blowup-8 evaluation starts behind the virtual radix-8 zero-padding pass (8 tile columns = its 8 cosets)
      const int m = log_n - log_pad, P = (m + 9) / 10;
      const int kb = (ntt_v2_regpass == 2) ? 7 : 10;   // MS_NTT_V2_REGPASS=2 (tests): the same plan shape on 2^7-row tiles, so that 2^15..2^19 points exercise every register radix
      if (ntt_v2_regpass && m - 2 * kb >= 1 && m - 2 * kb <= 5) {
        // 2^21..2^25 non-zero points: 2^10 x 2^10 on the large tiles, then the remaining 2^(m-20) points of every output in registers - a streaming
        // pass at copy speed instead of a third tile pass (MS_NTT_V2_REGPASS=0: the three-pass plan of 2^8-row round-1 tiles)
        use_v2 = true;
        pl->log_rho = 0; pl->log_r0 = log_pad; pl->npass = 3;
        pl->K[0] = kb; pl->K[1] = kb; pl->K[2] = m - 2 * kb;
        pl->LC[0] = log_pad ? 3 : v2_lc; pl->LC[1] = v2_lc; pl->LC[2] = 0;
        pl->v2[0] = pl->v2[1] = true; pl->regp[2] = true;
      } else if (P <= ntt_v2_maxpass && m / P >= 7) {   // measured (r02): two passes of 2^10-row tiles beat three of 2^8; with three or more passes the round-1 tiles win
        use_v2 = true;
        pl->log_rho = 0; pl->log_r0 = log_pad; pl->npass = P;
        for (int i = 0; i < P; i++) {
          pl->K[i] = m / P + (i < m % P ? 1 : 0); pl->LC[i] = (i == 0 && log_pad) ? 3 : v2_lc; pl->v2[i] = true;
        }
      }
    }
    if (use_v2) {}
    else if (log_n <= msntt::MAX_LOG_R) { pl->npass = 1; pl->K[0] = log_n; }
    else {
      // virtual first pass: radix 2^log_pad of pure zero padding times a real radix 2^log_rho (<= 4) over the
      // non-zero blocks; pick the smallest log_rho that minimises the number of real passes
      int best_rho = 0, bestP = 1 << 20;
      for (int lr = 0; lr <= (log_pad ? ntt_maxrho : 0); lr++) {
        const int m = log_n - log_pad - lr;
        if (m < 2) break;
        int P = (m + ntt_kmax - 1) / ntt_kmax;
        if (P < 2 && log_pad == 0) P = 2;
        if (P < 1) P = 1;
        if (P < bestP) { bestP = P; best_rho = lr; }
      }
      // measured on MI355X (tools/ntt_bench.py, r01): zero-padded transforms of 2^22..2^24 points are fastest as plain 3-pass plans
      // on the compile-time specialised tiles (2^8-row tiles); smaller and larger ones with the virtual-pass plans
      if (ntt_fast && bestP > 1 && (log_pad == 0 || (log_n >= ntt_fast_min && log_n <= ntt_fast_max))) {
        best_rho = 0; log_pad = 0;
        bestP = (log_n + ntt_kmax - 1) / ntt_kmax; if (bestP < 2) bestP = 2;
      }
      pl->log_rho = best_rho; pl->log_r0 = log_pad + best_rho;
      const int m = log_n - pl->log_r0, P = bestP;
      pl->npass = P;
      for (int i = 0; i < P; i++) pl->K[i] = m / P + (i < m % P ? 1 : 0);
    }
    const int log_r0 = pl->log_r0;
    T wn = f_root_of_unity<F>(log_n);
    if (inverse) wn = f_inv<F>(wn);
    const size_t n = (size_t)1 << log_n;
    pl->n_inv = f_inv<F>(F::from_u64(n % F::P));
    pl->lo_bits = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << pl->lo_bits, nhi = (size_t)1 << (log_n - pl->lo_bits);
    std::vector<T> lo(nlo), hi(nhi);
    T x = F::from_u64(1);
    for (size_t j = 0; j < nlo; j++) { lo[j] = F::to_tw(x); x = F::mul(x, wn); }
    T wh = x;  // w^nlo
    x = F::from_u64(1);
    for (size_t j = 0; j < nhi; j++) { hi[j] = F::to_tw(x); x = F::mul(x, wh); }
    if (pl->tw_lo.ensure(nlo * sizeof(T)) || pl->tw_hi.ensure(nhi * sizeof(T))) { delete pl; return fail(MS_ERR_NOMEM, "twiddle tables"); }
    CK(msrt::h2d(pl->tw_lo.p, lo.data(), nlo * sizeof(T), stream));
    CK(msrt::h2d(pl->tw_hi.p, hi.data(), nhi * sizeof(T), stream));
    CK(msrt::sync(stream));
    for (int i = 0; i < pl->npass; i++) {
      const size_t r = (size_t)1 << pl->K[i];
      T wr = f_root_of_unity<F>(pl->K[i]);
      if (inverse) wr = f_inv<F>(wr);
      std::vector<T> tab(r);
      x = F::from_u64(1);
      for (size_t j = 0; j < r; j++) { tab[j] = F::to_tw(x); x = F::mul(x, wr); }
      if (pl->w_r[i].ensure(r * sizeof(T))) { delete pl; return fail(MS_ERR_NOMEM, "w_r table"); }
      CK(msrt::h2d(pl->w_r[i].p, tab.data(), r * sizeof(T), stream));
      CK(msrt::sync(stream));
    }
    if (pl->log_r0) {  // w_r0^j
      const size_t cnt = (size_t)1 << log_r0;
      T wv = f_root_of_unity<F>(log_r0);
      if (inverse) wv = f_inv<F>(wv);
      std::vector<T> tab(cnt);
      x = F::from_u64(1);
      for (size_t j = 0; j < cnt; j++) { tab[j] = F::to_tw(x); x = F::mul(x, wv); }
      if (pl->w0.ensure(cnt * sizeof(T))) { delete pl; return fail(MS_ERR_NOMEM, "w0 table"); }
      CK(msrt::h2d(pl->w0.p, tab.data(), cnt * sizeof(T), stream));
      CK(msrt::sync(stream));
    }
    if (pl->log_r0) {  // w_(r0 r)^j for the virtual first pass folded into real pass 0
      const int lg = pl->log_r0 + pl->K[0];
      const size_t cnt = (size_t)1 << lg;
      T wv = f_root_of_unity<F>(lg);
      if (inverse) wv = f_inv<F>(wv);
      std::vector<T> tab(cnt);
      x = F::from_u64(1);
      for (size_t j = 0; j < cnt; j++) { tab[j] = F::to_tw(x); x = F::mul(x, wv); }
      if (pl->vtw.ensure(cnt * sizeof(T))) { delete pl; return fail(MS_ERR_NOMEM, "vtw table"); }
      CK(msrt::h2d(pl->vtw.p, tab.data(), cnt * sizeof(T), stream));
      CK(msrt::sync(stream));
    }
    plans[key] = pl;
    *out = pl;
    return 0;
  }

  template <bool INV, int K, int TH>
  int launch_fast(const msntt::PassParams<F>& pp, size_t tiles, size_t batch) {
    typedef msntt::PassKernelK<F, INV, K, TH> KK;
    name_next("msntt::PassKernelK<%s, %s, %d, %d>", fname(), INV ? "true" : "false", K, TH);
    return run<KK>(K_NTT_PASS, (unsigned)tiles, (unsigned)batch, TH, KK::lds_bytes(), pp);
  }
  // PassKernel2 instance for a pass of 2^K rows x 2^LC columns in MODE (ntt.hpp): tile shape -> threads, sub-rounds, arithmetic class
  template <class KK> int launch_v2i(const msntt::PassParams<F>& pp, unsigned grid, int nsub, int mode, int k, int lc) {
    if (!KK::applicable(pp)) return 998;   // NTT plan / PassKernel2 mismatch (a bug, not a runtime condition)
    name_next("msntt::PassKernel2<%s, %s, %s, %d, %d, %d, %d, %d>", fname(), aname<typename KK::Arith>(), KK::INVERSE ? "true" : "false", k, lc, KK::THREADS, nsub, mode);
    return run_coop<KK>(K_NTT_PASS, grid, KK::THREADS, KK::lds_bytes(), pp);
  }
  template <bool INV, int K, int LC, int MODE>
  int launch_v2m(const msntt::PassParams<F>& pp, size_t tiles, size_t batch) {
    if constexpr (F::ID == 0 && (K == 8 || K == 9) && LC == 3) {
      if (ntt_v2_sub3) {   // smaller tiles of the three-pass plans: 8 elements per thread, radix 8 x 8 x 4 (or 8), many workgroups per CU
        constexpr int TH3 = (K == 8) ? 256 : 512;
        return launch_v2i<msntt::PassKernel2<F, GLM, INV, K, LC, TH3, 3, MODE>>(pp, coop_grid(tiles * batch, K == 8 ? 8 : 4), 3, MODE, K, LC);
      }
    }
    if constexpr (F::ID == 0 && K == 10 && LC == 3) {
      if (ntt_v2_sub3)     // 512 threads, three sub-rounds, exec-masked arithmetic: 16 waves per CU
        return launch_v2i<msntt::PassKernel2<F, GLM, INV, K, LC, 512, 3, MODE>>(pp, coop_grid(tiles * batch), 3, MODE, K, LC);
    }
    if constexpr (F::ID == 1 && K == 10) {
      // BabyBear, 64 KiB tiles of 16 columns (MODE 0 / 1) or 32 KiB tiles of the 8 cosets (MODE 2): three sub-rounds (radix 16 x 8 x 8) with the
      // fused tail - 512 threads x 32 elements, resp. 256 threads x 32 elements and four workgroups per CU: 16 waves per CU either way
      if (ntt_v2_sub3) {
        constexpr int THB = (LC == 4) ? 512 : 256;
        return launch_v2i<msntt::PassKernel2<F, BB, INV, K, LC, THB, 3, MODE>>(pp, coop_grid(tiles * batch, LC == 4 ? 2 : 4), 3, MODE, K, LC);
      }
    }
    return launch_v2i<msntt::PassKernel2<F, typename msntt::NttArith<F>::type, INV, K, LC, 256, 2, MODE>>(pp, coop_grid(tiles * batch), 2, MODE, K, LC);
  }
  // persistent grid of the cooperative pass kernels: two workgroups per CU (their 72-80 KiB of LDS), a multiple of 8 (XCD-aware tile walk)
  int ntt_coop_wgs = 512, ntt_share = 1;   // MS_NTT_SHARE=0: never the shared-table instance (A/B)
  unsigned coop_grid(size_t work_items, int per_cu = 2) const { const size_t g = (size_t)ntt_coop_wgs * per_cu / 2; return (unsigned)(work_items < g ? work_items : g); }
  template <bool INV, int K, int LC>
  int launch_v2(const msntt::PassParams<F>& pp, size_t tiles, size_t batch) {
    if (pp.log_r0) {
      if constexpr (LC == 3) {
        // Goldilocks, several columns, enough tiles to feed the persistent grid one tile at a time: the instance that builds the per-tile twiddle tables once
        // per tile for all columns (MODE 3; measured r03: six-column LDE 0.560 -> 0.555 ms; BabyBear 0.333 -> 0.375 ms, so not for it)
        if constexpr (F::ID == 0 && K == 10) { if (ntt_v2_sub3 && ntt_share && batch > 1 && tiles >= (size_t)coop_grid(tiles * batch) && (tiles & 7) == 0) return launch_v2m<INV, K, LC, 3>(pp, tiles, batch); }
        return launch_v2m<INV, K, LC, 2>(pp, tiles, batch);
      } else return 996;
    }
    if (pp.log_Rp == 0) return launch_v2m<INV, K, LC, 0>(pp, tiles, batch);
    return launch_v2m<INV, K, LC, 1>(pp, tiles, batch);
  }
  template <bool INV, int LC>
  int launch_v2k(const msntt::PassParams<F>& pp, size_t tiles, size_t batch) {
    switch (pp.log_r) {
      case 7: return launch_v2<INV, 7, LC>(pp, tiles, batch);
      case 8: return launch_v2<INV, 8, LC>(pp, tiles, batch);
      case 9: return launch_v2<INV, 9, LC>(pp, tiles, batch);
      case 10: return launch_v2<INV, 10, LC>(pp, tiles, batch);
      default: return 999;
    }
  }
  template <bool INV, int K>
  int launch_reg(const msntt::PassParams<F>& pp, size_t batch) {
    typedef msntt::RegPassKernel<F, typename std::conditional<F::ID == 0, GLM, F>::type, INV, K> KK;
    if (!KK::applicable(pp)) return 995;
    name_next("msntt::RegPassKernel<%s, %s, %s, %d>", fname(), F::ID == 0 ? "GLM" : "BB", INV ? "true" : "false", K);
    return run<KK>(K_NTT_PASS, KK::grid(pp), (unsigned)batch, KK::THREADS, 0, pp);
  }
  template <bool INV>
  int launch_pass(const msntt::PassParams<F>& pp, size_t tiles, size_t batch, bool v2 = false, bool regp = false) {
    if (regp) {
      switch (pp.log_r) {
        case 1: return launch_reg<INV, 1>(pp, batch);
        case 2: return launch_reg<INV, 2>(pp, batch);
        case 3: return launch_reg<INV, 3>(pp, batch);
        case 4: return launch_reg<INV, 4>(pp, batch);
        case 5: return launch_reg<INV, 5>(pp, batch);
        default: return 994;
      }
    }
    if (v2) {
      if constexpr (F::ID == 0) return pp.log_C == 3 ? launch_v2k<INV, 3>(pp, tiles, batch) : 997;
      else return pp.log_C == 3 ? launch_v2k<INV, 3>(pp, tiles, batch) : launch_v2k<INV, 4>(pp, tiles, batch);
    }
    // compile-time specialised tiles for the large transforms (no virtual pass, 16 columns)
    if (ntt_fast && pp.log_C == msntt::TILE_LOG_C && pp.log_r0 == 0 && (pp.log_Rp == 0 || pp.log_Rp >= msntt::TILE_LOG_C)) {
      switch (pp.log_r) {
        case 6: return launch_fast<INV, 6, 256>(pp, tiles, batch);
        case 7: return launch_fast<INV, 7, 256>(pp, tiles, batch);
        case 8: return launch_fast<INV, 8, 256>(pp, tiles, batch);
        case 9: return launch_fast<INV, 9, 512>(pp, tiles, batch);
        default: break;
      }
    }
    const size_t lds = msntt::PassKernel<F, INV, 256>::lds_bytes(pp.log_r, pp.log_C, pp.log_Rp, pp.last != 0);
    const bool th512 = ntt_th512 && pp.log_r >= 9 && pp.log_C == msntt::TILE_LOG_C;   // 8192-element tiles: 16 elements per thread
    name_next("msntt::PassKernel<%s, %s, %d>", fname(), INV ? "true" : "false", th512 ? 512 : 256);
    if (th512) {
      return run<msntt::PassKernel<F, INV, 512>>(K_NTT_PASS, (unsigned)tiles, (unsigned)batch, 512, lds, pp);
    }
    return run<msntt::PassKernel<F, INV, 256>>(K_NTT_PASS, (unsigned)tiles, (unsigned)batch, 256, lds, pp);
  }

  // batch transforms of size 2^log_n: src (n_in valid elements per entry, zero padded) -> dst
  int ntt_run(int log_n, bool inverse, const T* src, size_t src_bstride, size_t n_in, T* dst, size_t dst_bstride, size_t batch) {
    if (batch == 0) return 0;
    const size_t n = (size_t)1 << log_n;
    if (log_n == 0) {  // size-1 transform: identity
      for (size_t b = 0; b < batch; b++) {
        if (n_in >= 1) { if (src + b * src_bstride != dst + b * dst_bstride) CK(msrt::d2d(dst + b * dst_bstride, src + b * src_bstride, sizeof(T), stream)); }
        else CK(msrt::memset_dev(dst + b * dst_bstride, 0, sizeof(T), stream));
      }
      return 0;
    }
    // zero padding: n_in <= n >> log_pad  (log_pad <= 3)
    int log_pad = 0;
    while (log_pad < ntt_maxpad && (n_in << (log_pad + 1)) <= n) log_pad++;
    Plan* pl;
    RQ(get_plan(log_n, log_pad, inverse, &pl));
    const int log_r0 = pl->log_r0;
    const int P = pl->npass;
    const bool aliased = ((const void*)src == (const void*)dst);
    T* scr = nullptr;
    const bool last_inplace = P >= 2 && !(aliased && P % 2 == 0);
    const bool need_scr = P >= 3 || (P == 2 && !last_inplace);
    if (need_scr) {
      if (ntt_scratch.ensure(batch * n * sizeof(T))) return fail(MS_ERR_NOMEM, "ntt scratch");
      scr = ntt_scratch.as<T>();
    }
    // Column chunks (MS_NTT_COLBATCH = columns per chunk; 0 = the default: all columns per launch; -1: as many as fit MS_NTT_MALL_MIB):
    // the passes of a multi-pass plan run chunk by chunk, so that what one pass writes is still in the 256 MiB Infinity Cache when the
    // next pass reads it back.  Measured (r02, profiles/r02_stride_probe.log, r02_colbatch.log): the later pass's access PATTERN alone
    // takes 24.5 us per resident 64 MiB column against 47.7 us when six columns are swept per launch - but the complete kernel takes
    // 45 us per column either way (a one-column launch is no faster than a third of a three-column one): the pass is bound by exposed
    // latency at 4 waves per SIMD, not by HBM or by the pattern, so chunking buys nothing (LDE 6 x 2^20: 0.641 ms chunked, 0.627 not).
    size_t chunk = batch;
    if (P >= 2 && batch > 1) {
      if (ntt_colbatch > 0) chunk = (size_t)ntt_colbatch;
      else if (ntt_colbatch < 0) { const size_t col = n * sizeof(T), budget = (size_t)ntt_mall_mib << 20; chunk = col >= budget ? (col <= ((size_t)128 << 20) ? 1 : batch) : budget / col; }
      if (chunk < 1 || chunk > batch) chunk = batch;
    }
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
    const size_t nb = batch - b0 < chunk ? batch - b0 : chunk;
    const T* in = src + b0 * src_bstride; size_t in_bs = src_bstride;
    int log_Rp = log_r0;
    for (int k = 0; k < P; k++) {
      T* out; size_t out_bs;
      if (k == P - 1) { out = dst + b0 * dst_bstride; out_bs = dst_bstride; }
      else {
        const bool even = ((P - 1 - (k + 1)) % 2 == 0);
        const bool to_dst = last_inplace ? even : !even;
        out = to_dst ? dst + b0 * dst_bstride : scr + b0 * n; out_bs = to_dst ? dst_bstride : n;
      }
      msntt::PassParams<F> pp;
      pp.src = in; pp.dst = out; pp.src_bstride = in_bs; pp.dst_bstride = out_bs;
      pp.n_in = (k == 0) ? n_in : n;
      pp.tw_lo = pl->tw_lo.template as<T>(); pp.tw_hi = pl->tw_hi.template as<T>(); pp.w_r = pl->w_r[k].template as<T>();
      pp.vtw = pl->vtw.template as<T>(); pp.w0 = pl->w0.template as<T>();
      pp.do_scale = (inverse && k == P - 1) ? 1 : 0;
      pp.scale = F::to_tw(pp.do_scale ? pl->n_inv : F::from_u64(1));
      pp.log_n = log_n; pp.log_r = pl->K[k]; pp.log_Rp = log_Rp; pp.lo_bits = pl->lo_bits;
      pp.log_r0 = (k == 0) ? log_r0 : 0; pp.log_rho = (k == 0) ? pl->log_rho : 0;
      const int cols_log = log_n - pl->K[k];
      pp.log_C = cols_log < pl->LC[k] ? cols_log : pl->LC[k];
      pp.last = (k == P - 1); pp.nbatch = (u32)nb;
      const size_t tiles = ((size_t)1 << cols_log) >> pp.log_C;
      next_bytes = (double)(n_in + n) * nb * sizeof(T) / P;  // (n_in + n)*s per transform, shared by its P real passes
      if (inverse) CK(launch_pass<true>(pp, tiles, nb, pl->v2[k], pl->regp[k])); else CK(launch_pass<false>(pp, tiles, nb, pl->v2[k], pl->regp[k]));
      in = out; in_bs = out_bs;
      log_Rp += pl->K[k];
    }
    }
    return 0;
  }

  // ------------------------------------------------------------------ pool of zeroed device words
  // Counters that kernels bump (deferred-block lists, the degree result) must start at zero.  One memset clears a 256 KiB
  // pool; zero_alloc hands out fresh pieces of it and clears it again only when it runs out (stream order keeps earlier
  // users ahead of the clear) — a proof needs ~50 such counters, i.e. one memset instead of ~50 four-microsecond fills.
  // page-locked staging for the per-proof job tables (r03): uploads out of it are plain DMA, not the runtime's pageable-memory path (which pins or stages the
  // caller's pages on the fly), and need no synchronisation of their own - the area is rewritten by the NEXT proof's same stage, behind that stage's final
  // stream synchronisation
  void* h_tabs = nullptr; size_t h_tabs_cap = 0;
  int tabs_host(size_t bytes, u8** out) {
    CK(msrt::sync(stream));                         // an upload out of the area may still be in flight if the stage that issued it left on an error path; free otherwise: the stream is idle here
    if (bytes > h_tabs_cap) {
      if (h_tabs) msrt::free_host(h_tabs);
      h_tabs = nullptr; h_tabs_cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      if (msrt::malloc_host(&h_tabs, want)) { h_tabs = nullptr; return fail(MS_ERR_NOMEM, "page-locked table staging"); }
      h_tabs_cap = want;
    }
    *out = reinterpret_cast<u8*>(h_tabs);
    return 0;
  }
  DevBuf d_zero; size_t zero_used = 0, zero_cap = 0;
  int zero_alloc(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (!d_zero.p) { if (d_zero.ensure(256 << 10)) return fail(MS_ERR_NOMEM, "zero pool"); zero_cap = 256 << 10; zero_used = zero_cap; }
    if (bytes > zero_cap) return fail(MS_ERR_NOMEM, "zero pool request too large");
    if (zero_used + bytes > zero_cap) { CK(msrt::memset_dev(d_zero.p, 0, zero_cap, stream)); zero_used = 0; }
    *out = d_zero.as<u8>() + zero_used; zero_used += bytes;
    return 0;
  }

  // ------------------------------------------------------------------ Merkle
  // sharded (ms_set_shard): this rank holds the subtree over leaf groups [rank*Mloc, (rank+1)*Mloc) followed by the replicated top
  // (world subtree roots and the levels above); local_nodes = nodes held here, root last in both cases
  struct TreeShape { size_t leaf_num = 0, lpn = 0, ic = 0, levels = 0, nodes = 0, local_nodes = 0, Mloc = 0; bool sharded = false; };
  // src/merkle.rs:89-118 (shape checks and node count)
  int tree_shape(size_t leaf_num, size_t lpn, size_t ic, TreeShape* ts) {
    if (lpn == 0 || ic < 2 || !is_pow2(ic)) return fail(MS_ERR_SHAPE, "merkle: bad leafs_per_node / inner_children");
    const size_t node_num = leaf_num / lpn;
    long lg = log_two_k(node_num, ic);
    if (lg < 0) return fail(MS_ERR_SHAPE, lg == -1 ? "number if not a power of 2" : "number if not a power of base");
    if (leaf_num % lpn != 0) return fail(MS_ERR_SHAPE, "merkle: leaf_num % leafs_per_node != 0");
    if (lg >= 64 || node_num == 0) return fail(MS_ERR_SHAPE, "Tree is not full!");
    ts->leaf_num = leaf_num; ts->lpn = lpn; ts->ic = ic; ts->levels = (size_t)lg + 1;
    size_t total = 0, m = node_num;
    for (;;) { total += m; if (m == 1) break; m /= ic; }
    ts->nodes = total; ts->local_nodes = total; ts->Mloc = 0; ts->sharded = false;
    return 0;
  }
  // leaf-group digests of `ngroups` groups into `out`: LeafHashKernel + the compacted pad-only blocks it deferred
  template <int EL>
  int leaf_hash(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, size_t lpn, size_t ngroups, u32* out,
                size_t g_first = 0, u32 run_len = 0, u32 run_stride = 0, const msmerkle::LinColSpec* lin = nullptr, size_t out_g0 = 0) {
    if (ngroups >> 32) return fail(MS_ERR_SHAPE, "more than 2^32 leaf groups");
    // deferred pad-only blocks: OVF_LISTS lists, list l fed by the workgroups bx = l (mod OVF_LISTS); capacity = all their threads
    const size_t nwg = grid1(ngroups, msmerkle::THREADS), lists = msmerkle::OVF_LISTS;
    const size_t cap = ((nwg + lists - 1) / lists) * msmerkle::THREADS;
    if (d_ovf.ensure(lists * cap * msmerkle::OVF_WORDS * 4)) return fail(MS_ERR_NOMEM, "deferred-block lists");
    void* counters;
    RQ(zero_alloc(lists * 4, &counters));
    typename msmerkle::LeafHashKernel<F, EL>::Params lp;
    lp.base = base; lp.col_stride = col_stride; lp.row_stride = row_stride; lp.limb_stride = limb_stride;
    lp.width = width; lp.lpn = (u32)lpn; lp.zero_as_empty = zae; lp.ngroups = ngroups; lp.nodes = out;
    lp.ovf_count = reinterpret_cast<u32*>(counters); lp.ovf = d_ovf.as<u32>(); lp.ovf_cap = (u32)cap;
    lp.g_first = g_first; lp.run_len = run_len; lp.run_stride = run_stride; lp.lin = lin; lp.out_g0 = out_g0;
    next_bytes = (double)ngroups * (lpn * EL * sizeof(T) + 32);
    if (lpn * EL >= (size_t)leaf_lazy_min) {  // long messages (wide rows): the two-block buffer that compresses wave-synchronously
      typedef msmerkle::LeafHashKernel<F, EL, true> LK;
      typename LK::Params ll;
      ll.base = lp.base; ll.col_stride = lp.col_stride; ll.row_stride = lp.row_stride; ll.limb_stride = lp.limb_stride; ll.width = lp.width; ll.lpn = lp.lpn;
      ll.zero_as_empty = lp.zero_as_empty; ll.ngroups = lp.ngroups; ll.nodes = lp.nodes; ll.ovf_count = lp.ovf_count; ll.ovf = lp.ovf; ll.ovf_cap = lp.ovf_cap;
      ll.g_first = g_first; ll.run_len = run_len; ll.run_stride = run_stride; ll.lin = lin; ll.out_g0 = out_g0;
      CK(run<LK>(K_LEAF_HASH, grid1(ngroups, msmerkle::THREADS), 1, msmerkle::THREADS, LK::lds_bytes(), ll));
    } else
    CK(run<msmerkle::LeafHashKernel<F, EL>>(K_LEAF_HASH, grid1(ngroups, msmerkle::THREADS), 1, msmerkle::THREADS, msmerkle::LeafHashKernel<F, EL>::lds_bytes(), lp));
    msmerkle::PadOnlyBlockKernel::Params pp{lp.ovf_count, lp.ovf, (u32)cap, out};
    const size_t used = nwg < lists ? nwg : lists, per_list = grid1(cap, msmerkle::THREADS);
    CK(run<msmerkle::PadOnlyBlockKernel>(K_LEAF_HASH, (unsigned)used, (unsigned)(per_list < (size_t)msmerkle::PAD_GRID_Y ? per_list : (size_t)msmerkle::PAD_GRID_Y), msmerkle::THREADS, 0, pp));
    return 0;
  }
  template <int EL>
  int tree_build(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, const TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin = nullptr) {
    if (nodes.ensure(ts.nodes * 32)) return fail(MS_ERR_NOMEM, "merkle nodes");
    const size_t ngroups = ts.leaf_num / ts.lpn;
    RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, ngroups, nodes.as<u32>(), 0, 0, 0, lin)));
    RQ(inner_levels(nodes.as<u32>(), ngroups, ts.ic));
    return 0;
  }
  // inner levels above `nchildren` digests at nodes[0..): level-major, root last (merkle.rs:131-140)
  // `final_levels`: these levels end in the tree's root, which the last launch also stores to the page-locked slot host_root()
  // (root_on_host: read_root / read_degree_and_root then need no copy launch, only the stream synchronisation they do anyway)
  u32* host_root() const { return reinterpret_cast<u32*>(reinterpret_cast<u8*>(pinned) + 256); }
  bool root_on_host = false;
  unsigned long long* pending_aux = nullptr; bool aux_on_host = false;   // a device word the tree's final launch forwards to pinned[0] (the degree result)
  int inner_levels(u32* nodes, size_t nchildren, size_t ic, bool final_levels = true) {
    size_t child_off = 0;
    if (final_levels) root_on_host = false;
    while (nchildren > 1) {
      msmerkle::InnerHashKernel::Params ip;
      ip.nodes = nodes; ip.child_off = child_off; ip.nchildren = nchildren; ip.ic = (u32)ic; ip.host_root = nullptr; ip.aux_src = nullptr; ip.aux_dst = nullptr;
      const size_t nparents = nchildren / ic;
      if (ic == 2 && nparents <= subtree_parents && (nchildren & (nchildren - 1)) == 0) {   // latency-bound levels: up to 9 of them per launch, children in LDS
        typedef msmerkle::InnerSubtreeKernel SK;
        u32 nl = 0; for (size_t m = nchildren; m > 1 && nl < (u32)SK::MAX_LEVELS; m >>= 1) nl++;
        const size_t left = nchildren >> nl;
        if (final_levels && left == 1) {
          ip.host_root = host_root(); root_on_host = true;
          if (pending_aux) { ip.aux_src = pending_aux; ip.aux_dst = reinterpret_cast<unsigned long long*>(pinned); pending_aux = nullptr; aux_on_host = true; }
        }
        ip.nlevels = nl;
        next_bytes = (double)nchildren * 32 * 2;
        CK(run_coop<SK>(K_INNER_HASH, (unsigned)left, SK::THREADS, SK::lds_bytes(), ip));
        for (u32 l = 0; l < nl; l++) { child_off += nchildren; nchildren >>= 1; }
        continue;
      }
      if (final_levels && (nparents == 1 || nparents <= (size_t)tree_top_parents)) {
        ip.host_root = host_root(); root_on_host = true;
        if (pending_aux) { ip.aux_src = pending_aux; ip.aux_dst = reinterpret_cast<unsigned long long*>(pinned); pending_aux = nullptr; aux_on_host = true; }
      }
      if (nparents <= (size_t)tree_top_parents) {  // fused tree top: one workgroup walks the remaining levels
        u32 nl = 0; for (size_t m = nchildren; m > 1; m /= ic) nl++;
        ip.nlevels = nl;
        next_bytes = (double)nchildren * 32 * 2;
        if (ic == 2) CK(run<msmerkle::InnerHashKernel2>(K_INNER_HASH, 1, 1, msmerkle::THREADS, 0, ip));
        else CK(run<msmerkle::InnerHashKernel>(K_INNER_HASH, 1, 1, msmerkle::THREADS, 0, ip));
        break;
      }
      ip.nlevels = 1;
      next_bytes = (double)nparents * (ic * 32 + 32);
      if (ic == 2) CK(run<msmerkle::InnerHashKernel2>(K_INNER_HASH, (unsigned)((nparents + msmerkle::THREADS - 1) / msmerkle::THREADS), 1, msmerkle::THREADS, 0, ip));
      else CK(run<msmerkle::InnerHashKernel>(K_INNER_HASH, (unsigned)((nparents + msmerkle::THREADS - 1) / msmerkle::THREADS), 1, msmerkle::THREADS, 0, ip));
      child_off += nchildren; nchildren = nparents;
    }
    return 0;
  }
  // Sharded MerkleTree::new over a binary tree of M = leaf_num/lpn leaf groups, of which this rank hashes the groups
  // j = rank + W*i found at local group index i of the view (base, strides): digest all-to-all, subtree, root all-gather, top.
  template <int EL>
  int tree_build_sharded(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin = nullptr) {
    const size_t W = (size_t)sh_world, M = ts.leaf_num / ts.lpn, Mloc = M / W, per = Mloc / W;
    if (ts.ic != 2 || per == 0) return fail(MS_ERR_STATE, "sharded tree needs a binary tree with at least world^2 leaf groups");
    if (Mloc * 32 > xcap) return fail(MS_ERR_NOMEM, "exchange buffers too small for the sharded commitment (need 32 * leaf groups / world bytes)");
    const size_t sub_nodes = 2 * Mloc - 1, top_nodes = 2 * W - 1;
    if (nodes.ensure((sub_nodes + top_nodes) * 32)) return fail(MS_ERR_NOMEM, "merkle nodes");
    // The digest all-to-all overlaps the leaf hashing (r03): the local groups are hashed in `S` slices - slice s = the s-th part of EVERY peer's chunk - and
    // the digests of slice s travel (RCCL: on the context's communication stream, ordered by events) while slice s + 1 is hashed.  MS_SHARD_SLICES (4) /
    // MS_SHARD_SLICE_MIN (1024 groups per peer and slice; below that the commitment goes out in one piece).
    const size_t want_slices = (rccl_comm && !shard_slices_set) ? 1 : (size_t)shard_slices;
    const size_t S = (want_slices > 1 && per % want_slices == 0 && per / want_slices >= shard_slice_min) ? want_slices : 1;
    PartScope part(this);
    if (S == 1) {
      RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, Mloc, reinterpret_cast<u32*>(xs), 0, 0, 0, lin)));
      RQ(exchange(MS_XCHG_ALL_TO_ALL, per * 32));
    } else {
      const size_t q = per / S;
      for (size_t sl = 0; sl < S; sl++) {
        RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, W * q, reinterpret_cast<u32*>(xs), sl * q, (u32)q, (u32)per, lin)));
        RQ(exchange_slice(sl * q * 32, per * 32, q * 32, (int)sl, (int)S));
      }
    }
    msmerkle::InterleaveDigestsKernel::Params ik{reinterpret_cast<const msmerkle::uint4_t*>(xr), reinterpret_cast<msmerkle::uint4_t*>(nodes.p), per, (u32)W};
    CK(run<msmerkle::InterleaveDigestsKernel>(K_IO, grid1(Mloc * 2, msmerkle::InterleaveDigestsKernel::THREADS), 1, msmerkle::InterleaveDigestsKernel::THREADS, 0, ik));
    return finish_sharded_tree(ts, nodes, Mloc);
  }
  // the subtree over this rank's Mloc contiguous leaf digests (at nodes[0..)), the all-gather of the W subtree roots, the replicated top.
  // shard_aux (a device word, optional): rides on the root all-gather; every rank gets the maximum over the ranks back in the same word.
  unsigned long long* shard_aux = nullptr;
  int finish_sharded_tree(TreeShape& ts, DevBuf& nodes, size_t Mloc) {
    const size_t W = (size_t)sh_world, sub_nodes = 2 * Mloc - 1, top_nodes = 2 * W - 1;
    constexpr size_t REC = msmerkle::ShardTopKernel::REC;
    if (W * REC > xcap) return fail(MS_ERR_NOMEM, "exchange buffers too small for the subtree roots");
    { PartScope part(this); RQ(inner_levels(nodes.as<u32>(), Mloc, 2, false)); }
    CK(msrt::memset_dev(xs, 0, REC, stream));
    CK(msrt::d2d(xs, nodes.as<u8>() + (sub_nodes - 1) * 32, 32, stream));
    if (shard_aux) CK(msrt::d2d(xs + 32, shard_aux, 8, stream));
    RQ(exchange(MS_XCHG_ALL_GATHER, REC));
    u8* top = nodes.as<u8>() + sub_nodes * 32;
    msmerkle::ShardTopKernel::Params tk{xr, (u32)W, reinterpret_cast<u32*>(top), shard_aux};
    CK(run<msmerkle::ShardTopKernel>(K_IO, 1, 1, msmerkle::ShardTopKernel::THREADS, 0, tk));
    shard_aux = nullptr;
    RQ(inner_levels(reinterpret_cast<u32*>(top), W, 2));
    ts.sharded = true; ts.Mloc = Mloc; ts.local_nodes = sub_nodes + top_nodes;
    return 0;
  }
  // MerkleTree::new over data EVERY rank holds (the raw trace): rank k hashes the contiguous leaf groups [k*M/W, (k+1)*M/W) - no digest exchange at all -
  // builds that subtree, and the ranks all-gather the W subtree roots
  template <int EL>
  int tree_build_sharded_contiguous(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes) {
    const size_t W = (size_t)sh_world, M = ts.leaf_num / ts.lpn, Mloc = M / W;
    if (ts.ic != 2 || Mloc == 0 || (Mloc >> 32)) return fail(MS_ERR_STATE, "sharded tree needs a binary tree with at least world leaf groups");
    if (nodes.ensure((2 * Mloc - 1 + 2 * W - 1) * 32)) return fail(MS_ERR_NOMEM, "merkle nodes");
    // group g of the launch is leaf group rank*Mloc + g (one run of Mloc groups); its digest lands at nodes[g] (out_g0 = the rank's first group)
    PartScope part(this);
    RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, Mloc, nodes.as<u32>(), (size_t)sh_rank * Mloc, (u32)Mloc, 0, nullptr, (size_t)sh_rank * Mloc)));
    return finish_sharded_tree(ts, nodes, Mloc);
  }
  // root of the tree built LAST on this context (every caller reads it right behind the build)
  int read_root(const DevBuf& nodes, const TreeShape& ts, u8* root) {
    const bool on_host = root_on_host;
    if (!on_host) CK(msrt::d2h(pinned, nodes.as<u8>() + (ts.local_nodes - 1) * 32, 32, stream));
    CK(msrt::sync(stream));
    memcpy(root, on_host ? reinterpret_cast<const void*>(host_root()) : pinned, 32);
    return 0;
  }

  // ------------------------------------------------------------------ session state
  size_t N = 0, w = 0, L = 0, blowup = 0;
  int npolys = 0; size_t polys_cap = 0;
  struct Lin { std::vector<u64> s; std::vector<int> idx; };   // provenance of polynomial i: a linear combination of earlier ones (empty: none)
  std::vector<Lin> poly_lin;
  // r04, linear provenance carried through the coefficient domain: a polynomial ms_polys_lincomb defined is not computed when it is defined.  Its LDE column is the
  // combination of the computed LDE columns (as before), the mix is ONE linear combination of the polynomials without provenance with the scalars
  // sum_i r^i * (coefficient of that polynomial in f_i), and its DEEP-ALI value is the same combination of the opened values - exact field arithmetic, so every
  // output is bit-identical.  The coefficient vector is materialised on demand only (ms_poly_read, MS_LDE_LINEAR=0).  MS_LAZY_LINCOMB=0: computed at definition (r03).
  std::vector<char> poly_mat; int lazy_lin = 1;
  void expand(int i, T scale, std::map<int, T>& acc) const {   // f_i as a combination of the polynomials without provenance
    const Lin& li = poly_lin[i];
    if (li.idx.empty()) { auto it = acc.find(i); if (it == acc.end()) acc[i] = scale; else it->second = F::add(it->second, scale); return; }
    for (size_t t = 0; t < li.idx.size(); t++) expand(li.idx[t], F::mul(scale, F::from_u64(li.s[t] % F::P)), acc);
  }
  int materialize(int i) {
    if (poly_mat[i]) return 0;
    const Lin& li = poly_lin[i];
    for (int ix : li.idx) RQ(materialize(ix));
    RQ(lincomb_into(d_polys.as<T>(), N, N, li.s.data(), li.idx.data(), (int)li.idx.size(), i, d_polys.as<T>() + (size_t)i * N));
    poly_mat[i] = 1;
    return 0;
  }
  bool have_trace = false, have_polys = false, have_lde = false, have_validity = false;
  DevBuf d_trace, d_polys, d_coef, d_lde, d_trace_nodes, d_lde_nodes, d_io, d_partials, d_small;
  TreeShape trace_ts, lde_ts;
  size_t lde_c = 0;

  // dist (one proof over several ranks, r04): the round polynomial is worked on BY COEFFICIENT RANGE - rank k owns the coefficients [k*S, (k+1)*S), S = D / (blowup * world).
  // Round 0 keeps the whole (replicated) validity polynomial in `poly` and every rank uses its range of it; later rounds hold only their own S coefficients
  // (local_store: limb l of coefficient k*S + i at poly[l*S + i]).
  struct Round { DevBuf poly, cw, nodes; size_t cap = 0, ncoef = 0, D = 0; TreeShape ts; size_t m = 0; /* sharded: local codeword = limbs x 2 cosets x m */
                 bool dist = false, local_store = false; size_t S = 0; };
  size_t shard_gather_chunk = 0;
  int shard_dist = 1;          // MS_SHARD_DIST=0: the coefficient-domain work of a sharded proof stays replicated on every rank (r03 behaviour; A/B and tests)
  int proof_root_only = 0;     // ms_shard_proof_on_root: the FRI proof blob is assembled on rank 0 only
  const T* lpoly(const Round* r) const { return r->local_store ? r->poly.template as<T>() : r->poly.template as<T>() + (size_t)sh_rank * r->S; }
  size_t lstride(const Round* r) const { return r->local_store ? r->S : r->cap; }
  size_t lcount(const Round* r, size_t n) const { const size_t lo = (size_t)sh_rank * r->S; return n <= lo ? 0 : (n - lo < r->S ? n - lo : r->S); }
  // chunk of a distributed round polynomial on a domain of D points (0: the round stays replicated): the commitment must be sharded and the chunk even
  size_t dist_chunk(size_t D) const {
    if (!sh_on || !shard_dist || !fri_blowup || !shardable(D / 2)) return 0;
    const size_t den = fri_blowup * (size_t)sh_world;
    if (D % den) return 0;
    const size_t S = D / den;
    return (S >= 2 && !(S & 1)) ? S : 0;
  }
  DevBuf d_carry, d_lq, d_pack, d_fullpoly;
  std::vector<Round*> rounds; size_t nrounds_done = 0, fri_rounds = 0, fri_blowup = 0;
  bool have_deep = false; XE cur_z; XE cur_B[2];
  DevBuf d_folded, d_sh, d_blob, d_tabs, d_targets, d_idx, d_deg, d_ovf;
  size_t blob_size = 0;

  int ensure_polys(size_t count) {
    if (count <= polys_cap) return 0;
    size_t ncap = polys_cap ? polys_cap * 2 : 8;
    while (ncap < count) ncap *= 2;
    DevBuf nb;
    if (nb.ensure(ncap * N * sizeof(T))) return fail(MS_ERR_NOMEM, "polys");
    if (d_polys.p && npolys > 0) { CK(msrt::d2d(nb.p, d_polys.p, (size_t)npolys * N * sizeof(T), stream)); CK(msrt::sync(stream)); }
    d_polys.release();
    d_polys = nb; polys_cap = ncap;
    return 0;
  }

  int init(int dev, u32 flags) {
    device = dev; zae = (flags & MS_FLAG_ZERO_DISPLAY_EMPTY) ? 1 : 0; trace_mont = (flags & MS_FLAG_TRACE_MONT64) ? 1 : 0;
    if (const char* e = getenv("MS_NTT_KMAX")) { int v = atoi(e); if (v >= 5 && v <= msntt::MAX_LOG_R) ntt_kmax = v; }
    if (const char* e = getenv("MS_NTT_MAXPAD")) { int v = atoi(e); if (v >= 0 && v <= msntt::MAX_LOG_PAD) ntt_maxpad = v; }
    if (const char* e = getenv("MS_NTT_MAXRHO")) { int v = atoi(e); if (v >= 0 && v <= msntt::MAX_LOG_RHO) ntt_maxrho = v; }
    if (const char* e = getenv("MS_NTT_TH512")) ntt_th512 = atoi(e);
    if (const char* e = getenv("MS_NTT_FAST")) ntt_fast = atoi(e);
    if (const char* e = getenv("MS_NTT_V2")) ntt_v2 = atoi(e);
    if (const char* e = getenv("MS_NTT_V2_SUB3")) ntt_v2_sub3 = atoi(e);
    if (const char* e = getenv("MS_NTT_V2_REGPASS")) ntt_v2_regpass = atoi(e);
    if (const char* e = getenv("MS_NTT_SHARE")) ntt_share = atoi(e);
    if (const char* e = getenv("MS_NTT_COOP_WGS")) { int v = atoi(e); if (v >= 8 && v <= 65536) ntt_coop_wgs = v & ~7; }
    if (const char* e = getenv("MS_NTT_V2_MAXPASS")) { int v = atoi(e); if (v >= 1 && v <= 4) ntt_v2_maxpass = v; }
    if (const char* e = getenv("MS_NTT_COLBATCH")) ntt_colbatch = atoi(e);
    if (const char* e = getenv("MS_NTT_MALL_MIB")) { int v = atoi(e); if (v >= 1) ntt_mall_mib = v; }
    if (const char* e = getenv("MS_NTT_V2_MIN")) { int v = atoi(e); if (v >= 12 && v <= 32) ntt_v2_min = v; }
    if (const char* e = getenv("MS_LDE_LINEAR")) lde_linear = atoi(e);
    if (const char* e = getenv("MS_LAZY_LINCOMB")) lazy_lin = atoi(e);
    if (const char* e = getenv("MS_LDE_MULTI")) lde_multi = atoi(e);
    if (const char* e = getenv("MS_LDE_VIRTUAL")) lde_virtual = atoi(e);
    if (const char* e = getenv("MS_FRI_POINTWISE")) fri_pointwise = atoi(e);
    if (const char* e = getenv("MS_FOLD_SMALL_MAX")) fold_small_max = (size_t)atol(e);
    if (const char* e = getenv("MS_EVAL_SMALL_MAX")) eval_small_max = (size_t)atol(e);
    if (const char* e = getenv("MS_TREE_SUBTREE_PARENTS")) subtree_parents = (size_t)atol(e);
    // the boundary's bulk copies (r04): page-locked trace in / FRI proof out on SDMA engines through the HSA runtime by default (measured with 8 provers in flight,
    // tools/io_probe3.py: resident 251 proofs/s; upload by hipMemcpyAsync 238, by SDMA 251; read-back by hipMemcpyAsync 217-223, by SDMA 242-248; both by SDMA 245.5 = 0.978)
    if (const char* e = getenv("MS_UPLOAD")) upload_sdma = !strcmp(e, "hip") ? 0 : 1;
    if (const char* e = getenv("MS_READBACK")) readback_sdma = !strcmp(e, "hip") ? 0 : (!strcmp(e, "sdma-async") ? 1 : 2);   // sdma-async: only ms_fri_proof_read_async on the engine
    if (const char* e = getenv("MS_LEAF_LAZY_MIN")) leaf_lazy_min = atoi(e);
    if (const char* e = getenv("MS_TREE_TOP")) { int v = atoi(e); if (v >= 1 && v <= 65536) tree_top_parents = v; }
    if (const char* e = getenv("MS_NTT_FAST_MIN")) ntt_fast_min = atoi(e);
    if (const char* e = getenv("MS_NTT_FAST_MAX")) ntt_fast_max = atoi(e);
    if (const char* e = getenv("MS_SHARD_MIN_LEAVES")) { long v = atol(e); if (v >= 1) shard_min_leaves = (size_t)v; }
    if (const char* e = getenv("MS_SHARD_DIST")) shard_dist = atoi(e);
    if (const char* e = getenv("MS_SHARD_WORLD1")) allow_w1 = atoi(e);
    if (const char* e = getenv("MS_SHARD_GATHER_CHUNK")) { long v = atol(e); if (v >= 64) shard_gather_chunk = (size_t)v & ~(size_t)63; }
    if (const char* e = getenv("MS_SHARD_SLICES")) { int v = atoi(e); if (v >= 1 && v <= 64) { shard_slices = v; shard_slices_set = true; } }
    if (const char* e = getenv("MS_SHARD_SLICE_MIN")) { long v = atol(e); if (v >= 1) shard_slice_min = (size_t)v; }
    if (const char* e = getenv("MS_RCCL_MAX_PIECE")) { long long v = atoll(e); if (v >= 64 && v <= ((long long)1 << 30)) rccl_max_piece = (size_t)v & ~(size_t)63; }
    CK(msrt::set_device(dev));
    CK(msrt::stream_create(&own_stream));
    stream = own_stream;
    pinned_cap = 1 << 16;
    CK(msrt::malloc_host(&pinned, pinned_cap));
    if (d_small.ensure(4096)) return fail(MS_ERR_NOMEM, "small");
    return 0;
  }
  ~Ctx() {
    for (auto& kv : plans) { kv.second->tw_lo.release(); kv.second->tw_hi.release(); kv.second->vtw.release(); kv.second->w0.release(); for (auto& b : kv.second->w_r) b.release(); delete kv.second; }
    drop_rccl();
    if (sdma_pending) msrt::Sdma::get().wait(sdma_sig, 20.0);
    else if (copy_pending) msrt::event_sync(ev_copy);
    if (sdma_state == 1) { msrt::Sdma::get().signal_destroy(sdma_sig); msrt::Sdma::get().signal_destroy(sdma_up_sig); }
    for (msrt::Event* e : ev_hash) msrt::event_destroy(e);
    for (msrt::Event* e : ev_xchg) msrt::event_destroy(e);
    if (comm_stream) msrt::stream_destroy(comm_stream);
    if (ev_blob) msrt::event_destroy(ev_blob);
    if (ev_copy) msrt::event_destroy(ev_copy);
    if (copy_stream) msrt::stream_destroy(copy_stream);
    for (Round* r : rounds) { r->poly.release(); r->cw.release(); r->nodes.release(); delete r; }
    DevBuf* bufs[] = {&ntt_scratch, &d_trace, &d_polys, &d_coef, &d_lde, &d_trace_nodes, &d_lde_nodes, &d_io, &d_partials, &d_small, &d_folded, &d_sh, &d_blob, &d_tabs, &d_targets, &d_idx, &d_deg, &d_ovf, &d_zero, &d_lin, &d_cubic, &d_carry, &d_lq, &d_pack, &d_fullpoly};
    for (DevBuf* b : bufs) b->release();
    if (pinned) msrt::free_host(pinned);
    if (h_tabs) msrt::free_host(h_tabs);
    if (own_stream) msrt::stream_destroy(own_stream);
  }
  int ext_degree() const override { return E; }
  void bind_device() const override { msrt::set_device(device); }
  int set_stream(void* s) override { stream = s ? reinterpret_cast<msrt::Stream*>(s) : own_stream; zero_used = zero_cap; return 0; }
  int synchronize() override { CK(msrt::sync(stream)); return 0; }

  static unsigned grid1(size_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

  // staged copies between the u64 ABI and device storage
  int upload_narrow(const u64* host, size_t n, T* dst) {
    if (d_io.ensure(n * 8)) return fail(MS_ERR_NOMEM, "io staging");
    CK(msrt::h2d(d_io.p, host, n * 8, stream));
    typename mspoly::NarrowKernel<F>::Params p{d_io.as<u64>(), dst, n};
    CK(run<mspoly::NarrowKernel<F>>(K_IO, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
    return 0;
  }
  int download_widen(const T* src, size_t n, size_t limb_stride, u32 e, u64* host) {
    if (n == 0) return 0;
    if (d_io.ensure(n * e * 8)) return fail(MS_ERR_NOMEM, "io staging");
    typename mspoly::WidenKernel<F>::Params p{src, d_io.as<u64>(), n, limb_stride, e};
    CK(run<mspoly::WidenKernel<F>>(K_IO, grid1(n * e, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
    CK(msrt::d2h(host, d_io.p, n * e * 8, stream));
    CK(msrt::sync(stream));
    return 0;
  }
  static bool canonical(const u64* v, size_t n) { for (size_t i = 0; i < n; i++) if (v[i] >= F::P) return false; return true; }

  // ------------------------------------------------------------------ starks.rs:68-73
  int trace_commit(const u64* trace, bool on_device, size_t N_, size_t w_, size_t lpn, u8* root) override {
    if (!trace || !root) return fail(MS_ERR_ARG, "null argument");
    if (!N_ || !w_ || !is_pow2(N_)) return fail(MS_ERR_SHAPE, "trace length must be a power of two (air.rs:23)");
    if (ctz64(N_) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "trace domain larger than the field's two-adicity (air.rs:74)");
    TreeShape ts;
    RQ(tree_shape(N_ * w_, lpn, 2, &ts));
    // (the canonical range of the trace is checked by the transposing kernel for host and device input alike: a host-side scan of the N x w matrix
    //  cost ~3 ms of the proving thread per 2^20-row proof, with its stream idle - r04, I/O leg)
    have_trace = have_polys = have_lde = have_validity = false; npolys = 0; nrounds_done = 0; have_deep = false; blob_size = 0;
    if (N_ != N) { polys_cap = 0; d_polys.release(); }
    N = N_; w = w_;
    const u64* dsrc;
    if (on_device) dsrc = trace;
    else {
      if (d_trace.ensure(N * w * 8)) return fail(MS_ERR_NOMEM, "trace");
      bool sent = false;
      if (upload_sdma && sdma_ready() && msrt::is_pinned_host(trace)) {   // MS_UPLOAD=sdma: the trace on an SDMA engine through the HSA runtime (page-locked sources only); the host waits for it
        msrt::Sdma& S = msrt::Sdma::get();
        CK(msrt::sync(stream));                                           // d_trace may still be read by the previous proof's transposing kernel
        if (S.h2d_engine(sdma_gpu) && !S.copy_h2d(sdma_gpu, d_trace.p, trace, N * w * 8, sdma_up_sig, S.h2d_engine(sdma_gpu))) {
          if (S.wait(sdma_up_sig, 20.0)) return fail(MS_ERR_HIP, "SDMA upload did not complete within 20 s");
          sent = true;
        }
      }
      if (!sent) CK(msrt::h2d(d_trace.p, trace, N * w * 8, stream));
      dsrc = d_trace.as<u64>();
    }
    RQ(ensure_polys(w + 1));
    // 2^-64 mod p: arkworks stores Montgomery representatives (R = 2^64 for the one-limb Fp of both fields)
    const T rinv = f_inv<F>(F::from_u64((u64)(((unsigned __int128)1 << 64) % F::P)));
    void* badw;
    RQ(zero_alloc(4, &badw));  // device input cannot be range-checked on the host: the kernel flags elements >= p
    typename mspoly::TransposeInKernel<F>::Params tp{dsrc, d_polys.as<T>(), N, w, N, rinv, trace_mont, reinterpret_cast<u32*>(badw)};
    if (w >= 16 && N >= 64) {   // wide traces: 64 x 64 tiles through LDS (coalesced both ways)
      typedef mspoly::TransposeInTiledKernel<F> TK;
      CK(run<TK>(K_TRANSPOSE, (unsigned)(N / TK::TILE), (unsigned)((w + TK::TILE - 1) / TK::TILE), TK::THREADS, TK::lds_bytes(), tp));
    } else
    CK(run<mspoly::TransposeInKernel<F>>(K_TRANSPOSE, grid1(N * w, mspoly::THREADS), 1, mspoly::THREADS, 0, tp));
    // element f of trace.get_data() = column f % w, row f / w of the column-major copy
    // one proof over several ranks: every rank holds the whole trace, so rank k hashes the contiguous leaf groups [k*M/W, (k+1)*M/W) and only the W subtree roots travel (r04)
    if (sh_on && shard_dist && shardable(ts.leaf_num / ts.lpn)) RQ((tree_build_sharded_contiguous<1>(d_polys.as<T>(), N, 1, 0, (u32)w, ts, d_trace_nodes)));
    else
    RQ((tree_build<1>(d_polys.as<T>(), N, 1, 0, (u32)w, ts, d_trace_nodes)));
    trace_ts = ts;
    CK(msrt::d2h(reinterpret_cast<u8*>(pinned) + 128, badw, 4, stream));
    RQ(read_root(d_trace_nodes, ts, root));
    if (*reinterpret_cast<const u32*>(reinterpret_cast<const u8*>(pinned) + 128)) return fail(MS_ERR_ARG, "trace element not canonical (>= p)");
    have_trace = true;
    return MS_OK;
  }
  // ------------------------------------------------------------------ air.rs:147-160
  int interpolate() override {
    if (!have_trace) return fail(MS_ERR_STATE, "interpolate before trace_commit");
    RQ(ntt_run(ctz64(N), true, d_polys.as<T>(), N, N, d_polys.as<T>(), N, w));
    poly_lin.assign(w, Lin());   // (first: if this allocation fails the session keeps no half-set state)
    poly_mat.assign(w, 1);
    npolys = (int)w; have_polys = true; have_lde = have_validity = false;
    return MS_OK;
  }
  int polys_lincomb(const u64* s, const int* idx, int k) override {
    if (!have_polys) return fail(MS_ERR_STATE, "lincomb before interpolate");
    if (!s || !idx || k < 1) return fail(MS_ERR_ARG, "bad lincomb arguments");
    for (int t = 0; t < k; t++) if (idx[t] < 0 || idx[t] >= npolys || s[t] >= F::P) return fail(MS_ERR_ARG, "lincomb index/scalar out of range");
    RQ(ensure_polys(npolys + 2));
    if (!lazy_lin) RQ(lincomb_into(d_polys.as<T>(), N, N, s, idx, k, npolys, d_polys.as<T>() + (size_t)npolys * N));
    poly_mat.push_back(lazy_lin ? 0 : 1);
    { Lin l; l.s.assign(s, s + k); l.idx.assign(idx, idx + k); poly_lin.push_back(l); }
    npolys++; have_lde = have_validity = false;
    return MS_OK;
  }
  int polys_append(const u64* coeffs, size_t n) override {
    if (!have_polys) return fail(MS_ERR_STATE, "append before interpolate");
    if (!coeffs || n > N) return fail(MS_ERR_SHAPE, "constraint polynomial has more than N coefficients (starks.rs:118-119 asserts)");
    if (!canonical(coeffs, n)) return fail(MS_ERR_ARG, "coefficient not canonical");
    RQ(ensure_polys(npolys + 2));
    T* dst = d_polys.as<T>() + (size_t)npolys * N;
    CK(msrt::memset_dev(dst, 0, N * sizeof(T), stream));
    if (n) { RQ(upload_narrow(coeffs, n, dst)); CK(msrt::sync(stream)); }   // the caller's buffer is only read during the call (include/ministark.h)
    poly_mat.push_back(1);
    poly_lin.push_back(Lin());
    npolys++; have_lde = have_validity = false;
    return MS_OK;
  }
  int polys_count() const override { return npolys; }
  int poly_read(int i, u64* out) override {
    if (!have_polys || i < 0 || i >= npolys || !out) return fail(MS_ERR_ARG, "poly_read");
    RQ(materialize(i));
    return download_widen(d_polys.as<T>() + (size_t)i * N, N, 0, 1, out);
  }

  // ------------------------------------------------------------------ starks.rs:80-95
  // one lincomb launch chain: dst = sum_t s[t] * base[idx[t]] over n elements (columns `stride` apart)
  int lincomb_into(const T* base, size_t stride, size_t n, const u64* sc, const int* idx, int k, int self_index, T* dst) {
    for (int t = 0; t < k;) {  // MAX_TERMS terms per launch; launches after the first spend one slot on the partial result
      typename mspoly::LincombKernel<F>::Params p;
      p.polys = base; p.stride = stride; p.n = n; p.dst = dst;
      int kk = 0;
      if (t > 0) { p.s[kk] = F::from_u64(1); p.idx[kk] = self_index; kk++; }  // accumulate onto the partial result
      for (; t < k && kk < mspoly::MAX_TERMS; t++, kk++) { p.s[kk] = F::from_u64(sc[t]); p.idx[kk] = idx[t]; }
      p.k = kk;
      CK(run<mspoly::LincombKernel<F>>(K_LINCOMB, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
    }
    return 0;
  }
  // The LDE columns of all polynomials with linear provenance (columns `stride` apart, n elements each).  Consecutive ones whose sources
  // are all transformed columns (no linear column among them) and fit LincombMultiKernel (<= 4 outputs over <= 8 distinct sources) share
  // one sweep; anything else goes through lincomb_into one by one.
  int lincomb_linear_columns(T* base, size_t stride, size_t n) {
    const size_t c = (size_t)npolys;
    typedef mspoly::LincombMultiKernel<F> MK;
    typename MK::Params mp; int nout = 0, nsrc = 0;
    auto flush = [&]() -> int {
      if (!nout) return 0;
      mp.polys = base; mp.stride = stride; mp.n = n; mp.nout = nout; mp.nsrc = nsrc;
      next_bytes = (double)(nout + nsrc) * n * sizeof(T);
      int e = run<MK>(K_LINCOMB, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, mp);
      nout = nsrc = 0;
      return e;
    };
    for (size_t i = 0; i < c; i++) {
      const Lin& li = poly_lin[i];
      if (li.idx.empty()) continue;
      bool simple = (lde_multi < 0 ? c >= 16 : lde_multi != 0) && li.idx.size() <= (size_t)mspoly::LCM_SRC;
      for (int ix : li.idx) if (ix < 0 || (size_t)ix >= c || !poly_lin[ix].idx.empty()) simple = false;   // a source that is itself a linear column: keep the order
      if (simple) {
        for (int attempt = 0; attempt < 2; attempt++) {
          int map[mspoly::LCM_SRC], ns = nsrc; bool fits = nout < mspoly::LCM_OUT;
          int srcs[mspoly::LCM_SRC]; for (int u = 0; u < nsrc; u++) srcs[u] = mp.src[u];
          for (size_t t = 0; fits && t < li.idx.size(); t++) {
            int u = 0; while (u < ns && srcs[u] != li.idx[t]) u++;
            if (u == ns) { if (ns == mspoly::LCM_SRC) { fits = false; break; } srcs[ns++] = li.idx[t]; }
            map[t] = u;
          }
          if (!fits) { CK(flush()); continue; }   // second attempt on an empty group always fits (<= LCM_SRC terms)
          for (int u = nsrc; u < ns; u++) mp.src[u] = srcs[u];
          for (int u = 0; u < mspoly::LCM_SRC; u++) mp.m[nout][u] = 0;
          for (size_t t = 0; t < li.idx.size(); t++) mp.m[nout][map[t]] = F::add(mp.m[nout][map[t]], F::from_u64(li.s[t] % F::P));   // a column named twice: coefficients add up
          mp.dst[nout] = base + i * stride;
          nsrc = ns; nout++;
          break;
        }
      } else {
        CK(flush());
        RQ(lincomb_into(base, stride, n, li.s.data(), li.idx.data(), (int)li.idx.size(), (int)i, base + i * stride));
      }
    }
    CK(flush());
    return 0;
  }
  // starks.rs:80-91.  The coset evaluation is linear, so a polynomial that ms_polys_lincomb defined as
  // sum_t s_t * P_idx[t] has LDE column sum_t s_t * LDE(P_idx[t]): only polynomials without such provenance
  // (the trace columns, ms_polys_append uploads) go through the NTT.
  int lde_compute(size_t blowup_, u64 shift) {
    const size_t c = (size_t)npolys;
    const size_t L_ = N * blowup_;
    if (d_coef.ensure(c * N * sizeof(T)) || d_lde.ensure(c * L_ * sizeof(T))) return fail(MS_ERR_NOMEM, "lde");
    if (!lde_linear) for (size_t i = 0; i < c; i++) RQ(materialize((int)i));
    const T sh = F::from_u64(shift), sh_step = f_pow<F>(sh, msntt::ScalePowKernel<F>::THREADS);
    const int per_block = msntt::ScalePowKernel<F>::THREADS * msntt::ScalePowKernel<F>::ITEMS;
    for (size_t i = 0; i < c;) {  // maximal runs of polynomials that need a transform
      if (lde_linear && !poly_lin[i].idx.empty()) { i++; continue; }
      size_t j = i;
      while (j < c && !(lde_linear && !poly_lin[j].idx.empty())) j++;
      typename msntt::ScalePowKernel<F>::Params sp;
      sp.src = d_polys.as<T>() + i * N; sp.dst = d_coef.as<T>() + i * N; sp.src_bstride = N; sp.dst_bstride = N; sp.n = N; sp.s = sh; sp.s_step = sh_step;
      CK(run<msntt::ScalePowKernel<F>>(K_SCALE_POW, grid1(N, per_block), (unsigned)(j - i), msntt::ScalePowKernel<F>::THREADS, 0, sp));
      RQ(ntt_run(ctz64(L_), false, d_coef.as<T>() + i * N, N, N, d_lde.as<T>() + i * L_, L_, j - i));
      i = j;
    }
    if (lde_linear) RQ(finish_linear_columns(L_, L_));
    return 0;
  }
  // The linear LDE columns: VIRTUAL when every one of them is a combination of at most LIN_MAXT stored (transformed) columns - the leaf-hash kernel
  // then evaluates them row by row while it hashes (they are never read again: the query phase opens FRI codewords only) and ms_lde_read materialises
  // them on demand; otherwise written out by the lincomb kernels.  MS_LDE_VIRTUAL: -1 (default) for AIRs of >= 16 polynomials, 0 never, 1 always.
  // Measured r03: wide AIR (64 linear columns of 2^25 rows) LDE commit 105.5 -> 103.6 ms - the 20 ms of lincomb launches go away, but the leaf kernel reads
  // its source columns again (out of L2 by then) and multiplies; Fibonacci AIR with 8 proofs in flight 239 -> 233 proofs/s (the prover is bound by its VALU
  // instruction count, and the leaf kernel's grew): so narrow AIRs keep the lincomb kernels.
  int lde_virtual = -1; bool lde_cols_virtual = false; size_t lde_col_stride = 0, lde_col_len = 0;
  DevBuf d_lin;
  int finish_linear_columns(size_t stride, size_t n) {
    const size_t c = (size_t)npolys;
    lde_cols_virtual = false; lde_col_stride = stride; lde_col_len = n;
    bool any = false, simple = lde_virtual < 0 ? c >= 16 : lde_virtual != 0;
    for (size_t i = 0; i < c; i++) {
      const Lin& li = poly_lin[i];
      if (li.idx.empty()) continue;
      any = true;
      if (li.idx.size() > (size_t)msmerkle::LIN_MAXT) simple = false;
      for (int ix : li.idx) if (ix < 0 || (size_t)ix >= c || !poly_lin[ix].idx.empty()) simple = false;
    }
    if (!any) return 0;
    if (!simple) return lincomb_linear_columns(d_lde.as<T>(), stride, n);
    std::vector<msmerkle::LinColSpec> spec(c);
    for (size_t i = 0; i < c; i++) {
      memset(&spec[i], 0, sizeof spec[i]);
      const Lin& li = poly_lin[i];
      spec[i].n = (u32)li.idx.size();
      for (size_t t = 0; t < li.idx.size(); t++) { spec[i].src[t] = (u32)li.idx[t]; spec[i].s[t] = li.s[t] % F::P; }
    }
    if (d_lin.ensure(c * sizeof(msmerkle::LinColSpec))) return fail(MS_ERR_NOMEM, "virtual column table");
    if (c * sizeof(msmerkle::LinColSpec) > pinned_cap - 8192) return lincomb_linear_columns(d_lde.as<T>(), stride, n);
    // staged through page-locked memory behind the small results: the previous proof's copy out of it completed before that proof's lde_commit returned
    memcpy(reinterpret_cast<u8*>(pinned) + 8192, spec.data(), c * sizeof(msmerkle::LinColSpec));
    CK(msrt::h2d(d_lin.p, reinterpret_cast<u8*>(pinned) + 8192, c * sizeof(msmerkle::LinColSpec), stream));
    lde_cols_virtual = true;
    return 0;
  }
  const msmerkle::LinColSpec* lde_lin() const { return lde_cols_virtual ? d_lin.as<msmerkle::LinColSpec>() : nullptr; }
  // Evaluations of `batch` polynomials (ncoef coefficients each) on this rank's share of the size-2^log_D domain shift*<w_D>:
  // the rows g*(rank + W*i) + t (t < g, i < m = D/(g*W)) — g cosets of <w_m> — land at dst[b*dst_bstride + t*m + i].
  int coset_eval(const T* coef, size_t coef_bstride, size_t ncoef, int log_D, T shift, size_t g, T* dst, size_t dst_bstride, size_t batch) {
    if (batch == 0) return 0;
    const size_t D = (size_t)1 << log_D, W = (size_t)sh_world, m = D / (g * W);
    if (m == 0) return fail(MS_ERR_STATE, "domain too small to shard");
    if (d_coef.ensure(batch * m * sizeof(T))) return fail(MS_ERR_NOMEM, "coset scratch");
    const T wD = f_root_of_unity<F>(log_D);
    typedef msntt::CosetFoldKernel<F> CF;
    const size_t lim = ncoef < m ? ncoef : m;
    for (size_t t = 0; t < g; t++) {
      const T zeta = F::mul(shift, f_pow<F>(wD, (u64)(g * (size_t)sh_rank + t)));
      typename CF::Params cp;
      cp.src = coef; cp.dst = d_coef.as<T>(); cp.src_bstride = coef_bstride; cp.dst_bstride = m; cp.n = ncoef; cp.m = m;
      cp.s = zeta; cp.s_step = f_pow<F>(zeta, CF::THREADS); cp.sm = f_pow<F>(zeta, (u64)m);
      CK(run<CF>(K_SCALE_POW, grid1(lim, CF::THREADS * CF::ITEMS), (unsigned)batch, CF::THREADS, 0, cp));
      RQ(ntt_run(ctz64(m), false, d_coef.as<T>(), m, lim, dst + t * m, dst_bstride, batch));
    }
    return 0;
  }
  // lde_compute for a sharded proof: column i of the local LDE (rows rank + W*j) at d_lde + i*m, m = L/W
  int lde_compute_sharded(size_t blowup_, u64 shift) {
    const size_t c = (size_t)npolys, L_ = N * blowup_, m = L_ / (size_t)sh_world;
    if (d_lde.ensure(c * m * sizeof(T))) return fail(MS_ERR_NOMEM, "lde");
    if (!lde_linear) for (size_t i = 0; i < c; i++) RQ(materialize((int)i));
    PartScope part(this);
    for (size_t i = 0; i < c;) {
      if (lde_linear && !poly_lin[i].idx.empty()) { i++; continue; }
      size_t j = i;
      while (j < c && !(lde_linear && !poly_lin[j].idx.empty())) j++;
      RQ(coset_eval(d_polys.as<T>() + i * N, N, N, ctz64(L_), F::from_u64(shift), 1, d_lde.as<T>() + i * m, m, j - i));
      i = j;
    }
    if (lde_linear) RQ(finish_linear_columns(m, m));
    return 0;
  }
  int lde_commit(size_t blowup_, u64 shift, size_t lpn, u8* root) override {
    if (!have_polys) return fail(MS_ERR_STATE, "lde_commit before interpolate");
    if (!root || !blowup_ || !is_pow2(blowup_) || shift == 0 || shift >= F::P) return fail(MS_ERR_ARG, "bad blowup/shift");
    const size_t L_ = N * blowup_;
    if (ctz64(L_) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "LDE domain larger than the field's two-adicity (starks.rs:82-83)");
    const size_t c = (size_t)npolys;
    TreeShape ts;
    RQ(tree_shape(L_ * c, lpn, 2, &ts));
    L = L_; blowup = blowup_; lde_c = c; lde_shift = shift;
    if (lpn == c && shardable(L_)) {  // one LDE row per leaf group: rank k evaluates and hashes the rows k (mod world)
      RQ(lde_compute_sharded(blowup_, shift));
      RQ((tree_build_sharded<1>(d_lde.as<T>(), L / (size_t)sh_world, 1, 0, (u32)c, ts, d_lde_nodes, lde_lin())));
    } else {
      RQ(lde_compute(blowup_, shift));
      RQ((tree_build<1>(d_lde.as<T>(), L, 1, 0, (u32)c, ts, d_lde_nodes, lde_lin())));
    }
    lde_ts = ts;
    RQ(read_root(d_lde_nodes, ts, root));
    have_lde = true;
    return MS_OK;
  }
  int bench_lde(size_t blowup_, u64 shift) override {
    if (!have_polys) return fail(MS_ERR_STATE, "bench_lde before interpolate");
    have_lde = false;                              // d_lde is overwritten: whatever ms_lde_commit left there is gone
    const int keep = lde_virtual; lde_virtual = 0; // time the columns as WRITTEN (a virtual column's work would move into a leaf kernel this entry does not run)
    const int rc = lde_compute(blowup_, shift);
    lde_virtual = keep; lde_cols_virtual = false;
    return rc;
  }
  int arith_selftest(int op, const u64* a, const u64* b, u64* out, size_t n) override {
    if (!a || !b || !out || op < 0 || op > 7 || (F::ID != 0 && op > 3)) return fail(MS_ERR_ARG, "arith_selftest: bad operation / null argument");
    if (!n) return MS_OK;
    if (!canonical(a, n) || (op < 6 && !canonical(b, n))) return fail(MS_ERR_ARG, "operand not canonical");
    DevBuf d;
    if (d.ensure(3 * n * 8)) return fail(MS_ERR_NOMEM, "arith_selftest");
    typedef typename std::conditional<F::ID == 0, GLM, F>::type A;
    typedef mspoly::ArithKernel<F, A> AK;
    int e = msrt::h2d(d.p, a, n * 8, stream);
    if (!e) e = msrt::h2d(d.as<u64>() + n, b, n * 8, stream);
    if (!e) e = msrt::sync(stream);
    typename AK::Params ap{d.as<u64>(), d.as<u64>() + n, d.as<u64>() + 2 * n, n, op};
    if (!e) e = run<AK>(K_IO, grid1(n, AK::THREADS), 1, AK::THREADS, 0, ap);
    if (!e) e = msrt::d2h(out, d.as<u64>() + 2 * n, n * 8, stream);
    if (!e) e = msrt::sync(stream);
    d.release();
    return e ? fail_rt(e, "arith_selftest") : MS_OK;
  }
  int lde_read(u64* out) override {
    if (!have_lde || !out) return fail(MS_ERR_STATE, "lde_read");
    if (lde_ts.sharded) return fail(MS_ERR_STATE, "lde_read: the LDE of a sharded proof is distributed over the ranks");
    if (lde_cols_virtual) { RQ(lincomb_linear_columns(d_lde.as<T>(), lde_col_stride, lde_col_len)); lde_cols_virtual = false; }   // the virtual columns, written out on demand
    const size_t tot = L * lde_c;
    if (d_io.ensure(tot * 8)) return fail(MS_ERR_NOMEM, "io");
    typename mspoly::TransposeOutKernel<F>::Params p{d_lde.as<T>(), d_io.as<u64>(), L, lde_c, L};
    CK(run<mspoly::TransposeOutKernel<F>>(K_IO, grid1(tot, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
    CK(msrt::d2h(out, d_io.p, tot * 8, stream));
    CK(msrt::sync(stream));
    return MS_OK;
  }
  // ------------------------------------------------------------------ starks.rs:108-119
  int mix(u64 r) override {
    if (!have_polys) return fail(MS_ERR_STATE, "mix before interpolate");
    if (r >= F::P) return fail(MS_ERR_ARG, "r not canonical");
    RQ(ensure_polys(npolys + 1));
    bool any_lazy = false;
    for (int i = 0; i < npolys; i++) any_lazy = any_lazy || !poly_mat[i];
    if (any_lazy) {   // sum_i r^i f_i as ONE combination of the polynomials without provenance (the lazily defined f_i are never formed)
      std::map<int, T> acc;
      T rp = F::from_u64(1);
      for (int i = 0; i < npolys; i++) { expand(i, rp, acc); rp = F::mul(rp, F::from_u64(r)); }
      std::vector<u64> sc; std::vector<int> ix;
      for (auto& kv : acc) if (kv.second != 0) { sc.push_back(F::to_u64(kv.second)); ix.push_back(kv.first); }
      T* dst = d_polys.as<T>() + (size_t)npolys * N;
      if (sc.empty()) CK(msrt::memset_dev(dst, 0, N * sizeof(T), stream));
      else RQ(lincomb_into(d_polys.as<T>(), N, N, sc.data(), ix.data(), (int)sc.size(), npolys, dst));
    } else {
    typename mspoly::MixKernel<F>::Params p{d_polys.as<T>(), N, N, npolys, F::from_u64(r), d_polys.as<T>() + (size_t)npolys * N};
    CK(run<mspoly::MixKernel<F>>(K_MIX, grid1(N, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
    }
    have_validity = true; validity_len = N; nrounds_done = 0;
    return MS_OK;
  }
  // build-defined degree-3 composition with the true quotient (include/ministark.h; kernel: mspoly::CubicComposeKernel)
  size_t validity_len = 0; u64 lde_shift = 0; DevBuf d_cubic;
  size_t validity_len_() const override { return have_validity ? validity_len : 0; }
  int mix_cubic(u64 r, const int* spec, const u64* sc, int ncons) override {
    if (!have_lde) return fail(MS_ERR_STATE, "mix_cubic before lde_commit");
    if (lde_ts.sharded) return fail(MS_ERR_STATE, "mix_cubic: the LDE of a sharded proof is distributed over the ranks");
    if (!spec || !sc || ncons < 1 || ncons > 4096 || r >= F::P) return fail(MS_ERR_ARG, "mix_cubic arguments");
    if (blowup < 4) return fail(MS_ERR_SHAPE, "mix_cubic needs blowup >= 4 (the quotient has 2N coefficients, the composition 3N)");
    const size_t c = lde_c;
    for (int t = 0; t < ncons; t++) { for (int u = 0; u < 5; u++) if (spec[5 * t + u] < 0 || (size_t)spec[5 * t + u] >= c) return fail(MS_ERR_ARG, "mix_cubic: polynomial index out of range"); if (sc[t] >= F::P) return fail(MS_ERR_ARG, "mix_cubic: scalar not canonical"); }
    if (lde_cols_virtual) { RQ(lincomb_linear_columns(d_lde.as<T>(), lde_col_stride, lde_col_len)); lde_cols_virtual = false; }
    typedef mspoly::CubicSpec<F> CS;
    typedef mspoly::CubicComposeKernel<F> CK_;
    const int logL = ctz64(L), logN = ctz64(N);
    const T gL = f_root_of_unity<F>(logL), wN = f_root_of_unity<F>(logN), sh = F::from_u64(lde_shift);
    // x^N - 1 on the coset: shift^N * zeta^(i mod blowup) - 1, zeta = g_L^N
    const T shN = f_pow<F>(sh, (u64)N), zeta = f_pow<F>(gL, (u64)N);
    std::vector<T> dinv(blowup);
    { T z = F::from_u64(1);
      for (size_t k = 0; k < blowup; k++) { const T den = F::sub(F::mul(shN, z), F::from_u64(1)); if (den == 0) return fail(MS_ERR_SHAPE, "mix_cubic: the LDE coset meets the trace domain (shift^N is a blowup-th root of unity)"); dinv[k] = f_inv<F>(den); z = F::mul(z, zeta); } }
    std::vector<CS> hs(ncons);
    { T rp = F::from_u64(1);
      for (int t = 0; t < ncons; t++) { hs[t].j = (u32)spec[5 * t]; hs[t].a = (u32)spec[5 * t + 1]; hs[t].b = (u32)spec[5 * t + 2]; hs[t].c = (u32)spec[5 * t + 3]; hs[t].d = (u32)spec[5 * t + 4]; hs[t].s = F::from_u64(sc[t]); hs[t].rpow = rp; rp = F::mul(rp, F::from_u64(r)); } }
    const size_t tab_bytes = hs.size() * sizeof(CS) + dinv.size() * sizeof(T);
    if (d_cubic.ensure(2 * L * sizeof(T)) || d_tabs.ensure(tab_bytes + 64)) return fail(MS_ERR_NOMEM, "mix_cubic buffers");
    u8* ht;
    RQ(tabs_host(tab_bytes + 64, &ht));     // (the previous user of the area, the last proof's query phase, ended with a stream synchronisation)
    memcpy(ht, hs.data(), hs.size() * sizeof(CS));
    memcpy(ht + hs.size() * sizeof(CS), dinv.data(), dinv.size() * sizeof(T));
    CK(msrt::h2d(d_tabs.p, ht, tab_bytes, stream));
    typename CK_::Params cp;
    cp.lde = d_lde.as<T>(); cp.L = L; cp.blowup = (u32)blowup; cp.ncons = (u32)ncons; cp.spec = d_tabs.as<CS>();
    cp.den_inv = reinterpret_cast<const T*>(d_tabs.as<u8>() + hs.size() * sizeof(CS));
    cp.shift = sh; cp.gL = gL; cp.gL_step = f_pow<F>(gL, (u64)CK_::THREADS); cp.w_last = f_pow<F>(wN, (u64)(N - 1)); cp.out = d_cubic.as<T>();
    CK(run<CK_>(K_MIX, grid1(L, CK_::THREADS * CK_::ITEMS), 1, CK_::THREADS, 0, cp));
    // evaluations on shift * <g_L>  ->  coefficients of Q(shift y)  ->  q_k = coefficient_k * shift^-k
    T* coef = d_cubic.as<T>() + L;
    RQ(ntt_run(logL, true, d_cubic.as<T>(), L, L, coef, L, 1));
    RQ(ensure_polys(npolys + 2));
    // exactness: nothing above 2N coefficients (the reference's `assert_eq!(rest, zero)` of starks.rs:119, for the true quotient)
    unsigned long long* dres;
    RQ(degree_launch1(coef, L, &dres));
    CK(msrt::d2h(pinned, dres, 8, stream));
    CK(msrt::sync(stream));
    if (*reinterpret_cast<unsigned long long*>(pinned) > 2 * N) return fail(MS_ERR_SHAPE, "mix_cubic: the constraints do not vanish on the trace domain (the quotient by x^N - 1 is not a polynomial of 2N coefficients)");
    const T shi = f_inv<F>(sh);
    typename msntt::ScalePowKernel<F>::Params sp;
    sp.src = coef; sp.dst = d_polys.as<T>() + (size_t)npolys * N; sp.src_bstride = 0; sp.dst_bstride = 0; sp.n = 2 * N; sp.s = shi; sp.s_step = f_pow<F>(shi, msntt::ScalePowKernel<F>::THREADS);
    CK(run<msntt::ScalePowKernel<F>>(K_SCALE_POW, grid1(2 * N, msntt::ScalePowKernel<F>::THREADS * msntt::ScalePowKernel<F>::ITEMS), 1, msntt::ScalePowKernel<F>::THREADS, 0, sp));
    have_validity = true; validity_len = 2 * N; nrounds_done = 0;
    return MS_OK;
  }
  int degree_launch1(const T* poly, size_t n, unsigned long long** dres_out) {   // trimmed length of a base-field coefficient vector
    void* zr;
    RQ(zero_alloc(8, &zr));
    typename mspoly::DegreeKernel<F, 1>::Params dp{poly, 0, n, reinterpret_cast<unsigned long long*>(zr), 0};
    CK(run<mspoly::DegreeKernel<F, 1>>(K_DEGREE, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
    *dres_out = reinterpret_cast<unsigned long long*>(zr);
    return 0;
  }
  int validity_read(u64* out) override {
    if (!have_validity || !out) return fail(MS_ERR_STATE, "validity_read");
    return download_widen(d_polys.as<T>() + (size_t)npolys * N, validity_len, 0, 1, out);
  }

  // evaluate `npoly` polynomials (views) at ext point z into dst as [npoly][E] T
  template <int EC>
  int eval_views(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst) {
    size_t maxc = 0;
    for (int i = 0; i < npoly; i++) if (count[i] > maxc) maxc = count[i];
    return maxc <= eval_small_max ? eval_views_i<EC, 4>(base, poly_stride, limb_stride, kstride, off, count, npoly, z, dst, maxc)
                                  : eval_views_i<EC, 16>(base, poly_stride, limb_stride, kstride, off, count, npoly, z, dst, maxc);
  }
  template <int EC, int ITEMS>
  int eval_views_i(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst, size_t maxc) {
    typedef mspoly::EvalKernel<F, EC, E, ITEMS> EK;
    const size_t chunk = (size_t)EK::THREADS * EK::ITEMS;
    const size_t nblocks = maxc ? (maxc + chunk - 1) / chunk : 1;
    if (nblocks > 1 && d_partials.ensure(nblocks * npoly * E * sizeof(T))) return fail(MS_ERR_NOMEM, "partials");
    typename EK::Params p;
    p.base = base; p.poly_stride = poly_stride; p.limb_stride = limb_stride; p.kstride = kstride; p.npoly = npoly;
    for (int i = 0; i < mspoly::MAX_POLYS; i++) { p.off[i] = i < npoly ? off[i] : 0; p.count[i] = i < npoly ? count[i] : 0; }
    XE sq = z;
    for (int i = 0; i < 9; i++) { p.zpow2[i] = sq; sq = e_mul<F>(sq, sq); }
    p.partials = nblocks > 1 ? d_partials.as<T>() : dst;  // single block: P_0 is the value
    CK(run_coop<EK>(K_EVAL, (unsigned)nblocks, EK::THREADS, EK::lds_bytes(), p));
    if (nblocks > 1) {
      typedef mspoly::ReducePartialsKernel<F, E> RK;
      typename RK::Params rp;
      rp.partials = d_partials.as<T>(); rp.nblocks = nblocks; rp.per_thread = (nblocks + RK::THREADS - 1) / RK::THREADS; rp.npoly = npoly; rp.out = dst;
      XE zc = p.zpow2[8];  // z^256
      for (int i = 256; i < (int)chunk; i *= 2) zc = e_mul<F>(zc, zc);  // z^CH
      XE zs = zc;
      for (int i = 0; i < 8; i++) { rp.zs2[i] = zs; zs = e_mul<F>(zs, zs); }   // zc^(2^i)
      rp.zc_step = zs;                                                          // zc^256 = zc^THREADS
      static_assert(RK::THREADS == 256, "zc_step = zc^THREADS");
      CK(run_coop<RK>(K_EVAL_REDUCE, 1, RK::THREADS, RK::lds_bytes(), rp));
    }
    return 0;
  }
  static bool load_ext(const u64* v, XE* out) { for (int l = 0; l < E; l++) { if (v[l] >= F::P) return false; out->c[l] = F::from_u64(v[l]); } return true; }

  // sharded proof: out[i] = sum_r part_r[i] * zstep^r for n extension elements of the all-gathered partials (rank r's payload rank_stride limbs apart, the elements from `off` on)
  int shard_combine_launch(size_t off, size_t rank_stride, u32 n, const XE& zstep, T* out) {
    typedef mspoly::ShardCombineKernel<F, E> CKn;
    typename CKn::Params cp{reinterpret_cast<const T*>(xr) + off, rank_stride, n, (u32)sh_world, zstep, out};
    CK(run<CKn>(K_EVAL_REDUCE, grid1(n, CKn::THREADS), 1, CKn::THREADS, 0, cp));
    return 0;
  }
  // DEEP-ALI evaluations by coefficient range (r04): rank k evaluates the coefficients [k*Sx, (k+1)*Sx) of every polynomial at every point, ONE all-gather of the partial
  // sums, and every rank combines them with z^Sx
  bool dist_eval() const { return sh_on && shard_dist && N >= shard_min_leaves && N >= (size_t)sh_world * 2; }
  // the polynomials the DEEP-ALI kernels evaluate: those whose coefficients exist, then the validity polynomial (index npolys); the lazily defined ones get their
  // values on the host as the combination their provenance names (eval_finish)
  std::vector<int> eval_set() const { std::vector<int> ev; for (int i = 0; i < npolys; i++) if (poly_mat[i]) ev.push_back(i); ev.push_back(npolys); return ev; }
  // page-locked results [q][ev.size()][E] -> out [q][npolys + 1][E] (u64)
  int eval_finish(int q, const std::vector<int>& ev, u64* out) {
    const size_t nev = ev.size(), np = (size_t)npolys + 1;
    const T* h = reinterpret_cast<const T*>(pinned);
    std::vector<int> slot(np, -1);
    for (size_t k = 0; k < nev; k++) slot[ev[k]] = (int)k;
    std::vector<std::map<int, T>> exp_(np);
    for (int i = 0; i < npolys; i++) if (slot[i] < 0) expand(i, F::from_u64(1), exp_[i]);
    for (int t = 0; t < q; t++) {
      const T* ht = h + (size_t)t * nev * E;
      for (size_t i = 0; i < np; i++) {
        u64* o = out + ((size_t)t * np + i) * E;
        if (slot[i] >= 0) { for (int l = 0; l < E; l++) o[l] = F::to_u64(ht[(size_t)slot[i] * E + l]); continue; }
        XE acc = e_zero<F, E>();
        for (auto& kv : exp_[i]) { XE v; for (int l = 0; l < E; l++) v.c[l] = ht[(size_t)slot[kv.first] * E + l]; acc = e_add<F, E>(acc, e_mul_base<F, E>(v, kv.second)); }
        for (int l = 0; l < E; l++) o[l] = F::to_u64(acc.c[l]);
      }
    }
    return MS_OK;
  }
  int eval_ext_sharded(const u64* z, int q, u64* out) {
    const std::vector<int> ev = eval_set();
    const int nev = (int)ev.size();
    const size_t tot = (size_t)q * nev * E, W = (size_t)sh_world;
    const size_t maxlen = validity_len > N ? validity_len : N, Sx = maxlen / W, lo = (size_t)sh_rank * Sx;
    if (tot * sizeof(T) > pinned_cap) return fail(MS_ERR_ARG, "too many evaluation points");
    if (tot * sizeof(T) * W > xcap) return fail(MS_ERR_NOMEM, "exchange buffers too small for the DEEP-ALI partial sums");
    std::vector<XE> zs((size_t)q);
    for (int t = 0; t < q; t++) {
      if (!load_ext(z + (size_t)t * E, &zs[t])) return fail(MS_ERR_ARG, "query point not canonical");
      for (int i0 = 0; i0 < nev; i0 += mspoly::MAX_POLYS) {
        const int nb = (nev - i0 < mspoly::MAX_POLYS) ? nev - i0 : mspoly::MAX_POLYS;
        size_t off[mspoly::MAX_POLYS], cnt[mspoly::MAX_POLYS];
        for (int i = 0; i < nb; i++) { const int pi = ev[i0 + i]; const size_t len = (pi == npolys) ? validity_len : N; off[i] = (size_t)pi * N + lo; cnt[i] = len <= lo ? 0 : (len - lo < Sx ? len - lo : Sx); }
        { PartScope part(this); RQ((eval_views<1>(d_polys.as<T>(), 0, 0, 1, off, cnt, nb, zs[t], reinterpret_cast<T*>(xs) + ((size_t)t * nev + i0) * E))); }
      }
    }
    if (tot) {
      RQ(exchange(MS_XCHG_ALL_GATHER, tot * sizeof(T)));
      for (int t = 0; t < q; t++) RQ(shard_combine_launch((size_t)t * nev * E, tot, (u32)nev, e_pow<F, E>(zs[t], (u64)Sx), reinterpret_cast<T*>(pinned) + (size_t)t * nev * E));
      CK(msrt::sync(stream));
      RQ(eval_finish(q, ev, out));
    }
    return MS_OK;
  }

  // ------------------------------------------------------------------ starks.rs:124-151
  int eval_ext(const u64* z, int q, u64* out) override {
    if (!have_validity) return fail(MS_ERR_STATE, "eval_ext before mix");
    if (!z || !out || q < 0) return fail(MS_ERR_ARG, "eval_ext");
    if (dist_eval()) return eval_ext_sharded(z, q, out);
    const std::vector<int> ev = eval_set();
    const int nev = (int)ev.size();
    const size_t tot = (size_t)q * nev * E;
    if (d_small.ensure(tot * sizeof(T) + 4096)) return fail(MS_ERR_NOMEM, "small");
    if (tot * sizeof(T) > pinned_cap) return fail(MS_ERR_ARG, "too many evaluation points");
    for (int t = 0; t < q; t++) {
      XE zz;
      if (!load_ext(z + (size_t)t * E, &zz)) return fail(MS_ERR_ARG, "query point not canonical");
      for (int i0 = 0; i0 < nev; i0 += mspoly::MAX_POLYS) {
        const int nb = (nev - i0 < mspoly::MAX_POLYS) ? nev - i0 : mspoly::MAX_POLYS;
        size_t off[mspoly::MAX_POLYS], cnt[mspoly::MAX_POLYS];
        for (int i = 0; i < nb; i++) { const int pi = ev[i0 + i]; off[i] = (size_t)pi * N; cnt[i] = (pi == npolys) ? validity_len : N; }   // the validity polynomial has 2N coefficients after ms_mix_cubic
        RQ((eval_views<1>(d_polys.as<T>(), 0, 0, 1, off, cnt, nb, zz, reinterpret_cast<T*>(pinned) + ((size_t)t * nev + i0) * E)));   // results land in page-locked host memory
      }
    }
    if (tot) {
      CK(msrt::sync(stream));
      RQ(eval_finish(q, ev, out));
    }
    return MS_OK;
  }

  // ------------------------------------------------------------------ FRI rounds (fri.rs:314-352)
  Round* round_slot(size_t i) { while (rounds.size() <= i) rounds.push_back(new Round()); return rounds[i]; }
  // codeword + tree of rounds[i] from its coefficient limbs (ncoef_in valid coefficients)
  // `nonzero_limbs`: limbs >= this are identically zero (round 0: extend_poly embeds base coefficients), so their
  // transform is all zeros and is not computed
  // `prev` != nullptr: the codeword is folded out of prev's codeword in the evaluation domain (FriFoldEvalKernel) instead of
  // transforming the round polynomial — same values, a quarter of the arithmetic
  int round_commit(Round* r, size_t ncoef_in, int nonzero_limbs = E, const Round* prev = nullptr, const XE* alpha = nullptr) {
    if (ctz64(r->D) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "FRI domain larger than the field's two-adicity");
    RQ(tree_shape(r->D, 2, 2, &r->ts));  // starks.rs:290-295: leafs_per_node 2, inner_children 2
    r->m = 0;
    {
      const bool shard_next = shardable(r->D / 2);
      bool z_outside_base = false;   // then y - z != 0 on the whole (base-field) domain
      for (int l = 1; l < E; l++) z_outside_base = z_outside_base || cur_z.c[l] != 0;
      if (prev && fri_pointwise && z_outside_base && prev->D == 2 * r->D && prev->ts.sharded == shard_next) {
        Plan* pl;
        RQ(get_plan(ctz64(prev->D), 0, false, &pl));   // w_D^e tables of the previous domain
        // one output per thread in the late rounds (at most MS_FOLD_SMALL_MAX outputs: their launches are latency, not throughput), eight otherwise
        const size_t W = (size_t)sh_world;
        const size_t m_out = shard_next ? r->D / (2 * W) : r->D;
        const size_t local = shard_next ? 2 * m_out : r->D;       // elements per limb held here
        if (shard_next) r->m = m_out;
        if (r->cw.ensure(local * E * sizeof(T))) return fail(MS_ERR_NOMEM, "codeword");
        const size_t total = m_out * (shard_next ? 2 : 1);
        const XE c = e_add<F, E>(cur_B[0], e_mul<F>(cur_B[1], *alpha));   // B(alpha), fri.rs:99
        auto fold_launch = [&](auto* kernel) {
          typedef typename std::remove_pointer<decltype(kernel)>::type FK;
          typename FK::Params fp;
          fp.src = prev->cw.template as<T>(); fp.src_limb_stride = shard_next ? 2 * prev->m : prev->D;
          fp.dst = r->cw.template as<T>(); fp.dst_limb_stride = local;
          fp.m_out = m_out; fp.log_m = (u32)ctz64(m_out); fp.groups = shard_next ? 2 : 1; fp.shard_W = shard_next ? (u32)W : 0; fp.shard_k = (u32)sh_rank;
          fp.tw_lo = pl->tw_lo.template as<T>(); fp.tw_hi = pl->tw_hi.template as<T>(); fp.lo_bits = (u32)pl->lo_bits; fp.log_D = (u32)ctz64(prev->D);
          fp.alpha = *alpha;
          fp.c2 = e_add<F, E>(c, c);
          fp.z = cur_z;
          fp.inv2 = f_inv<F>(F::from_u64(2));
          return run<FK>(K_FOLD, grid1(total, FK::THREADS * FK::ITEMS), 1, FK::THREADS, 0, fp);
        };
        next_bytes = (double)total * E * sizeof(T) * 3;   // two inputs read, one output written per element
        if (shard_next) part_depth++;
        const int e_ = total <= fold_small_max ? fold_launch((mspoly::FriFoldEvalKernel<F, E, 1>*)nullptr) : fold_launch((mspoly::FriFoldEvalKernel<F, E, 8>*)nullptr);
        if (shard_next) part_depth--;
        CK(e_);
        if (shard_next) RQ((tree_build_sharded<E>(r->cw.template as<T>(), m_out, 1, 2 * m_out, 2, r->ts, r->nodes)));
        else RQ((tree_build<E>(r->cw.template as<T>(), 0, 1, r->D, 1, r->ts, r->nodes)));
        return 0;
      }
    }
    if (shardable(r->D / 2)) {  // leaf group j = codeword elements 2j, 2j+1: rank k owns the groups k (mod world) = two cosets of size m
      const size_t m = r->D / (2 * (size_t)sh_world);
      r->m = m;
      if (r->cw.ensure(2 * m * E * sizeof(T))) return fail(MS_ERR_NOMEM, "codeword");
      const T* coef = r->poly.template as<T>(); size_t coef_stride = r->cap;
      if (r->local_store) {   // the transform needs every coefficient (a DEEP point in the base field / MS_FRI_POINTWISE=0): gather the parts of the distributed polynomial
        const size_t full = r->S * (size_t)sh_world;
        if (d_fullpoly.ensure(full * E * sizeof(T))) return fail(MS_ERR_NOMEM, "gathered polynomial");
        if (ncoef_in) RQ(gather_poly(lpoly(r), r->S, ncoef_in, d_fullpoly.as<T>(), full));
        coef = d_fullpoly.as<T>(); coef_stride = full;
      }
      { PartScope part(this); RQ(coset_eval(coef, coef_stride, ncoef_in, ctz64(r->D), F::from_u64(1), 2, r->cw.template as<T>(), 2 * m, (size_t)nonzero_limbs)); }
      if (nonzero_limbs < E) CK(msrt::memset_dev(r->cw.template as<T>() + (size_t)nonzero_limbs * 2 * m, 0, (size_t)(E - nonzero_limbs) * 2 * m * sizeof(T), stream));
      RQ((tree_build_sharded<E>(r->cw.template as<T>(), m, 1, 2 * m, 2, r->ts, r->nodes)));
      return 0;
    }
    if (r->cw.ensure(r->D * E * sizeof(T))) return fail(MS_ERR_NOMEM, "codeword");
    RQ(ntt_run(ctz64(r->D), false, r->poly.template as<T>(), r->cap, ncoef_in, r->cw.template as<T>(), r->D, (size_t)nonzero_limbs));  // fri.rs:350
    if (nonzero_limbs < E) CK(msrt::memset_dev(r->cw.template as<T>() + (size_t)nonzero_limbs * r->D, 0, (size_t)(E - nonzero_limbs) * r->D * sizeof(T), stream));
    RQ((tree_build<E>(r->cw.template as<T>(), 0, 1, r->D, 1, r->ts, r->nodes)));                                       // fri.rs:351
    return 0;
  }
  // trimmed length of a round polynomial (DegreeKernel) into a zeroed device word
  int degree_launch(const T* poly, size_t limb_stride, size_t n, unsigned long long** dres_out) {
    void* zr;
    RQ(zero_alloc(8, &zr));
    unsigned long long* dres = reinterpret_cast<unsigned long long*>(zr);
    if (n) {
      typename mspoly::DegreeKernel<F, E>::Params dp{poly, limb_stride, n, dres, 0};
      CK(run<mspoly::DegreeKernel<F, E>>(K_DEGREE, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
    }
    *dres_out = dres;
    return 0;
  }
  int read_degree_and_root(const T* poly, size_t limb_stride, size_t n, Round* r, size_t* ncoef, u8* root) {
    unsigned long long* dres = nullptr;
    RQ(degree_launch(poly, limb_stride, n, &dres));
    const bool on_host = r && root_on_host;   // r's tree is the one built last (round_commit just before)
    CK(msrt::d2h(pinned, dres, 8, stream));
    if (r && !on_host) CK(msrt::d2h(reinterpret_cast<u8*>(pinned) + 64, r->nodes.template as<u8>() + (r->ts.local_nodes - 1) * 32, 32, stream));
    CK(msrt::sync(stream));
    *ncoef = (size_t)(*reinterpret_cast<unsigned long long*>(pinned));
    if (r && root) memcpy(root, on_host ? reinterpret_cast<const u8*>(host_root()) : reinterpret_cast<const u8*>(pinned) + 64, 32);
    return 0;
  }
  // fri.rs:73-82
  int fri_begin(size_t blowup_, size_t nrounds, u8* root0) override {
    if (!have_validity) return fail(MS_ERR_STATE, "fri_begin before mix");
    if (!root0 || nrounds < 1 || !blowup_) return fail(MS_ERR_ARG, "fri_begin");
    nrounds_done = 0; have_deep = false; blob_size = 0;
    fri_rounds = nrounds; fri_blowup = blowup_;
    if (d_deg.p) CK(msrt::memset_dev(d_deg.p, 0, 256, stream));   // the self-clearing degree word of fri_fold_commit: zero again even if an earlier proof was abandoned mid-round
    Round* r = round_slot(0);
    const size_t VL = validity_len;   // N (ms_mix) or 2N (ms_mix_cubic)
    r->cap = VL;
    if (r->poly.ensure(VL * E * sizeof(T))) return fail(MS_ERR_NOMEM, "round poly");
    // field.rs:23-32 extend_poly: limb 0 = validity, higher limbs zero
    CK(msrt::memset_dev(r->poly.p, 0, VL * E * sizeof(T), stream));
    CK(msrt::d2d(r->poly.p, d_polys.as<T>() + (size_t)npolys * N, VL * sizeof(T), stream));
    size_t nc;
    RQ(read_degree_and_root(r->poly.template as<T>(), r->cap, VL, nullptr, &nc, nullptr));
    r->ncoef = nc;
    const size_t deg = nc ? nc - 1 : 0;
    size_t dsize = (deg + 1) * blowup_;  // fri.rs:74 (quirk Q11)
    size_t D = 1; while (D < dsize) D <<= 1;
    r->D = D;
    r->S = dist_chunk(D); r->dist = r->S != 0; r->local_store = false;   // the validity polynomial is replicated: a distributed round 0 means every rank WORKS on its range of it
    RQ(round_commit(r, nc, 1));
    RQ(read_root(r->nodes, r->ts, root0));
    nrounds_done = 1;
    return MS_OK;
  }
  // fri.rs:89-94
  int fri_deep(const u64* z, u64* B) override {
    if (nrounds_done == 0 || nrounds_done >= fri_rounds) return fail(MS_ERR_STATE, "fri_deep out of order");
    if (!z || !B) return fail(MS_ERR_ARG, "fri_deep");
    if (!load_ext(z, &cur_z)) return fail(MS_ERR_ARG, "z not canonical");
    Round* r = rounds[nrounds_done - 1];
    size_t off[2] = {0, 1}, cnt[2] = {(r->ncoef + 1) / 2, r->ncoef / 2};
    T* dst = reinterpret_cast<T*>(pinned);   // the last kernel of the evaluation stores its 2 E words straight into page-locked host memory
    if (r->dist) {   // even(z), odd(z) by coefficient range: partial sums over this rank's coefficients, one all-gather, combination with z^(S/2)
      const size_t lc = lcount(r, r->ncoef);
      cnt[0] = (lc + 1) / 2; cnt[1] = lc / 2;   // (the rank's first coefficient has an even index: S is even)
      { PartScope part(this); RQ((eval_views<E>(lpoly(r), 0, lstride(r), 2, off, cnt, 2, cur_z, reinterpret_cast<T*>(xs)))); }
      RQ(exchange(MS_XCHG_ALL_GATHER, 2 * E * sizeof(T)));
      RQ(shard_combine_launch(0, 2 * E, 2, e_pow<F, E>(cur_z, (u64)(r->S / 2)), dst));
    } else
    RQ((eval_views<E>(r->poly.template as<T>(), 0, r->cap, 2, off, cnt, 2, cur_z, dst)));  // fri.rs:354-359
    CK(msrt::sync(stream));
    const T* h = reinterpret_cast<const T*>(pinned);
    for (int s = 0; s < 2; s++) for (int l = 0; l < E; l++) { cur_B[s].c[l] = h[s * E + l]; B[s * E + l] = F::to_u64(h[s * E + l]); }
    have_deep = true;
    return MS_OK;
  }

  // ---- suffix Horner job planning (shared by the DEEP quotient and the query quotients).
  // A logical job (view, m, z, out, h0) expands into one kernel job per level; level buffers
  // (aggregates = input of the level above, carries = output of the level above) come from d_sh.
  typedef mspoly::SHJob<F, E> SHJ;
  typedef mspoly::SuffixHornerKernel<F, E> SHK;
  struct SHPlan { int nl; std::vector<SHJ> agg; std::vector<SHJ> fin; bool has_top_agg = false; SHJ top_agg; size_t P = 0; bool dist = false; };  // agg[l] for l < nl-1, fin[l] for l < nl
  // (P = BS^nl: where the kernel places a carry-in of the top level, in elements of the job)
  static size_t sh_scratch_elems(size_t m) {
    const size_t BS = mspoly::SH_BS;
    size_t tot = 0, cur = m;
    for (;;) { size_t nb = cur ? (cur + BS - 1) / BS : 1; tot += 2 * nb * E; if (nb <= 1) break; cur = nb; }
    return tot;
  }
  // `scratch` must hold sh_scratch_elems(m) elements of T
  // ext_carry (E limbs, device): carry-in of the top level (a rank of a sharded proof: the suffix sum over the higher ranks, scaled - ShardCarryKernel);
  // top_agg (E limbs, device): the job's aggregate over all its elements, stored by one extra AGG launch of the top level; out_h0: see SHJob
  SHPlan sh_plan(const T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const XE& z,
                 void* out, bool out_u64, size_t out_limb_stride, size_t out_off, size_t out_stride, T* h0, T* scratch,
                 const T* ext_carry = nullptr, T* top_agg = nullptr, bool out_h0 = false) {
    const size_t BS = mspoly::SH_BS;
    std::vector<size_t> ms; ms.push_back(m);
    while ((ms.back() ? (ms.back() + BS - 1) / BS : 1) > 1) ms.push_back((ms.back() + BS - 1) / BS);
    const int nl = (int)ms.size();
    std::vector<size_t> aoff(nl), coff(nl), nbs(nl);
    size_t tot = 0;
    for (int l = 0; l < nl; l++) { nbs[l] = ms[l] ? (ms[l] + BS - 1) / BS : 1; aoff[l] = tot; tot += nbs[l] * E; coff[l] = tot; tot += nbs[l] * E; }
    SHPlan pl; pl.nl = nl; pl.agg.resize(nl > 1 ? nl - 1 : 0); pl.fin.resize(nl);
    XE zl = z;
    for (int l = 0; l < nl; l++) {
      SHJ j;
      memset(&j, 0, sizeof j);
      j.z = zl;
      XE sq = e_pow<F, E>(zl, mspoly::SH_SEG);
      for (int i = 0; i < 9; i++) { j.zpow[i] = sq; sq = e_mul<F>(sq, sq); }
      zl = j.zpow[8];  // z^(SEG*256) = z^BS: multiplier of the level above
      j.m = ms[l];
      if (l == 0) { j.in = in; j.in_limb_stride = in_limb_stride; j.in_off = in_off; j.in_stride = in_stride; }
      else { j.in = scratch + aoff[l - 1]; j.in_limb_stride = nbs[l - 1]; j.in_off = 0; j.in_stride = 1; }
      if (l + 1 < nl) {  // aggregate job feeding level l+1
        SHJ a = j; a.agg = scratch + aoff[l]; a.agg_limb_stride = nbs[l];
        pl.agg[l] = a;
        j.carry = scratch + coff[l]; j.carry_limb_stride = nbs[l];
      } else {
        if (top_agg) { SHJ a = j; a.agg = top_agg; a.agg_limb_stride = 1; pl.top_agg = a; pl.has_top_agg = true; }
        if (ext_carry) { j.carry = ext_carry; j.carry_limb_stride = 1; }
      }
      if (l == 0) { j.out = out; j.out_u64 = out_u64 ? 1 : 0; j.out_limb_stride = out_limb_stride; j.out_off = out_off; j.out_stride = out_stride; j.h0 = h0; j.out_h0 = out_h0 ? 1 : 0; }
      else { j.out = scratch + coff[l - 1]; j.out_u64 = 0; j.out_limb_stride = nbs[l - 1]; j.out_off = 0; j.out_stride = 1; j.tail_zero = ext_carry ? 2 : 1; }
      pl.fin[l] = j;
    }
    pl.P = 1; for (int l = 0; l < nl; l++) pl.P *= BS;
    return pl;
  }
  // z^(m - P) = (1/z)^(P - m): moves a top-level carry-in from the padded position P to the job's end m (0 for z = 0: nothing then carries over)
  static XE carry_scale(const XE& z, size_t m, size_t P) { return e_pow<F, E>(e_inv<F>(z), (u64)(P - m)); }
  int sh_launch_inline(const SHJ& j, int final_mode) {
    typename SHK::Params p; p.jobs = nullptr; p.inline_job = j; p.final_mode = final_mode;
    const size_t nb = j.m ? (j.m + mspoly::SH_BS - 1) / mspoly::SH_BS : 1;
    CK(run_coop<SHK>(K_SUFFIX_HORNER, (unsigned)nb, SHK::THREADS, SHK::lds_bytes(), p));
    return 0;
  }
  // one logical job, launched level by level with the job inline in the kernel arguments
  int suffix_horner(const T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const XE& z,
                    T* out, size_t out_limb_stride, size_t out_off, size_t out_stride, T* h0) {
    if (d_sh.ensure(sh_scratch_elems(m) * sizeof(T))) return fail(MS_ERR_NOMEM, "scan levels");
    SHPlan pl = sh_plan(in, in_limb_stride, in_off, in_stride, m, z, out, false, out_limb_stride, out_off, out_stride, h0, d_sh.as<T>());
    for (int l = 0; l + 1 < pl.nl; l++) RQ(sh_launch_inline(pl.agg[l], 0));
    for (int l = pl.nl - 1; l >= 0; l--) RQ(sh_launch_inline(pl.fin[l], 1));
    return 0;
  }

  // all-gather of a distributed round polynomial's parts into one replicated vector (dst: E limbs, dst_stride apart, `count` coefficients)
  int gather_poly(const T* local, size_t S, size_t count, T* dst, size_t dst_stride) {
    const size_t bytes = S * E * sizeof(T);
    if (bytes * (size_t)sh_world > xcap) return fail(MS_ERR_NOMEM, "exchange buffers too small to gather a round polynomial");
    CK(msrt::d2d(xs, local, bytes, stream));
    RQ(exchange(MS_XCHG_ALL_GATHER, bytes));
    typedef mspoly::GatherPolyKernel<F, E> GK;
    typename GK::Params gp{reinterpret_cast<const T*>(xr), S * E, S, count, dst, dst_stride};
    CK(run<GK>(K_IO, grid1(count * E, GK::THREADS), 1, GK::THREADS, 0, gp));
    return 0;
  }
  // fri.rs:96-101 on a DISTRIBUTED round polynomial (r04): rank k folds its own coefficient pairs, runs the suffix Horner of (folded - B(alpha)) / (x - z) over its
  // own range with the sum over the higher ranks as carry-in (one all-gather of [first folded element | aggregate] per rank, ShardCarryKernel), and ends up with
  // its range [k*S', (k+1)*S') of the quotient = the next round polynomial, S' = S/2.  If the next round is too small to stay distributed the parts are
  // all-gathered into a replicated polynomial.  H_j for j in (lo, hi) comes from the rank's own job over f[lo+1 .. hi); H_hi = q_(hi-1) IS the carry-in.
  int fold_dist(Round* pr, Round* nr, const XE& a, size_t* nq_coef_out) {
    const size_t n = pr->ncoef, m = (n + 1) / 2, Sn = pr->S / 2;
    const bool next_dist = dist_chunk(nr->D) == Sn;
    const size_t cntp = lcount(pr, n), cnt = (cntp + 1) / 2;   // this rank's coefficients of the round polynomial / of the folded polynomial
    T* lq;
    if (next_dist) { if (nr->poly.ensure(Sn * E * sizeof(T))) return fail(MS_ERR_NOMEM, "fold"); lq = nr->poly.template as<T>(); }
    else { if (d_lq.ensure(Sn * E * sizeof(T))) return fail(MS_ERR_NOMEM, "fold"); lq = d_lq.as<T>(); }
    if (d_folded.ensure((Sn + 1) * E * sizeof(T)) || d_carry.ensure(4096)) return fail(MS_ERR_NOMEM, "fold");
    if (4 * E * sizeof(T) * (size_t)sh_world > xcap) return fail(MS_ERR_NOMEM, "exchange buffers");
    size_t nq_coef = 0;
    PartScope part(this);
    if (m >= 2) {
      typename mspoly::FoldKernel<F, E>::Params fp{lpoly(pr), lstride(pr), cntp, d_folded.as<T>(), Sn, a};  // fri.rs:361-372
      CK(run<mspoly::FoldKernel<F, E>>(K_FOLD, grid1(cnt, mspoly::THREADS), 1, mspoly::THREADS, 0, fp));
      T* pay = reinterpret_cast<T*>(xs);        // [first folded element (E limbs) | aggregate of the job (E limbs)]
      CK(msrt::memset_dev(xs, 0, 2 * E * sizeof(T), stream));
      if (cnt) {
        typename mspoly::CopyLimbsKernel<F>::Params cl{d_folded.as<T>(), Sn, pay, 1, (u32)E};
        CK(run<mspoly::CopyLimbsKernel<F>>(K_IO, 1, 1, mspoly::CopyLimbsKernel<F>::THREADS, 0, cl));
      }
      const size_t mj = cnt ? cnt - 1 : 0;
      SHPlan pl; pl.nl = 0;
      if (mj) {
        if (d_sh.ensure(sh_scratch_elems(mj) * sizeof(T))) return fail(MS_ERR_NOMEM, "scan levels");
        pl = sh_plan(d_folded.as<T>(), Sn, 1, 1, mj, cur_z, lq, false, Sn, 1, 1, nullptr, d_sh.as<T>(), d_carry.as<T>(), pay + E, true);
        for (int l = 0; l + 1 < pl.nl; l++) RQ(sh_launch_inline(pl.agg[l], 0));
        RQ(sh_launch_inline(pl.top_agg, 0));
      }
      RQ(exchange(MS_XCHG_ALL_GATHER, 2 * E * sizeof(T)));
      typedef mspoly::ShardCarryKernel<F, E> CKn;
      typename CKn::Params cp;
      memset(&cp, 0, sizeof cp);
      cp.jobs = nullptr; cp.njobs = 1; cp.W = (u32)sh_world; cp.rank = (u32)sh_rank; cp.gathered = reinterpret_cast<const T*>(xr); cp.rank_stride = 2 * E;
      cp.inline_job.first_off = 0; cp.inline_job.agg_off = E; cp.inline_job.has_first = 1;
      cp.inline_job.zA = cur_z; cp.inline_job.zB = e_pow<F, E>(cur_z, (u64)(Sn - 1));
      cp.inline_job.scale = mj ? carry_scale(cur_z, mj, pl.P) : e_one<F, E>();
      cp.inline_job.carry_out = d_carry.as<T>();
      cp.inline_job.tail_out = cnt ? lq + (cnt - 1) : nullptr; cp.inline_job.tail_stride = Sn;   // q_(hi-1) = H_hi (zero, and beyond the polynomial, on the rank that holds its top)
      cp.inline_job.h0_out = nullptr;
      CK(run<CKn>(K_SUFFIX_HORNER, 1, 1, CKn::THREADS, 0, cp));
      if (mj) for (int l = pl.nl - 1; l >= 0; l--) RQ(sh_launch_inline(pl.fin[l], 1));
      nq_coef = m - 1;
    }
    nr->S = Sn;
    if (next_dist) { nr->dist = true; nr->local_store = true; }
    else {   // the round after this one is small: replicate the quotient (one all-gather of S' coefficients per rank) and go on as an unsharded prover would
      if (nr->poly.ensure(nr->cap * E * sizeof(T))) return fail(MS_ERR_NOMEM, "fold");
      if (nq_coef) RQ(gather_poly(lq, Sn, nq_coef, nr->poly.template as<T>(), nr->cap));
      nr->S = 0;
    }
    *nq_coef_out = nq_coef;
    return 0;
  }
  // fri.rs:96-109
  int fri_fold_commit(const u64* alpha, u8* root) override {
    if (!have_deep) return fail(MS_ERR_STATE, "fri_fold_commit before fri_deep");
    XE a;
    if (!alpha || !root || !load_ext(alpha, &a)) return fail(MS_ERR_ARG, "alpha");
    Round* pr = rounds[nrounds_done - 1];
    if (pr->D < 4) return fail(MS_ERR_SHAPE, "FRI round domain too small to fold (merkle.rs:93-104 panics)");
    const size_t n = pr->ncoef, m = (n + 1) / 2;
    Round* nr = round_slot(nrounds_done);
    nr->cap = m ? m : 1;
    nr->D = pr->D / 2;  // fri.rs:104, 374-376
    nr->dist = false; nr->local_store = false; nr->S = 0;
    size_t nq_coef = 0;
    if (pr->dist) RQ(fold_dist(pr, nr, a, &nq_coef));
    else {
    if (nr->poly.ensure(nr->cap * E * sizeof(T)) || d_folded.ensure((m + 1) * E * sizeof(T))) return fail(MS_ERR_NOMEM, "fold");
    if (m >= 2) {
      typename mspoly::FoldKernel<F, E>::Params fp{pr->poly.template as<T>(), pr->cap, n, d_folded.as<T>(), m, a};  // fri.rs:361-372
      CK(run<mspoly::FoldKernel<F, E>>(K_FOLD, grid1(m, mspoly::THREADS), 1, mspoly::THREADS, 0, fp));
      // (folded - B(alpha)) / (x - z): quotient coefficients are H_1.. of the suffix Horner in z (fri.rs:99-101)
      RQ(suffix_horner(d_folded.as<T>(), m, 0, 1, m, cur_z, nr->poly.template as<T>(), nr->cap, 0, 1, nullptr));
      nq_coef = m - 1;
    }
    }
    // the degree scan runs BEFORE the commitment and the tree's final launch forwards its 8-byte result, with the root, into page-locked host memory:
    // no copy launch in front of the round's one stream synchronisation (r03; the trace showed a 4 us copyBuffer kernel + its launch gap per round)
    // (the word lives in d_deg, zeroed once: the forwarding thread clears it again - the pool of zero_alloc may be wiped while the tree is being built)
    if (!d_deg.p) { if (d_deg.ensure(256)) return fail(MS_ERR_NOMEM, "degree word"); CK(msrt::memset_dev(d_deg.p, 0, 256, stream)); }
    unsigned long long* dres = d_deg.as<unsigned long long>();
    if (nr->dist) {   // this rank's part reports the GLOBAL trimmed length; the maximum over the ranks comes back with the subtree roots (finish_sharded_tree)
      const size_t lc = lcount(nr, nq_coef);
      if (lc) {
        typename mspoly::DegreeKernel<F, E>::Params dp{lpoly(nr), lstride(nr), lc, dres, (size_t)sh_rank * nr->S};
        PartScope part(this);
        CK(run<mspoly::DegreeKernel<F, E>>(K_DEGREE, grid1(lc, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
      }
      shard_aux = dres;
    } else if (nq_coef) {
      typename mspoly::DegreeKernel<F, E>::Params dp{nr->poly.template as<T>(), nr->cap, nq_coef, dres, 0};
      CK(run<mspoly::DegreeKernel<F, E>>(K_DEGREE, grid1(nq_coef, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
    }
    pending_aux = dres; aux_on_host = false;
    RQ(round_commit(nr, nq_coef, E, pr, &a));
    pending_aux = nullptr; shard_aux = nullptr;
    if (!aux_on_host) { CK(msrt::d2h(pinned, dres, 8, stream)); CK(msrt::memset_dev(dres, 0, 8, stream)); }
    if (!root_on_host) CK(msrt::d2h(reinterpret_cast<u8*>(pinned) + 64, nr->nodes.template as<u8>() + (nr->ts.local_nodes - 1) * 32, 32, stream));
    CK(msrt::sync(stream));
    nr->ncoef = (size_t)(*reinterpret_cast<unsigned long long*>(pinned));
    memcpy(root, root_on_host ? reinterpret_cast<const u8*>(host_root()) : reinterpret_cast<const u8*>(pinned) + 64, 32);
    nrounds_done++; have_deep = false;
    return MS_OK;
  }
  int fri_round_info(int r, u64* ncoef, u64* D) override {
    if (r < 0 || (size_t)r >= nrounds_done) return fail(MS_ERR_ARG, "round index");
    if (ncoef) *ncoef = rounds[r]->ncoef;
    if (D) *D = rounds[r]->D;
    return MS_OK;
  }
  int fri_round_poly_read(int r, u64* out) override {
    if (r < 0 || (size_t)r >= nrounds_done || !out) return fail(MS_ERR_ARG, "round index");
    if (rounds[r]->local_store) return fail(MS_ERR_STATE, "round_poly_read: the polynomial of a sharded round is distributed over the ranks");
    return download_widen(rounds[r]->poly.template as<T>(), rounds[r]->ncoef, rounds[r]->cap, E, out);
  }
  int fri_round_codeword_read(int r, u64* out) override {
    if (r < 0 || (size_t)r >= nrounds_done || !out) return fail(MS_ERR_ARG, "round index");
    if (rounds[r]->ts.sharded) return fail(MS_ERR_STATE, "codeword_read: the codeword of a sharded round is distributed over the ranks");
    return download_widen(rounds[r]->cw.template as<T>(), rounds[r]->D, rounds[r]->D, E, out);
  }

  // ------------------------------------------------------------------ fri.rs:115-189
  // The whole query phase is a fixed handful of batched launches, whatever the number of
  // rounds and queries: every per-(window, query) step is a job in a device-side table.
  // ext_out != nullptr (ms_fri_query_into): the query-phase kernels write the MSFP blob straight into the caller's buffer - page-locked host
  // memory (ms_pinned_alloc; hipHostMalloc memory is mapped into the device's address space) or device memory - instead of d_blob:
  // no read-back copy afterwards (r02: the runtime executed most of the 64 MiB read-back as shader copies, -15 % on the I/O-inclusive rate)
  bool blob_external = false;
  int fri_query(const u64* betas, int nq, u8* ext_out, size_t ext_cap, size_t* ext_len) override {
    if (nrounds_done != fri_rounds || fri_rounds == 0) return fail(MS_ERR_STATE, "fri_query before the commit phase finished");
    if (!betas || nq < 1) return fail(MS_ERR_ARG, "fri_query");
    const size_t W = fri_rounds - 1;  // windows (previous, round); round W is only evaluated
    // ---- layout of the MSFP blob
    std::vector<size_t> rec_off(W * nq), path_off(W * nq * 2); std::vector<u64> qlen(W ? W : 1);
    size_t pos = 0;
    for (size_t i = 0; i < W; i++) {
      Round* pr = rounds[i];
      if (pr->D / 2 != rounds[i + 1]->D) return fail(MS_ERR_SHAPE, "round domains do not halve (fri.rs:134-137)");
      qlen[i] = pr->ncoef >= 3 ? pr->ncoef - 2 : 0;
      const size_t nlev = pr->ts.levels - 1;
      const size_t path_bytes = 8 + 2 * E * 8 + 8 + nlev * 2 * 32;
      for (int j = 0; j < nq; j++) {
        rec_off[i * nq + j] = pos;
        pos += 6 * E * 8 + 8 + qlen[i] * E * 8;
        path_off[(i * nq + j) * 2] = pos; pos += path_bytes;
        path_off[(i * nq + j) * 2 + 1] = pos; pos += path_bytes;
      }
    }
    if (ext_len) *ext_len = pos;
    if (ext_out && ext_cap == 0) return MS_OK;   // size query: nothing is computed and the previous proof stays readable
    if (ext_out && ext_cap < pos) return fail(MS_ERR_ARG, "ms_fri_query_into: buffer too small (the size needed is in *len)");
    if (W * nq * 2 * 8 > pinned_cap) return fail(MS_ERR_ARG, "too many queries");
    blob_size = 0;
    // ---- device buffers
    size_t sh_elems = 0;
    for (size_t i = 0; i <= W; i++) sh_elems += (size_t)nq * (sh_scratch_elems((rounds[i]->ncoef + 1) / 2) + sh_scratch_elems(rounds[i]->ncoef / 2));
    const size_t n_h0 = (W + 1) * nq * 2 * E, n_tg = (W ? W : 1) * 2 * nq * E;
    if (copy_pending && !ext_out) {   // an asynchronous read-back of the previous proof: finish it before the blob moves, order it before the blob is rewritten
      if (sdma_pending || pos + 8 > d_blob.cap) RQ(fri_proof_wait()); else CK(msrt::stream_wait_event(stream, ev_copy));   // (an SDMA copy is waited for on the host: it ended a whole proof ago)
    }
    if ((!ext_out && d_blob.ensure(pos + 8)) || d_sh.ensure(sh_elems * sizeof(T)) || d_targets.ensure((n_h0 + n_tg) * sizeof(T)) || d_idx.ensure((W ? W : 1) * nq * 2 * 8))
      return fail(MS_ERR_NOMEM, "query buffers");
    u8* blob = ext_out ? ext_out : d_blob.as<u8>();
    blob_external = ext_out != nullptr;
    T* d_h0 = d_targets.as<T>();
    T* d_tg = d_h0 + n_h0;
    unsigned long long* d_ix = d_idx.as<unsigned long long>();
    // ---- suffix Horner jobs: (f - g)/((x-x1)(x-x2)) = Qe(x^2) + x Qo(x^2), Qe = (even(y) - even(x3))/(y - x3) and the
    //      same for odd (fri.rs:159-167); the H_0 outputs are even(x3), odd(x3) (fri.rs:151-153)
    std::vector<T> x1h((W + 1) * nq);
    std::vector<std::vector<SHJ>> tables;  // launch order: every aggregate launch (n_agg_tables of them), then - sharded proofs: behind the carry exchange - every final launch
    std::vector<int> table_mode, table_part;   // table_part: every job of the launch works on this rank's part of a distributed round (profile: partitioned work)
    size_t n_agg_tables = 0;
    // ---- sharded proof, distributed rounds (r04): the quotient jobs BY COEFFICIENT RANGE.  Rank k runs every job over its own half-range [k*S/2, (k+1)*S/2) of the even / odd
    // coefficients with the sum over the higher ranks as carry-in (ONE all-gather of the jobs' aggregates, ShardCarryKernel, which also gives every rank the H_0's),
    // and writes its slice of every quotient polynomial - a contiguous range [t0, t1) of the record's interleaved (even, odd) coefficients - into a packed buffer;
    // the slices are then all-gathered (or gathered to rank 0: ms_shard_proof_on_root) and copied into the blob.
    typedef mspoly::CarryJob<F, E> CJ;
    std::vector<CJ> carry_jobs;
    const size_t Wr = (size_t)sh_world;
    std::vector<size_t> sl_t0(Wr * W * nq, 0), sl_len(Wr * W * nq, 0), sl_off(Wr * W * nq, 0), sl_tot(Wr, 0);   // per rank and (window, query): slice start / length (extension elements) / packed byte offset
    bool any_dist = false;
    for (size_t i = 0; i <= W; i++) any_dist = any_dist || rounds[i]->dist;
    for (size_t i = 0; i < W; i++) {
      Round* pr = rounds[i];
      if (!pr->dist) continue;
      const size_t Sh = pr->S / 2;
      for (size_t r = 0; r < Wr; r++) for (int j = 0; j < nq; j++) {
        const size_t lo_h = r * Sh, hi_h = lo_h + Sh, q_ = qlen[i];
        size_t t1 = 2 * (hi_h - 1); if (t1 > q_) t1 = q_;
        size_t t0 = 2 * ((lo_h > 1 ? lo_h : 1) - 1); if (t0 > t1) t0 = t1;
        const size_t k = (r * W + i) * nq + j;
        sl_t0[k] = t0; sl_len[k] = t1 - t0; sl_off[k] = sl_tot[r]; sl_tot[r] += (t1 - t0) * E * 8;
      }
    }
    size_t pack_max = 0;
    for (size_t r = 0; r < Wr; r++) if (sl_tot[r] > pack_max) pack_max = sl_tot[r];
    if (any_dist && (d_pack.ensure(pack_max + 64) || d_carry.ensure(((W + 1) * nq * 2 + 1) * E * sizeof(T)))) return fail(MS_ERR_NOMEM, "query slices");
    if (any_dist && ((W + 1) * nq * 2 * E * sizeof(T) * Wr > xcap || xcap / Wr < 4096)) return fail(MS_ERR_NOMEM, "exchange buffers too small for the query phase");
    {
      std::vector<SHPlan> shplans;
      T* scr = d_sh.as<T>();
      for (size_t i = 0; i <= W; i++) {
        Round* pr = rounds[i];
        const T gp = f_root_of_unity<F>(ctz64(pr->D));
        const size_t n = pr->ncoef, mm[2] = {(n + 1) / 2, n / 2};
        for (int j = 0; j < nq; j++) {
          u64 beta = betas[j];
          if (i < W) { if (beta > pr->D) beta %= pr->D; }  // fri.rs:144-146 (quirk Q6: `>`)
          else beta %= pr->D;                             // round.domain.element(beta) wraps (fri.rs:150)
          const T x1 = f_pow<F>(gp, beta);                // fri.rs:148
          x1h[i * nq + j] = x1;
          const XE X3 = e_from_base<F, E>(F::mul(x1, x1));
          for (int sgn = 0; sgn < 2; sgn++) {
            void* out = nullptr;
            T* h0 = d_h0 + ((i * nq + j) * 2 + sgn) * E;
            if (pr->dist) {
              const size_t Sh = pr->S / 2, lo_h = (size_t)sh_rank * Sh, cl = mm[sgn] <= lo_h ? 0 : (mm[sgn] - lo_h < Sh ? mm[sgn] - lo_h : Sh);
              const size_t q = carry_jobs.size();
              T* slot_carry = d_carry.as<T>() + q * E;
              size_t out_off = 0;
              if (i < W) {   // the record's coefficient t sits E u64 words after coefficient t - 1: `out` is where coefficient 0 WOULD be in the packed buffer
                const size_t k = ((size_t)sh_rank * W + i) * nq + j;
                out = reinterpret_cast<u64*>(d_pack.as<u8>() + sl_off[k]) - sl_t0[k] * E;
                out_off = (size_t)sgn * E + lo_h * 2 * E;
              }
              CJ cj; memset(&cj, 0, sizeof cj);
              cj.agg_off = (u32)(q * E); cj.has_first = 0; cj.zA = e_one<F, E>(); cj.zB = e_pow<F, E>(X3, (u64)Sh);
              cj.carry_out = slot_carry; cj.tail_out = nullptr; cj.h0_out = h0; cj.scale = e_zero<F, E>();
              if (cl) {
                SHPlan pl = sh_plan(lpoly(pr), lstride(pr), sgn, 2, cl, X3, out, true, 1, out_off, 2 * E, nullptr, scr, slot_carry, reinterpret_cast<T*>(xs) + q * E, sh_rank > 0);
                cj.scale = carry_scale(X3, cl, pl.P);
                pl.dist = true;
                shplans.push_back(pl);
                scr += sh_scratch_elems(cl);
              }
              carry_jobs.push_back(cj);
              continue;
            }
            if (i < W) out = blob + rec_off[i * nq + j] + (6 * E + 1) * 8;
            shplans.push_back(sh_plan(pr->poly.template as<T>(), pr->cap, sgn, 2, mm[sgn], X3, out, true, 1, (size_t)sgn * E, 2 * E, h0, scr));
            scr += sh_scratch_elems(mm[sgn]);
          }
        }
      }
      int max_nl = 1;
      for (auto& pl : shplans) if (pl.nl > max_nl) max_nl = pl.nl;
      // group by level count so that every launch is homogeneous: aggregates bottom-up (the ranks' top-level aggregates last), finals top-down
      for (int nl = 1; nl <= max_nl; nl++) {
        for (int l = 0; l + 1 < nl; l++) { std::vector<SHJ> t; bool ad = true; for (auto& pl : shplans) if (pl.nl == nl) { t.push_back(pl.agg[l]); ad = ad && pl.dist; } if (!t.empty()) { tables.push_back(t); table_mode.push_back(0); table_part.push_back(ad); } }
        { std::vector<SHJ> t; for (auto& pl : shplans) if (pl.nl == nl && pl.has_top_agg) t.push_back(pl.top_agg); if (!t.empty()) { tables.push_back(t); table_mode.push_back(0); table_part.push_back(1); } }
      }
      n_agg_tables = tables.size();
      for (int nl = 1; nl <= max_nl; nl++)
        for (int l = nl - 1; l >= 0; l--) { std::vector<SHJ> t; bool ad = true; for (auto& pl : shplans) if (pl.nl == nl) { t.push_back(pl.fin[l]); ad = ad && pl.dist; } if (!t.empty()) { tables.push_back(t); table_mode.push_back(1); table_part.push_back(ad); } }
    }
    // ---- the slices' way into the blob: chunks of at most xcap / world bytes per rank and exchange
    std::vector<msmerkle::CopyJob> unp; std::vector<size_t> unp_first, unp_cnt, chunk_c0, chunk_len;
    if (any_dist && pack_max) {
      size_t Cb = (pack_max < ((xcap / Wr) & ~(size_t)63)) ? pack_max : ((xcap / Wr) & ~(size_t)63);
      if (shard_gather_chunk && shard_gather_chunk < Cb) Cb = shard_gather_chunk;   // MS_SHARD_GATHER_CHUNK (tests): several exchanges at small sizes
      for (size_t c0 = 0; c0 < pack_max; c0 += Cb) {
        const size_t len = pack_max - c0 < Cb ? pack_max - c0 : Cb;
        chunk_c0.push_back(c0); chunk_len.push_back(len); unp_first.push_back(unp.size());
        for (size_t r = 0; r < Wr; r++) for (size_t i = 0; i < W; i++) for (int j = 0; j < nq; j++) {
          const size_t k = (r * W + i) * nq + j, b0 = sl_off[k], b1 = b0 + sl_len[k] * E * 8;
          const size_t p0 = b0 > c0 ? b0 : c0, p1 = b1 < c0 + len ? b1 : c0 + len;
          if (p0 >= p1) continue;
          unp.push_back(msmerkle::CopyJob{xr + r * len + (p0 - c0), blob + rec_off[i * nq + j] + (6 * E + 1) * 8 + sl_t0[k] * E * 8 + (p0 - b0), p1 - p0});
        }
        unp_cnt.push_back(unp.size() - unp_first.back());
      }
    }
    // ---- find-first and path jobs
    typedef mspoly::FindJob<F, E> FJ;
    typedef msmerkle::PathJob<F, E> PJ;
    // Sharded proof (ms_set_shard): paths are staged in the exchange buffer — every byte written by exactly one rank
    // (replicated rounds: rank 0), summed over the ranks, then copied into the blob.
    typedef msmerkle::ShardPathJob<F, E> SPJ;
    const bool shard = sh_on;
    std::vector<FJ> fjobs(W); std::vector<PJ> pjobs; std::vector<SPJ> sjobs; std::vector<msmerkle::CopyJob> cjobs;
    size_t stage_bytes = 0;
    for (size_t i = 0; i < W; i++) {
      Round* pr = rounds[i];
      const size_t nlev = pr->ts.levels - 1, path_bytes = 8 + 2 * E * 8 + 8 + nlev * 2 * 32;
      if (pr->ts.sharded) {
        fjobs[i] = FJ{pr->cw.template as<T>(), 2 * pr->m, 2 * pr->m, d_tg + i * 2 * nq * E, 2 * nq, d_ix + i * 2 * nq, 2, (u32)sh_world, (u32)sh_rank, pr->m};
      } else fjobs[i] = FJ{pr->cw.template as<T>(), pr->D, pr->D, d_tg + i * 2 * nq * E, 2 * nq, d_ix + i * 2 * nq, 0, 0, 0, 0};
      for (int t = 0; t < 2 * nq; t++) {
        u8* dst = blob + path_off[(i * nq + t / 2) * 2 + (t & 1)];
        u8* out = dst;
        if (shard) { out = xs + stage_bytes; cjobs.push_back(msmerkle::CopyJob{out, dst, path_bytes}); stage_bytes += path_bytes; }
        if (pr->ts.sharded)
          sjobs.push_back(SPJ{pr->cw.template as<T>(), 2 * pr->m, pr->m, pr->nodes.template as<u32>(), pr->nodes.template as<u32>() + (2 * pr->ts.Mloc - 1) * 8, pr->ts.Mloc,
                              2, (u32)sh_world, (u32)sh_rank, (u32)nlev, d_ix + i * 2 * nq + t, out});
        else if (!shard || sh_rank == 0)
          pjobs.push_back(PJ{pr->cw.template as<T>(), pr->D, pr->nodes.template as<u32>(), pr->D, 2, 2, (u32)nlev, d_ix + i * 2 * nq + t, out});
      }
    }
    if (shard && (stage_bytes > xcap || W * nq * 2 * 8 > xcap)) return fail(MS_ERR_NOMEM, "exchange buffers too small for the query phase");
    // ---- one upload: [SH tables][find jobs][path jobs][rec_off][qlen][x1]
    size_t bytes = 0;
    std::vector<size_t> toff(tables.size());
    for (size_t k = 0; k < tables.size(); k++) { toff[k] = bytes; bytes += tables[k].size() * sizeof(SHJ); }
    const size_t off_f = bytes; bytes += fjobs.size() * sizeof(FJ);
    const size_t off_p = bytes; bytes += pjobs.size() * sizeof(PJ);
    const size_t off_sp = bytes; bytes += sjobs.size() * sizeof(SPJ);
    const size_t off_cj = bytes; bytes += cjobs.size() * sizeof(msmerkle::CopyJob);
    const size_t off_rec = bytes; bytes += rec_off.size() * sizeof(size_t);
    const size_t off_ql = bytes; bytes += qlen.size() * 8;
    const size_t off_x1 = bytes; bytes += x1h.size() * sizeof(T);
    bytes = (bytes + 15) & ~(size_t)15;
    const size_t off_cr = bytes; bytes += carry_jobs.size() * sizeof(CJ);
    const size_t off_un = bytes; bytes += unp.size() * sizeof(msmerkle::CopyJob);
    u8* tab;
    RQ(tabs_host(bytes + 8, &tab));
    for (size_t k = 0; k < tables.size(); k++) memcpy(tab + toff[k], tables[k].data(), tables[k].size() * sizeof(SHJ));
    if (!fjobs.empty()) memcpy(tab + off_f, fjobs.data(), fjobs.size() * sizeof(FJ));
    if (!pjobs.empty()) memcpy(tab + off_p, pjobs.data(), pjobs.size() * sizeof(PJ));
    if (!sjobs.empty()) memcpy(tab + off_sp, sjobs.data(), sjobs.size() * sizeof(SPJ));
    if (!cjobs.empty()) memcpy(tab + off_cj, cjobs.data(), cjobs.size() * sizeof(msmerkle::CopyJob));
    if (!rec_off.empty()) memcpy(tab + off_rec, rec_off.data(), rec_off.size() * sizeof(size_t));
    memcpy(tab + off_ql, qlen.data(), qlen.size() * 8);
    memcpy(tab + off_x1, x1h.data(), x1h.size() * sizeof(T));
    if (!carry_jobs.empty()) memcpy(tab + off_cr, carry_jobs.data(), carry_jobs.size() * sizeof(CJ));
    if (!unp.empty()) memcpy(tab + off_un, unp.data(), unp.size() * sizeof(msmerkle::CopyJob));
    if (d_tabs.ensure(bytes + 8)) return fail(MS_ERR_NOMEM, "query tables");
    CK(msrt::h2d(d_tabs.p, tab, bytes + 8, stream));   // page-locked source: no synchronisation here (this stage ends with one; the area is next written by the next proof)
    const u8* dt = d_tabs.as<u8>();
    // ---- launches
    if (!carry_jobs.empty()) CK(msrt::memset_dev(xs, 0, carry_jobs.size() * E * sizeof(T), stream));   // aggregates of the jobs this rank has no coefficients for
    for (size_t k = 0; k < tables.size(); k++) {
      if (k == n_agg_tables && !carry_jobs.empty()) {   // every aggregate is in the send buffer: one all-gather, then the carry-ins of this rank's jobs and every job's H_0
        RQ(exchange(MS_XCHG_ALL_GATHER, carry_jobs.size() * E * sizeof(T)));
        typedef mspoly::ShardCarryKernel<F, E> CKn;
        typename CKn::Params cp; memset(&cp, 0, sizeof cp);
        cp.jobs = reinterpret_cast<const CJ*>(dt + off_cr); cp.njobs = (u32)carry_jobs.size(); cp.W = (u32)sh_world; cp.rank = (u32)sh_rank;
        cp.gathered = reinterpret_cast<const T*>(xr); cp.rank_stride = carry_jobs.size() * E;
        CK(run<CKn>(K_SUFFIX_HORNER, grid1(carry_jobs.size(), CKn::THREADS), 1, CKn::THREADS, 0, cp));
      }
      size_t maxnb = 1;
      for (auto& j : tables[k]) { size_t nb = j.m ? (j.m + mspoly::SH_BS - 1) / mspoly::SH_BS : 1; if (nb > maxnb) maxnb = nb; }
      typename SHK::Params p; p.jobs = reinterpret_cast<const SHJ*>(dt + toff[k]); p.final_mode = table_mode[k];
      memset(&p.inline_job, 0, sizeof p.inline_job);
      if (table_part[k]) part_depth++;
      const int e_ = run_coop<SHK>(K_SUFFIX_HORNER, (unsigned)maxnb, SHK::THREADS, SHK::lds_bytes(), p, (unsigned)tables[k].size());
      if (table_part[k]) part_depth--;
      CK(e_);
    }
    if (n_agg_tables == tables.size() && !carry_jobs.empty()) {   // (no job on this rank at all: the exchange still has to happen - the other ranks are in it)
      RQ(exchange(MS_XCHG_ALL_GATHER, carry_jobs.size() * E * sizeof(T)));
      typedef mspoly::ShardCarryKernel<F, E> CKn;
      typename CKn::Params cp; memset(&cp, 0, sizeof cp);
      cp.jobs = reinterpret_cast<const CJ*>(dt + off_cr); cp.njobs = (u32)carry_jobs.size(); cp.W = (u32)sh_world; cp.rank = (u32)sh_rank;
      cp.gathered = reinterpret_cast<const T*>(xr); cp.rank_stride = carry_jobs.size() * E;
      CK(run<CKn>(K_SUFFIX_HORNER, grid1(carry_jobs.size(), CKn::THREADS), 1, CKn::THREADS, 0, cp));
    }
    for (size_t c = 0; c < chunk_c0.size(); c++) {   // the ranks' slices of the quotient polynomials into the blob
      CK(msrt::d2d(xs, d_pack.as<u8>() + chunk_c0[c], chunk_len[c], stream));
      RQ(exchange(proof_root_only ? MS_XCHG_GATHER : MS_XCHG_ALL_GATHER, chunk_len[c]));
      if (unp_cnt[c] && (!proof_root_only || sh_rank == 0)) {
        size_t maxw = 1;
        for (size_t u = 0; u < unp_cnt[c]; u++) { const size_t wgs = (unp[unp_first[c] + u].bytes / 8 + msmerkle::CopyRangesKernel::WORDS - 1) / msmerkle::CopyRangesKernel::WORDS; if (wgs > maxw) maxw = wgs; }
        msmerkle::CopyRangesKernel::Params up{reinterpret_cast<const msmerkle::CopyJob*>(dt + off_un) + unp_first[c], (u32)unp_cnt[c]};
        CK(run<msmerkle::CopyRangesKernel>(K_IO, (unsigned)maxw, (unsigned)unp_cnt[c], msmerkle::CopyRangesKernel::THREADS, 0, up));
      }
    }
    if (W) {
      typename mspoly::QueryPointsKernel<F, E>::Params qp{d_h0, reinterpret_cast<const T*>(dt + off_x1), reinterpret_cast<const u64*>(dt + off_ql), (int)W, nq, blob,
                                                         reinterpret_cast<const size_t*>(dt + off_rec), d_tg};
      CK(run<mspoly::QueryPointsKernel<F, E>>(K_QUERY_POINTS, grid1(W * nq, 64), 1, 64, 0, qp));
      CK(msrt::memset_dev(d_ix, 0xFF, W * nq * 2 * 8, stream));
      // leaf lookup BY VALUE, first match (merkle.rs:216-225, quirk Q7): big codewords one launch each, the rest batched
      size_t first_small = W;
      for (size_t i = 0; i < W; i++) if (rounds[i]->D <= ((size_t)1 << 16)) { first_small = i; break; }
      for (size_t i = 0; i < first_small; i++) {
        typename mspoly::FindFirstKernel<F, E>::Params fp; fp.jobs = nullptr; fp.inline_job = fjobs[i];
        if (rounds[i]->ts.sharded) part_depth++;
        const int e_ = run<mspoly::FindFirstKernel<F, E>>(K_FIND_FIRST, grid1(rounds[i]->ts.sharded ? 2 * rounds[i]->m : rounds[i]->D, mspoly::THREADS), 1, mspoly::THREADS, 0, fp);
        if (rounds[i]->ts.sharded) part_depth--;
        CK(e_);
      }
      if (first_small < W) {
        typename mspoly::FindFirstKernel<F, E>::Params fp; fp.jobs = reinterpret_cast<const FJ*>(dt + off_f) + first_small; fp.inline_job = fjobs[first_small];
        CK(run<mspoly::FindFirstKernel<F, E>>(K_FIND_FIRST, grid1(rounds[first_small]->D, mspoly::THREADS), (unsigned)(W - first_small), mspoly::THREADS, 0, fp));
      }
      if (shard) {  // first match over ALL ranks' parts: minimum global index (replicated rounds: every rank holds the same value)
        CK(msrt::d2d(xs, d_ix, W * nq * 2 * 8, stream));
        RQ(exchange(MS_XCHG_ALL_REDUCE_MIN_U64, W * nq * 2 * 8));
        CK(msrt::d2d(d_ix, xs, W * nq * 2 * 8, stream));
        CK(msrt::memset_dev(xs, 0, stage_bytes, stream));
      }
      if (!pjobs.empty()) {
        typename msmerkle::PathKernel<F, E>::Params pk{reinterpret_cast<const PJ*>(dt + off_p), (u32)pjobs.size()};
        CK(run<msmerkle::PathKernel<F, E>>(K_PATH, (unsigned)pjobs.size(), 1, 64, 0, pk));   // one wave per opening
      }
      if (!sjobs.empty()) {
        typename msmerkle::ShardPathKernel<F, E>::Params sk{reinterpret_cast<const SPJ*>(dt + off_sp), (u32)sjobs.size()};
        CK(run<msmerkle::ShardPathKernel<F, E>>(K_PATH, grid1(sjobs.size(), 64), 1, 64, 0, sk));
      }
      if (shard) {
        RQ(exchange(MS_XCHG_ALL_REDUCE_SUM_U8, stage_bytes));
        msmerkle::CopyJobsKernel::Params ck{reinterpret_cast<const msmerkle::CopyJob*>(dt + off_cj), (u32)cjobs.size()};
        CK(run<msmerkle::CopyJobsKernel>(K_PATH, (unsigned)cjobs.size(), 1, msmerkle::CopyJobsKernel::THREADS, 0, ck));
      }
      CK(msrt::d2h(pinned, d_ix, W * nq * 2 * 8, stream));
      CK(msrt::sync(stream));
      const unsigned long long* hidx = reinterpret_cast<const unsigned long long*>(pinned);
      for (size_t t = 0; t < W * nq * 2; t++) if (hidx[t] == ~0ULL) return fail(MS_ERR_LEAF_NOT_FOUND, "leaf is not included in the tree");
    } else CK(msrt::sync(stream));
    blob_size = (proof_root_only && sh_world > 1 && sh_rank != 0) ? 0 : pos;   // ms_shard_proof_on_root: the blob is whole on rank 0 only
    return MS_OK;
  }
  size_t fri_proof_size() const override { return blob_size; }
  int fri_proof_read(u8* out) override {
    if (!blob_size || !out) return fail(MS_ERR_STATE, "no FRI proof");
    if (blob_external) return fail(MS_ERR_STATE, "the FRI proof was written to the caller's buffer (ms_fri_query_into)");
    RQ(fri_proof_wait());
    if (readback_sdma > 1 && sdma_ready() && msrt::is_pinned_host(out)) {   // MS_READBACK=sdma-all: the blocking read on the copy engine as well (page-locked destinations only)
      RQ(fri_proof_read_async(out));
      return fri_proof_wait();
    }
    last_io_engine = 0;
    CK(msrt::d2h(out, d_blob.p, blob_size, stream));
    CK(msrt::sync(stream));
    return MS_OK;
  }
  // The same copy on the context's COPY stream, ordered behind the query phase by an event: the call returns at once and the next proof's
  // stages run while the ~64 MiB travel; the next ms_fri_query waits (on the device, not on the host) for the copy before it rewrites the blob.
  msrt::Stream* copy_stream = nullptr; msrt::Event* ev_blob = nullptr; msrt::Event* ev_copy = nullptr; bool copy_pending = false;
  // r04: the read-back as an explicit SDMA copy through the HSA runtime (rt.hpp, msrt::Sdma) - queued on a copy engine whatever the engines' load, never a
  // blit kernel; completion is an HSA signal the host waits on (ms_fri_proof_wait, or the next ms_fri_query before it rewrites the blob).  MS_READBACK=hip keeps
  // hipMemcpyAsync on the copy stream (A/B); io_engine() reports which path the last read-back took.
  int readback_sdma = 2, sdma_gpu = -1, sdma_state = 0 /* 0 untried, 1 bound, 2 unavailable */, last_io_engine = 0;
  msrt::Sdma::Signal sdma_sig{0}, sdma_up_sig{0}; bool sdma_pending = false; int upload_sdma = 1;
  bool sdma_ready() {
    if (!readback_sdma) return false;
    if (sdma_state == 0) {
      msrt::Sdma& S = msrt::Sdma::get();
      if (!S.bind_device(device, &sdma_gpu)) sdma_state = (!S.signal_create(&sdma_sig) && !S.signal_create(&sdma_up_sig)) ? 1 : 2;
      else if (S.unavailable()) sdma_state = 2;   // no HSA runtime to bind: this context stays on the HIP runtime's copies.  (Engines merely busy: asked again at the next copy.)
    }
    return sdma_state == 1;
  }
  int io_engine() const override { return last_io_engine; }
  int fri_proof_read_async(u8* out) override {
    if (!blob_size || !out) return fail(MS_ERR_STATE, "no FRI proof");
    if (blob_external) return fail(MS_ERR_STATE, "the FRI proof was written to the caller's buffer (ms_fri_query_into)");
    RQ(fri_proof_wait());   // one read-back in flight per context
    if (sdma_ready() && msrt::is_pinned_host(out)) {     // ms_fri_query ended with a stream synchronisation: the blob is complete, the copy needs no dependency (the synchronisation here returns at once)
      CK(msrt::sync(stream));
      msrt::Sdma& S = msrt::Sdma::get();
      const int e = S.copy_d2h(sdma_gpu, out, d_blob.p, blob_size, sdma_sig, S.d2h_engine(sdma_gpu));
      if (!e) { sdma_pending = true; copy_pending = true; last_io_engine = 1; return MS_OK; }
      sdma_state = 2;       // refused (engine id / access): this context stays on the runtime's copy from here on
    }
    last_io_engine = 0;
    if (!copy_stream) { CK(msrt::stream_create(&copy_stream)); CK(msrt::event_create(&ev_blob)); CK(msrt::event_create(&ev_copy)); }
    CK(msrt::event_record(ev_blob, stream));
    CK(msrt::stream_wait_event(copy_stream, ev_blob));
    CK(msrt::d2h(out, d_blob.p, blob_size, copy_stream));
    CK(msrt::event_record(ev_copy, copy_stream));
    copy_pending = true;
    return MS_OK;
  }
  int fri_proof_wait() override {
    if (sdma_pending) {
      sdma_pending = false; copy_pending = false;
      if (msrt::Sdma::get().wait(sdma_sig, 20.0)) return fail(MS_ERR_HIP, "SDMA read-back did not complete within 20 s");
      return MS_OK;
    }
    if (copy_pending) { CK(msrt::event_sync(ev_copy)); copy_pending = false; }
    return MS_OK;
  }

  // ------------------------------------------------------------------ standalone entry points
  int merkle_commit(const u64* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, u8* nodes_out, size_t cap, size_t* nn, u8* root) override {
    if (!leafs && leaf_num) return fail(MS_ERR_ARG, "null leafs");
    if (ext != 1 && ext != E) return fail(MS_ERR_ARG, "ext must be 1 or the field's extension degree");
    TreeShape ts;
    RQ(tree_shape(leaf_num, lpn, ic, &ts));
    if (!canonical(leafs, leaf_num * ext)) return fail(MS_ERR_ARG, "leaf not canonical");
    DevBuf dl, dn;
    if (dl.ensure(leaf_num * ext * sizeof(T))) return fail(MS_ERR_NOMEM, "leafs");
    int rc = upload_narrow(leafs, leaf_num * ext, dl.as<T>());
    // AoS view: element f limb k at base + f*ext + k
    if (!rc) rc = (ext == 1) ? tree_build<1>(dl.as<T>(), 0, 1, 0, 1, ts, dn) : tree_build<E>(dl.as<T>(), 0, (size_t)ext, 1, 1, ts, dn);
    if (!rc && nn) *nn = ts.nodes;
    if (!rc && nodes_out) {
      if (cap < ts.nodes) rc = fail(MS_ERR_ARG, "nodes_out too small");
      else { int e = msrt::d2h(nodes_out, dn.p, ts.nodes * 32, stream); if (!e) e = msrt::sync(stream); if (e) rc = fail_rt(e, "nodes d2h"); }
    }
    if (!rc && root) rc = read_root(dn, ts, root);
    msrt::sync(stream);
    dl.release(); dn.release();
    return rc;
  }
  // src/merkle.rs:272-288 on a standalone binary tree: leaves de-interleaved to SoA limbs, then the same
  // LeafHash / InnerHash / FindFirst / MerklePath kernels the FRI query phase uses
  template <int EL>
  int merkle_prove_t(const u64* leafs, size_t leaf_num, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) {
    TreeShape ts;
    RQ(tree_shape(leaf_num, lpn, 2, &ts));
    if (!canonical(leafs, leaf_num * EL) || !canonical(leaf, EL)) return fail(MS_ERR_ARG, "leaf not canonical");
    const size_t plen = 8 + lpn * EL * 8 + 8 + (ts.levels - 1) * 2 * 32;
    if (len) *len = plen;
    if (!out || cap < plen) return fail(MS_ERR_ARG, "path buffer too small");
    DevBuf ds, dn, dw;  // SoA leaves | nodes | {target | idx | job tables | path}
    typedef mspoly::FindJob<F, EL> FJ;
    typedef msmerkle::PathJob<F, EL> PJ;
    const size_t off_ix = 256, off_pj = 512, off_path = 1024;
    int rc = 0;
    if (ds.ensure(leaf_num * EL * sizeof(T)) || dw.ensure(off_path + plen) || d_io.ensure(leaf_num * EL * 8)) rc = fail(MS_ERR_NOMEM, "merkle_prove");
    if (!rc) {
      int e = msrt::h2d(d_io.p, leafs, leaf_num * EL * 8, stream);
      typename mspoly::TransposeInKernel<F>::Params tp{d_io.as<u64>(), ds.as<T>(), leaf_num, (size_t)EL, leaf_num, F::from_u64(1), 0, nullptr};
      if (!e) e = run<mspoly::TransposeInKernel<F>>(K_TRANSPOSE, grid1(leaf_num * EL, mspoly::THREADS), 1, mspoly::THREADS, 0, tp);
      if (e) rc = fail_rt(e, "leaf upload");
    }
    if (!rc) rc = tree_build<EL>(ds.as<T>(), 0, 1, leaf_num, 1, ts, dn);
    if (!rc) {
      T tgt[EL]; for (int l = 0; l < EL; l++) tgt[l] = F::from_u64(leaf[l]);
      unsigned long long none = ~0ULL;
      u8* base = dw.as<u8>();
      PJ pj{ds.as<T>(), leaf_num, dn.as<u32>(), leaf_num, (u32)lpn, 2, (u32)(ts.levels - 1), reinterpret_cast<unsigned long long*>(base + off_ix), base + off_path};
      int e = msrt::h2d(base, tgt, sizeof tgt, stream);
      if (!e) e = msrt::h2d(base + off_ix, &none, 8, stream);
      if (!e) e = msrt::h2d(base + off_pj, &pj, sizeof pj, stream);
      if (!e) e = msrt::sync(stream);  // stack sources
      typename mspoly::FindFirstKernel<F, EL>::Params fp; fp.jobs = nullptr;
      fp.inline_job = FJ{ds.as<T>(), leaf_num, leaf_num, reinterpret_cast<const T*>(base), 1, reinterpret_cast<unsigned long long*>(base + off_ix)};
      if (!e) e = run<mspoly::FindFirstKernel<F, EL>>(K_FIND_FIRST, grid1(leaf_num, mspoly::THREADS), 1, mspoly::THREADS, 0, fp);   // merkle.rs:216-225
      typename msmerkle::PathKernel<F, EL>::Params pk{reinterpret_cast<const PJ*>(base + off_pj), 1};
      if (!e) e = run<msmerkle::PathKernel<F, EL>>(K_PATH, 1, 1, 64, 0, pk);                                                          // merkle.rs:230-288
      if (!e) e = msrt::d2h(pinned, base + off_ix, 8, stream);
      if (!e) e = msrt::d2h(out, base + off_path, plen, stream);
      if (!e) e = msrt::sync(stream);
      if (e) rc = fail_rt(e, "merkle_prove");
      else if (*reinterpret_cast<unsigned long long*>(pinned) == ~0ULL) rc = fail(MS_ERR_LEAF_NOT_FOUND, "leaf is not included in the tree");
    }
    msrt::sync(stream);
    ds.release(); dn.release(); dw.release();
    return rc;
  }
  int merkle_prove(const u64* leafs, size_t leaf_num, int ext, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) override {
    if (!leafs || !leaf) return fail(MS_ERR_ARG, "null argument");
    if (ext == 1) return merkle_prove_t<1>(leafs, leaf_num, lpn, leaf, out, cap, len);
    if (ext == E) return merkle_prove_t<E>(leafs, leaf_num, lpn, leaf, out, cap, len);
    return fail(MS_ERR_ARG, "ext must be 1 or the field's extension degree");
  }
  int ntt(u64* data, size_t n, size_t batch, int inverse) override {
    if (!data || !n || !is_pow2(n)) return fail(MS_ERR_SHAPE, "ntt size must be a power of two");
    if (ctz64(n) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "ntt size exceeds two-adicity");
    if (!canonical(data, n * batch)) return fail(MS_ERR_ARG, "element not canonical");
    DevBuf d;
    if (d.ensure(n * batch * sizeof(T))) return fail(MS_ERR_NOMEM, "ntt buffer");
    int rc = upload_narrow(data, n * batch, d.as<T>());
    if (!rc) rc = ntt_run(ctz64(n), inverse != 0, d.as<T>(), n, n, d.as<T>(), n, batch);
    if (!rc) rc = download_widen(d.as<T>(), n * batch, 0, 1, data);
    msrt::sync(stream);
    d.release();
    return rc;
  }
  int coset_lde(const u64* coeffs, size_t ncoef, size_t batch, u64 shift, u64* out, size_t L_) override {
    if (!coeffs || !out || !L_ || !is_pow2(L_) || ncoef > L_ || !ncoef) return fail(MS_ERR_SHAPE, "coset_lde shape");
    if (ctz64(L_) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "domain exceeds two-adicity");
    if (shift == 0 || shift >= F::P || !canonical(coeffs, ncoef * batch)) return fail(MS_ERR_ARG, "not canonical");
    DevBuf a, b;
    if (a.ensure(ncoef * batch * sizeof(T)) || b.ensure(L_ * batch * sizeof(T))) return fail(MS_ERR_NOMEM, "lde buffers");
    int rc = upload_narrow(coeffs, ncoef * batch, a.as<T>());
    if (!rc) {
      typename msntt::ScalePowKernel<F>::Params sp;
      sp.src = a.as<T>(); sp.dst = a.as<T>(); sp.src_bstride = ncoef; sp.dst_bstride = ncoef; sp.n = ncoef;
      sp.s = F::from_u64(shift); sp.s_step = f_pow<F>(sp.s, msntt::ScalePowKernel<F>::THREADS);
      const int per_block = msntt::ScalePowKernel<F>::THREADS * msntt::ScalePowKernel<F>::ITEMS;
      int e = run<msntt::ScalePowKernel<F>>(K_SCALE_POW, grid1(ncoef, per_block), (unsigned)batch, msntt::ScalePowKernel<F>::THREADS, 0, sp);
      if (e) rc = fail_rt(e, "scale");
    }
    if (!rc) rc = ntt_run(ctz64(L_), false, a.as<T>(), ncoef, ncoef, b.as<T>(), L_, batch);
    if (!rc) rc = download_widen(b.as<T>(), L_ * batch, 0, 1, out);
    msrt::sync(stream);
    a.release(); b.release();
    return rc;
  }
};

inline CtxBase* B(ms_ctx* c) { return reinterpret_cast<CtxBase*>(c); }
inline const CtxBase* B(const ms_ctx* c) { return reinterpret_cast<const CtxBase*>(c); }

}  // namespace

// ---- "never unwind".  The stage functions build std::vector job tables, std::string error texts and a std::map of NTT plans, and ms_create
// constructs the context: every entry point that can reach an allocation runs inside this guard - std::bad_alloc becomes MS_ERR_NOMEM, anything else
// MS_ERR_HIP - so that no C++ exception ever crosses the C ABI into a Rust / C caller (undefined behaviour there).  The error text is set without
// allocating (literals short enough for the small-string buffer).  tests/test_emu_parity.py::test_c_abi_never_unwinds drives a whole proof with the N-th
// allocation of the emulation build failing, for every N.
static int on_exception(const ms_ctx* ctx, bool oom) noexcept {
  try { if (ctx) const_cast<CtxBase*>(B(ctx))->err = oom ? "out of memory" : "C++ exception"; } catch (...) {}
  return oom ? MS_ERR_NOMEM : MS_ERR_HIP;
}
#define MS_ENTRY(ctx, expr) do { if (!(ctx)) return MS_ERR_ARG; try { B(ctx)->bind_device(); return (expr); } \
  catch (const std::bad_alloc&) { return on_exception(ctx, true); } catch (...) { return on_exception(ctx, false); } } while (0)

#ifdef MS_EMU
// ---- allocation-failure hook of the EMULATION build (tests only): operator new of this library alone (hidden visibility: the process's other libraries keep theirs)
// counts its calls and throws std::bad_alloc on the one ms_emu_fail_alloc_after() armed
static long g_emu_alloc_count = 0, g_emu_fail_at = -1;
static void* emu_alloc(size_t n) {
  const long k = g_emu_alloc_count++;
  if (g_emu_fail_at >= 0 && k == g_emu_fail_at) { g_emu_fail_at = -1; throw std::bad_alloc(); }
  void* p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
// (kept out of the dynamic symbol table by tests/emu/emu.map: only this library allocates through them)
#define MS_HIDDEN
MS_HIDDEN void* operator new(size_t n) { return emu_alloc(n); }
MS_HIDDEN void* operator new[](size_t n) { return emu_alloc(n); }
MS_HIDDEN void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return emu_alloc(n); } catch (...) { return nullptr; } }
MS_HIDDEN void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return emu_alloc(n); } catch (...) { return nullptr; } }
MS_HIDDEN void operator delete(void* p) noexcept { std::free(p); }
MS_HIDDEN void operator delete[](void* p) noexcept { std::free(p); }
MS_HIDDEN void operator delete(void* p, size_t) noexcept { std::free(p); }
MS_HIDDEN void operator delete[](void* p, size_t) noexcept { std::free(p); }
extern "C" void ms_emu_fail_alloc_after(long n) { g_emu_fail_at = n < 0 ? -1 : g_emu_alloc_count + n; }
extern "C" long ms_emu_alloc_count() { return g_emu_alloc_count; }
#endif

extern "C" {

int ms_create(ms_ctx** out, int device, ms_field field, uint32_t flags) {
  if (!out) return MS_ERR_ARG;
  *out = nullptr;
  if (field != MS_FIELD_GOLDILOCKS && field != MS_FIELD_BABYBEAR) return MS_ERR_ARG;
  CtxBase* c = nullptr;
  try {
    int rc;
    if (field == MS_FIELD_GOLDILOCKS) { auto* g = new (std::nothrow) Ctx<GL>(); if (!g) return MS_ERR_NOMEM; c = g; rc = g->init(device, flags); }
    else { auto* b = new (std::nothrow) Ctx<BB>(); if (!b) return MS_ERR_NOMEM; c = b; rc = b->init(device, flags); }
    if (rc) { delete c; return rc; }
  } catch (const std::bad_alloc&) { delete c; return MS_ERR_NOMEM; } catch (...) { delete c; return MS_ERR_HIP; }
  *out = reinterpret_cast<ms_ctx*>(c);
  return MS_OK;
}
void ms_destroy(ms_ctx* ctx) { if (ctx) { try { B(ctx)->bind_device(); delete B(ctx); } catch (...) {} } }
const char* ms_last_error(const ms_ctx* ctx) { return ctx ? B(ctx)->err.c_str() : "null context"; }
int ms_ext_degree(const ms_ctx* ctx) { return ctx ? B(ctx)->ext_degree() : MS_ERR_ARG; }
int ms_set_stream(ms_ctx* ctx, void* s) { MS_ENTRY(ctx, B(ctx)->set_stream(s)); }
int ms_set_shard(ms_ctx* ctx, int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user) {
  MS_ENTRY(ctx, B(ctx)->set_shard(rank, world, d_send, d_recv, cap, fn, user));
}
void* ms_pinned_alloc(size_t bytes) { void* p = nullptr; return msrt::malloc_host(&p, bytes) ? nullptr : p; }
void ms_pinned_free(void* p) { if (p) msrt::free_host(p); }
int ms_rccl_unique_id(uint8_t out[128]) {
  if (!out) return MS_ERR_ARG;
  try {
    msrt::Rccl& R = msrt::Rccl::get();
    if (R.load()) return MS_ERR_HIP;
    msrt::Rccl::UniqueId id;
    if (R.get_unique_id(&id)) return MS_ERR_HIP;
    memcpy(out, id.internal, 128);
    return MS_OK;
  } catch (const std::bad_alloc&) { return MS_ERR_NOMEM; } catch (...) { return MS_ERR_HIP; }
}
int ms_set_shard_rccl(ms_ctx* ctx, int rank, int world, const uint8_t unique_id[128], size_t cap_bytes) {
  MS_ENTRY(ctx, B(ctx)->set_shard_rccl(rank, world, unique_id, cap_bytes));
}
int ms_rccl_selftest(ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->rccl_selftest()); }
int ms_shard_stats(ms_ctx* ctx, uint64_t out[8]) { MS_ENTRY(ctx, B(ctx)->shard_stats(out)); }
int ms_shard_round_is_distributed(ms_ctx* ctx, int round) { if (!ctx) return MS_ERR_ARG; return B(ctx)->shard_round_is_distributed(round); }
int ms_shard_proof_is_elsewhere(const ms_ctx* ctx) { return ctx ? B(ctx)->shard_proof_is_elsewhere() : MS_ERR_ARG; }
int ms_shard_proof_on_root(ms_ctx* ctx, int on) { if (!ctx) return MS_ERR_ARG; return B(ctx)->shard_proof_on_root(on); }
int ms_shard_slice_layout(ms_ctx* ctx, size_t* offset, size_t* stride) { MS_ENTRY(ctx, B(ctx)->shard_slice_layout(offset, stride)); }
int ms_synchronize(ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->synchronize()); }

int ms_is_power_of_two(uint64_t n) { return is_pow2(n) ? 1 : 0; }
long ms_logarithm_of_two_k(uint64_t n, uint64_t base) { return log_two_k(n, base); }
uint64_t ms_ceil_log2_k(uint64_t n, uint64_t base) { return ceil_log2_k(n, base); }
// src/starks.rs:312-332
int ms_num_queries(ms_field f, uint64_t security_bits, uint64_t blowup, uint64_t steps, uint64_t* linking, uint64_t* fri) {
  if (!linking || !fri || !blowup || !steps) return MS_ERR_ARG;
  if (security_bits < 20) return MS_ERR_SHAPE;  // starks.rs:317-320 panics
  const u64 modulus_bits = (f == MS_FIELD_GOLDILOCKS) ? 64 : 31;
  const u64 log_steps = ceil_log2_k(steps, 2);
  if (log_steps >= modulus_bits) return MS_ERR_SHAPE;          // starks.rs:322 would divide by zero / underflow (panics)
  if (steps > ~(u64)0 / blowup) return MS_ERR_SHAPE;           // steps * blowup overflows u64 (starks.rs:277 panics in debug)
  const u64 den = modulus_bits - log_steps;
  *linking = (security_bits + den - 1) / den;
  const u64 rounds = ceil_log2_k(steps * blowup, 2);
  const double rho = 1.0 / (double)blowup;
  const double denominator = __builtin_log2(2.0 / (1.0 + rho));
  const double total = (double)security_bits / denominator;
  *fri = (u64)__builtin_ceil(total / (double)rounds);
  return MS_OK;
}
uint64_t ms_root_of_unity(ms_field f, uint64_t n) {
  if (!n || !is_pow2(n)) return 0;
  const int lg = ctz64(n);
  if (f == MS_FIELD_GOLDILOCKS) return lg <= GL::TWO_ADICITY ? GL::to_u64(f_root_of_unity<GL>(lg)) : 0;
  return lg <= BB::TWO_ADICITY ? BB::to_u64(f_root_of_unity<BB>(lg)) : 0;
}

int ms_trace_commit(ms_ctx* ctx, const uint64_t* t, size_t N, size_t w, size_t lpn, uint8_t root[32]) { MS_ENTRY(ctx, B(ctx)->trace_commit(t, false, N, w, lpn, root)); }
int ms_trace_commit_device(ms_ctx* ctx, const void* t, size_t N, size_t w, size_t lpn, uint8_t root[32]) { MS_ENTRY(ctx, B(ctx)->trace_commit(reinterpret_cast<const u64*>(t), true, N, w, lpn, root)); }
int ms_interpolate(ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->interpolate()); }
int ms_polys_lincomb(ms_ctx* ctx, const uint64_t* s, const int* idx, int k) { MS_ENTRY(ctx, B(ctx)->polys_lincomb(s, idx, k)); }
int ms_polys_append(ms_ctx* ctx, const uint64_t* c, size_t n) { MS_ENTRY(ctx, B(ctx)->polys_append(c, n)); }
int ms_polys_count(const ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->polys_count()); }
int ms_poly_read(ms_ctx* ctx, int i, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->poly_read(i, out)); }
int ms_lde_commit(ms_ctx* ctx, size_t blowup, uint64_t shift, size_t lpn, uint8_t root[32]) { MS_ENTRY(ctx, B(ctx)->lde_commit(blowup, shift, lpn, root)); }
int ms_lde_read(ms_ctx* ctx, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->lde_read(out)); }
int ms_mix(ms_ctx* ctx, uint64_t r) { MS_ENTRY(ctx, B(ctx)->mix(r)); }
int ms_validity_read(ms_ctx* ctx, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->validity_read(out)); }
int ms_mix_cubic(ms_ctx* ctx, uint64_t r, const int* spec, const uint64_t* s, int ncons) { MS_ENTRY(ctx, B(ctx)->mix_cubic(r, spec, s, ncons)); }
size_t ms_validity_len(const ms_ctx* ctx) { return ctx ? B(ctx)->validity_len_() : 0; }
int ms_eval_ext(ms_ctx* ctx, const uint64_t* z, int q, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->eval_ext(z, q, out)); }
int ms_fri_begin(ms_ctx* ctx, size_t blowup, size_t rounds, uint8_t root0[32]) { MS_ENTRY(ctx, B(ctx)->fri_begin(blowup, rounds, root0)); }
int ms_fri_deep(ms_ctx* ctx, const uint64_t* z, uint64_t* Bv) { MS_ENTRY(ctx, B(ctx)->fri_deep(z, Bv)); }
int ms_fri_fold_commit(ms_ctx* ctx, const uint64_t* a, uint8_t root[32]) { MS_ENTRY(ctx, B(ctx)->fri_fold_commit(a, root)); }
int ms_fri_round_info(ms_ctx* ctx, int r, uint64_t* nc, uint64_t* D) { MS_ENTRY(ctx, B(ctx)->fri_round_info(r, nc, D)); }
int ms_fri_round_poly_read(ms_ctx* ctx, int r, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->fri_round_poly_read(r, out)); }
int ms_fri_round_codeword_read(ms_ctx* ctx, int r, uint64_t* out) { MS_ENTRY(ctx, B(ctx)->fri_round_codeword_read(r, out)); }
int ms_fri_query(ms_ctx* ctx, const uint64_t* betas, int nq) { MS_ENTRY(ctx, B(ctx)->fri_query(betas, nq, nullptr, 0, nullptr)); }
int ms_fri_query_into(ms_ctx* ctx, const uint64_t* betas, int nq, uint8_t* out, size_t cap, size_t* len) {
  if (!out && cap) return MS_ERR_ARG;
  static uint8_t probe;   // out == NULL, cap == 0: size query only (*len), nothing is computed
  MS_ENTRY(ctx, B(ctx)->fri_query(betas, nq, out ? out : &probe, out ? cap : 0, len));
}
size_t ms_fri_proof_size(const ms_ctx* ctx) { return ctx ? B(ctx)->fri_proof_size() : 0; }
int ms_fri_proof_read(ms_ctx* ctx, uint8_t* out) { MS_ENTRY(ctx, B(ctx)->fri_proof_read(out)); }
int ms_fri_proof_read_async(ms_ctx* ctx, uint8_t* out) { MS_ENTRY(ctx, B(ctx)->fri_proof_read_async(out)); }
int ms_fri_proof_wait(ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->fri_proof_wait()); }
int ms_io_engine(const ms_ctx* ctx) { return ctx ? B(ctx)->io_engine() : MS_ERR_ARG; }
int ms_merkle_commit(ms_ctx* ctx, const uint64_t* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, uint8_t* nodes_out, size_t cap, size_t* nn, uint8_t root[32]) {
  MS_ENTRY(ctx, B(ctx)->merkle_commit(leafs, leaf_num, ext, lpn, ic, nodes_out, cap, nn, root));
}
int ms_merkle_prove(ms_ctx* ctx, const uint64_t* leafs, size_t leaf_num, int ext, size_t lpn, const uint64_t* leaf, uint8_t* out, size_t cap, size_t* len) {
  MS_ENTRY(ctx, B(ctx)->merkle_prove(leafs, leaf_num, ext, lpn, leaf, out, cap, len));
}
int ms_ntt(ms_ctx* ctx, uint64_t* data, size_t n, size_t batch, int inverse) { MS_ENTRY(ctx, B(ctx)->ntt(data, n, batch, inverse)); }
int ms_coset_lde(ms_ctx* ctx, const uint64_t* c, size_t ncoef, size_t batch, uint64_t shift, uint64_t* out, size_t L) { MS_ENTRY(ctx, B(ctx)->coset_lde(c, ncoef, batch, shift, out, L)); }
int ms_bench_lde(ms_ctx* ctx, size_t blowup, uint64_t shift) { MS_ENTRY(ctx, B(ctx)->bench_lde(blowup, shift)); }
int ms_arith_selftest(ms_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { MS_ENTRY(ctx, B(ctx)->arith_selftest(op, a, b, out, n)); }
int ms_profile_begin(ms_ctx* ctx) { MS_ENTRY(ctx, B(ctx)->profile_begin()); }
int ms_profile_end(ms_ctx* ctx, char* json_out, size_t cap) { MS_ENTRY(ctx, B(ctx)->profile_end(json_out, cap)); }

}  // extern "C"
