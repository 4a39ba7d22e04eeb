// ctx.hpp — the prover session behind the C ABI (include/ministark.h): one class template per field, Ctx<F>, declared here and DEFINED in
// the translation units below (r05: one 2547-line unit until then; every kernel experiment paid its 2.5-minute rebuild).  A kernel template is
// instantiated - and its device code generated - in the unit that launches it, so the units also partition the device code:
//   session.cpp      ms_create's init, pools, I/O staging, profile, trace_commit / interpolate / polys_*           (transpose, narrow / widen)
//   io.cpp           the boundary's bulk transfers: trace in, FRI proof out, on SDMA engines or the HIP runtime; their failure handling
//   air_stages.cpp   constraint columns, LDE commit, mix, mix_cubic, DEEP-ALI evaluations                            (lincomb, mix, cubic, eval kernels)
//   ntt_plan.cpp     NTT plans and pass dispatch, coset evaluation, ms_ntt / ms_coset_lde                           (every NTT pass instance)
//   merkle_tree.cpp  MerkleTree::new, replicated and sharded, ms_merkle_commit                                      (SHA-256 leaf / inner kernels)
//   fri_commit.cpp   FRI commit phase: round commitments, DEEP evaluations, fold, suffix-Horner planning             (fold, scan, degree kernels)
//   fri_query.cpp    FRI query phase, proof read-back, ms_merkle_prove                                              (find-first, path, copy kernels)
//   shard.cpp        one proof over several ranks: exchange (RCCL or callback), ms_set_shard*
//   abi.cpp          the extern "C" entry points and their never-unwind guard
// Built with hipcc for gfx950 into libministark.so (csrc/Makefile).  There is no CPU fallback.
#pragma once
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

namespace msctx {

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
  // A copy engine did not finish a transfer within the time limit: it may still be reading or writing memory this context (or the caller) owns.  Every later
  // entry point returns MS_ERR_STATE; only ms_destroy is valid, and it waits for the transfer WITHOUT limit before it frees anything.
  bool poisoned = false;
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
  virtual int trace_upload_async(const u64* trace, size_t N, size_t w) = 0;
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

// runs `fn` when the scope ends, however it ends (early return, exception): state the stage functions set around a call that can fail
template <class Fn> struct ScopeExit { Fn fn; ~ScopeExit() { fn(); } };
template <class Fn> inline ScopeExit<Fn> scope_exit(Fn fn) { return ScopeExit<Fn>{fn}; }

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
  int shard_stub = 0;   // MS_SHARD_STUB=1 (the rank probe): ms_set_shard without a callback = a world without peers, every exchange a device copy of the rank's own payload
  u8* xs = nullptr; u8* xr = nullptr; size_t xcap = 0;   // caller-owned exchange buffers (device)
  ms_exchange_fn xfn = nullptr; void* xuser = nullptr;
  size_t shard_min_leaves = 32768;                        // MS_SHARD_MIN_LEAVES: smaller commitments stay replicated
  void* rccl_comm = nullptr; DevBuf rccl_send, rccl_recv;   // ms_set_shard_rccl: the library owns the communicator and the exchange buffers
  size_t rccl_max_piece = (size_t)1 << 30;                  // MS_RCCL_MAX_PIECE: most bytes of one ncclSend / ncclRecv / ncclAllGather (see exchange)
  u64 xstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // calls per op [0..3], bytes sent per op [4..7] (ms_shard_stats)
  int exchange(int op, size_t bytes);
  // one slice of a sliced digest all-to-all: peer r's piece at offset `off + r * stride`, `bytes` long, in both exchange buffers
  int shard_slices = 4; bool shard_slices_set = false; size_t shard_slice_min = 1024;   // (RCCL path: 1 unless MS_SHARD_SLICES is set - ADVICE r3: the two-stream overlap has never run on more than one GPU)
  size_t xl_off = 0, xl_stride = 0;
  msrt::Stream* comm_stream = nullptr; std::vector<msrt::Event*> ev_hash, ev_xchg;
  int shard_slice_layout(size_t* off, size_t* stride) override { if (!off || !stride) return fail(MS_ERR_ARG, "shard_slice_layout"); *off = xl_off; *stride = xl_stride; return MS_OK; }
  int exchange_slice(size_t off, size_t stride, size_t bytes, int sl, int S);
  void drop_rccl();
  void unshard() { sh_on = false; sh_rank = 0; sh_world = 1; xs = xr = nullptr; xcap = 0; xfn = nullptr; xuser = nullptr; have_lde = false; nrounds_done = 0; blob_size = 0; }
  int set_shard_rccl(int rank, int world, const u8* unique_id, size_t cap) override;
  // The four collectives on a one-rank communicator (send/recv to self, all-gather, both all-reduces) with known payloads:
  // checks the run-time binding of librccl.so (symbols, calling convention of the by-value ncclUniqueId, datatype / op enums)
  // and the stream ordering on a box with a single GPU.
  int rccl_selftest() override;
  int shard_proof_on_root(int on) override { proof_root_only = on ? 1 : 0; return MS_OK; }
  int shard_proof_is_elsewhere() const override { return (proof_root_only && sh_world > 1 && sh_rank != 0 && nrounds_done == fri_rounds && fri_rounds) ? 1 : 0; }
  int shard_round_is_distributed(int r) override { return (r >= 0 && (size_t)r < nrounds_done && rounds[r]->dist) ? 1 : 0; }
  int shard_stats(u64* out) override { if (!out) return fail(MS_ERR_ARG, "shard_stats"); memcpy(out, xstat, sizeof xstat); return MS_OK; }
  bool shardable(size_t leaf_groups) const { return sh_on && leaf_groups >= shard_min_leaves && leaf_groups >= (size_t)sh_world * (size_t)sh_world; }
  int set_shard(int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user) override;

  int fail_rt(int e, const char* what) { err = std::string("runtime error ") + std::to_string(e) + " in " + what + ": " + msrt::last_error_string(); return MS_ERR_HIP; }
  int fail(int code, const char* msg) { err = msg; return code; }

  // ---- optional per-kernel timing with HIP events on the launching stream (bench.py roofline leg)
  enum { K_NTT_PASS, K_SCALE_POW, K_LEAF_HASH, K_INNER_HASH, K_TRANSPOSE, K_IO, K_LINCOMB, K_MIX, K_EVAL, K_EVAL_REDUCE, K_FOLD,
         K_SUFFIX_HORNER, K_DEGREE, K_FIND_FIRST, K_PATH, K_QUERY_POINTS, K_FRI_TAIL, K_COUNT };
  struct ProfRec { int kid, sub; msrt::Event* a; msrt::Event* b; double bytes; bool part; };
  // sharded proofs: launches inside a PartScope work on this rank's PART of the proof (1 / world of it); everything else is replicated on every rank.
  // ms_profile_end reports both sums: the replicated one bounds the strong scaling (bench.py: sharded.replicated_ms_estimate)
  int part_depth = 0;
  struct PartScope { Ctx* c; explicit PartScope(Ctx* c_) : c(c_) { c->part_depth++; } ~PartScope() { c->part_depth--; } };
  struct PartScopeIf { Ctx* c; bool on; PartScopeIf(Ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->part_depth++; } ~PartScopeIf() { if (on) c->part_depth--; } };
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
  // MS_FLAG_LATENCY: the launch that ends a stage raises a sequence number in page-locked memory behind its results (msrt::HostFlag, rt.hpp) and the host polls that
  // word instead of synchronising with the stream (sync_results): 4-5 us less per host round trip, 44 of them per proof.  arm_flag() is for the Params of the NEXT
  // launch only: run / run_coop move the number from "pending" to "armed" and every other launch disarms, so a stale number never stands for work enqueued behind it;
  // stages that enqueue copies behind their last launch keep the stream synchronisation.
  int poll_sync = 0;   // MS_FLAG_LATENCY at ms_create (MS_SYNC_POLL=0/1 overrides: A/B)
  unsigned long long seq_no = 0, seq_pending = 0, seq_armed = 0;
  unsigned long long* host_seq() const { return reinterpret_cast<unsigned long long*>(reinterpret_cast<u8*>(pinned) + pinned_cap); }   // (64 bytes behind the staging area)
  msrt::HostFlag arm_flag() { if (!poll_sync || prof_on) return msrt::HostFlag{nullptr, 0}; seq_pending = ++seq_no; return msrt::HostFlag{host_seq(), seq_pending}; }
  void launched() { seq_armed = seq_pending; seq_pending = 0; }
  bool arm_next_eval = false;   // eval_views: the next evaluation is its caller's last (its launch carries the flag)
  const unsigned long long* fwd_next_eval = nullptr;   // ... and takes this device word along to host_aux2()
  unsigned long long* host_aux2() const { return host_seq() + 1; }
  // trimmed length of the validity polynomial: a device word since ms_mix (validity_len_dev), on the host once ms_eval_ext has brought it along (validity_len_host)
  unsigned long long* validity_len_dev = nullptr; bool validity_len_host = false; size_t validity_ncoef = 0;
  DevBuf d_evdone;              // + 128: the validity polynomial's length word (ms_mix)
  int ensure_evdone() { if (!d_evdone.p) { if (d_evdone.ensure(256)) return fail(MS_ERR_NOMEM, "length word"); CK(msrt::memset_dev(d_evdone.p, 0, 256, stream)); } return 0; }
  int sync_results() { const unsigned long long a = seq_armed; seq_armed = 0; return a ? msrt::sync_flag(stream, host_seq(), a) : msrt::sync(stream); }
  template <class K> int run_coop(int kid, unsigned gx, int threads, size_t lds, const typename K::Params& p, unsigned gy = 1) {
    launched();
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
    launched();
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
  int profile_end(char* out, size_t cap) override;

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
  int ntt_v2 = 1, ntt_v2_regpass = 1;   // MS_NTT_V2=0: round-1 kernels only (tests, A/B); MS_NTT_V2_REGPASS: the register-only last pass of 2^21..2^25-point transforms (0: off; 2: tests, on 2^7-row tiles)
  int ntt_fast = 1;            // MS_NTT_FAST=0: no compile-time specialised round-1 tiles (tests)

  // log_pad: the input is zero beyond n >> log_pad
  int get_plan(int log_n, int log_pad, bool inverse, Plan** out);

  template <bool INV, int K, int TH>
  int launch_fast(const msntt::PassParams<F>& pp, size_t tiles, size_t batch);
  // PassKernel2 instance for a pass of 2^K rows x 2^LC columns in MODE (ntt.hpp): tile shape -> threads, sub-rounds, arithmetic class
  template <class KK> int launch_v2i(const msntt::PassParams<F>& pp, unsigned grid, int nsub, int mode, int k, int lc);
  template <bool INV, int K, int LC, int MODE>
  int launch_v2m(const msntt::PassParams<F>& pp, size_t tiles, size_t batch);
  // persistent grid of the cooperative pass kernels: two workgroups per CU (their 72-80 KiB of LDS), a multiple of 8 (XCD-aware tile walk)
  int ntt_coop_wgs = 512, ntt_share = 1;   // MS_NTT_SHARE=0: never the shared-table instance (A/B)
  unsigned coop_grid(size_t work_items, int per_cu = 2) const { const size_t g = (size_t)ntt_coop_wgs * per_cu / 2; return (unsigned)(work_items < g ? work_items : g); }
  template <bool INV, int K, int LC>
  int launch_v2(const msntt::PassParams<F>& pp, size_t tiles, size_t batch);
  template <bool INV, int LC>
  int launch_v2k(const msntt::PassParams<F>& pp, size_t tiles, size_t batch);
  template <bool INV, int K>
  int launch_reg(const msntt::PassParams<F>& pp, size_t batch);
  template <bool INV>
  int launch_pass(const msntt::PassParams<F>& pp, size_t tiles, size_t batch, bool v2 = false, bool regp = false);

  // batch transforms of size 2^log_n: src (n_in valid elements per entry, zero padded) -> dst
  int ntt_run(int log_n, bool inverse, const T* src, size_t src_bstride, size_t n_in, T* dst, size_t dst_bstride, size_t batch);
  int scale_pow(const T* src, size_t src_bstride, T* dst, size_t dst_bstride, size_t n, T s, size_t batch);

  // ------------------------------------------------------------------ pool of zeroed device words
  // Counters that kernels bump (deferred-block lists, the degree result) must start at zero.  One memset clears a 256 KiB
  // pool; zero_alloc hands out fresh pieces of it and clears it again only when it runs out (stream order keeps earlier
  // users ahead of the clear) — a proof needs ~50 such counters, i.e. one memset instead of ~50 four-microsecond fills.
  // page-locked staging for the per-proof job tables (r03): uploads out of it are plain DMA, not the runtime's pageable-memory path (which pins or stages the
  // caller's pages on the fly), and need no synchronisation of their own - the area is rewritten by the NEXT proof's same stage, behind that stage's final
  // stream synchronisation
  void* h_tabs = nullptr; size_t h_tabs_cap = 0;
  int tabs_host(size_t bytes, u8** out);
  DevBuf d_zero; size_t zero_used = 0, zero_cap = 0;
  int zero_alloc(size_t bytes, void** out);

  // ------------------------------------------------------------------ Merkle
  // sharded (ms_set_shard): this rank holds the subtree over leaf groups [rank*Mloc, (rank+1)*Mloc) followed by the replicated top
  // (world subtree roots and the levels above); local_nodes = nodes held here, root last in both cases
  struct TreeShape { size_t leaf_num = 0, lpn = 0, ic = 0, levels = 0, nodes = 0, local_nodes = 0, Mloc = 0; bool sharded = false; };
  // src/merkle.rs:89-118 (shape checks and node count)
  int tree_shape(size_t leaf_num, size_t lpn, size_t ic, TreeShape* ts);
  // leaf-group digests of `ngroups` groups into `out`: LeafHashKernel + the compacted pad-only blocks it deferred
  template <int EL>
  int leaf_hash(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, size_t lpn, size_t ngroups, u32* out,
                size_t g_first = 0, u32 run_len = 0, u32 run_stride = 0, const msmerkle::LinColSpec* lin = nullptr, size_t out_g0 = 0);
  template <int EL>
  int tree_build(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, const TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin = nullptr);
  // inner levels above `nchildren` digests at nodes[0..): level-major, root last (merkle.rs:131-140)
  // `final_levels`: these levels end in the tree's root, which the last launch also stores to the page-locked slot host_root()
  // (root_on_host: read_root / read_degree_and_root then need no copy launch, only the stream synchronisation they do anyway)
  u32* host_root() const { return reinterpret_cast<u32*>(reinterpret_cast<u8*>(pinned) + 256); }
  bool root_on_host = false;
  unsigned long long* pending_aux = nullptr; bool aux_on_host = false;   // a device word the tree's final launch forwards to pinned[0] (the degree result)
  int inner_levels(u32* nodes, size_t nchildren, size_t ic, bool final_levels = true, u8* rec_out = nullptr);
  // Sharded MerkleTree::new over a binary tree of M = leaf_num/lpn leaf groups, of which this rank hashes the groups
  // j = rank + W*i found at local group index i of the view (base, strides): digest all-to-all, subtree, root all-gather, top.
  template <int EL>
  int tree_build_sharded(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin = nullptr);
  // the subtree over this rank's Mloc contiguous leaf digests (at nodes[0..)), the all-gather of the W subtree roots, the replicated top.
  // shard_aux (a device word, optional): rides on the root all-gather; every rank gets the maximum over the ranks back in the same word.
  unsigned long long* shard_aux = nullptr;
  int finish_sharded_tree(TreeShape& ts, DevBuf& nodes, size_t Mloc);
  // MerkleTree::new over data EVERY rank holds (the raw trace): rank k hashes the contiguous leaf groups [k*M/W, (k+1)*M/W) - no digest exchange at all -
  // builds that subtree, and the ranks all-gather the W subtree roots
  template <int EL>
  int tree_build_sharded_contiguous(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes);
  // root of the tree built LAST on this context (every caller reads it right behind the build)
  int read_root(const DevBuf& nodes, const TreeShape& ts, u8* root);

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
  int materialize(int i);
  bool have_trace = false, have_polys = false, have_lde = false, have_validity = false;
  DevBuf d_trace[2], d_polys, d_coef, d_lde, d_trace_nodes, d_lde_nodes, d_io, d_partials, d_small;
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

  int ensure_polys(size_t count);

  int init(int dev, u32 flags);
  ~Ctx() {
    for (auto& kv : plans) { kv.second->tw_lo.release(); kv.second->tw_hi.release(); kv.second->vtw.release(); kv.second->w0.release(); for (auto& b : kv.second->w_r) b.release(); delete kv.second; }
    drop_rccl();
    // transfers still in flight read or write memory freed below: wait for them, however long it takes (a poisoned context got here exactly because one did not finish in time)
    if (sdma_pending) msrt::Sdma::get().wait(sdma_sig, -1.0);
    else if (copy_pending) msrt::event_sync(ev_copy);
    if (up.pending) msrt::Sdma::get().wait(sdma_up_sig, -1.0);
    if (sdma_state == 1) { msrt::Sdma::get().signal_destroy(sdma_sig); msrt::Sdma::get().signal_destroy(sdma_up_sig); }
    for (msrt::Event* e : ev_hash) msrt::event_destroy(e);
    for (msrt::Event* e : ev_xchg) msrt::event_destroy(e);
    if (comm_stream) msrt::stream_destroy(comm_stream);
    if (side_stream) { msrt::sync(side_stream); msrt::stream_destroy(side_stream); }
    if (ev_side) msrt::event_destroy(ev_side);
    if (ev_blob) msrt::event_destroy(ev_blob);
    if (ev_copy) msrt::event_destroy(ev_copy);
    if (copy_stream) msrt::stream_destroy(copy_stream);
    for (Round* r : rounds) { r->poly.release(); r->cw.release(); r->nodes.release(); delete r; }
    DevBuf* bufs[] = {&ntt_scratch, &d_trace[0], &d_trace[1], &d_polys, &d_coef, &d_lde, &d_trace_nodes, &d_lde_nodes, &d_io, &d_partials, &d_small, &d_folded, &d_sh, &d_blob, &d_tabs, &d_targets, &d_idx, &d_deg, &d_ovf, &d_zero, &d_lin, &d_cubic, &d_carry, &d_lq, &d_pack, &d_fullpoly, &d_evdone};
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
  int upload_narrow(const u64* host, size_t n, T* dst);
  int transpose_in(const u64* src, T* dst, size_t rows, size_t cols, T rinv, int mont, u32* bad);
  int download_widen(const T* src, size_t n, size_t limb_stride, u32 e, u64* host);
  static bool canonical(const u64* v, size_t n) { for (size_t i = 0; i < n; i++) if (v[i] >= F::P) return false; return true; }

  // ------------------------------------------------------------------ starks.rs:68-73
  int trace_commit(const u64* trace, bool on_device, size_t N_, size_t w_, size_t lpn, u8* root) override;
  // ------------------------------------------------------------------ air.rs:147-160
  int interpolate() override;
  int polys_lincomb(const u64* s, const int* idx, int k) override;
  int polys_append(const u64* coeffs, size_t n) override;
  int polys_count() const override { return npolys; }
  int poly_read(int i, u64* out) override;

  // ------------------------------------------------------------------ starks.rs:80-95
  // one lincomb launch chain: dst = sum_t s[t] * base[idx[t]] over n elements (columns `stride` apart)
  int lincomb_into(const T* base, size_t stride, size_t n, const u64* sc, const int* idx, int k, int self_index, T* dst);
  // The LDE columns of all polynomials with linear provenance (columns `stride` apart, n elements each).  Consecutive ones whose sources
  // are all transformed columns (no linear column among them) and fit LincombMultiKernel (<= 4 outputs over <= 8 distinct sources) share
  // one sweep; anything else goes through lincomb_into one by one.
  int lincomb_linear_columns(T* base, size_t stride, size_t n);
  // starks.rs:80-91.  The coset evaluation is linear, so a polynomial that ms_polys_lincomb defined as
  // sum_t s_t * P_idx[t] has LDE column sum_t s_t * LDE(P_idx[t]): only polynomials without such provenance
  // (the trace columns, ms_polys_append uploads) go through the NTT.
  int lde_compute(size_t blowup_, u64 shift);
  // The linear LDE columns: VIRTUAL when every one of them is a combination of at most LIN_MAXT stored (transformed) columns - the leaf-hash kernel
  // then evaluates them row by row while it hashes (they are never read again: the query phase opens FRI codewords only) and ms_lde_read materialises
  // them on demand; otherwise written out by the lincomb kernels.  MS_LDE_VIRTUAL: -1 (default) for AIRs of >= 16 polynomials, 0 never, 1 always.
  // Measured r03: wide AIR (64 linear columns of 2^25 rows) LDE commit 105.5 -> 103.6 ms - the 20 ms of lincomb launches go away, but the leaf kernel reads
  // its source columns again (out of L2 by then) and multiplies; Fibonacci AIR with 8 proofs in flight 239 -> 233 proofs/s (the prover is bound by its VALU
  // instruction count, and the leaf kernel's grew): so narrow AIRs keep the lincomb kernels.
  int lde_virtual = -1; bool lde_cols_virtual = false; size_t lde_col_stride = 0, lde_col_len = 0;
  DevBuf d_lin;
  int finish_linear_columns(size_t stride, size_t n);
  const msmerkle::LinColSpec* lde_lin() const { return lde_cols_virtual ? d_lin.as<msmerkle::LinColSpec>() : nullptr; }
  // Evaluations of `batch` polynomials (ncoef coefficients each) on this rank's share of the size-2^log_D domain shift*<w_D>:
  // the rows g*(rank + W*i) + t (t < g, i < m = D/(g*W)) — g cosets of <w_m> — land at dst[b*dst_bstride + t*m + i].
  int coset_eval(const T* coef, size_t coef_bstride, size_t ncoef, int log_D, T shift, size_t g, T* dst, size_t dst_bstride, size_t batch);
  // lde_compute for a sharded proof: column i of the local LDE (rows rank + W*j) at d_lde + i*m, m = L/W
  int lde_compute_sharded(size_t blowup_, u64 shift);
  int lde_commit(size_t blowup_, u64 shift, size_t lpn, u8* root) override;
  int bench_lde(size_t blowup_, u64 shift) override;
  int arith_selftest(int op, const u64* a, const u64* b, u64* out, size_t n) override;
  int lde_read(u64* out) override;
  // ------------------------------------------------------------------ starks.rs:108-119
  int mix(u64 r) override;
  // build-defined degree-3 composition with the true quotient (include/ministark.h; kernel: mspoly::CubicComposeKernel)
  size_t validity_len = 0; u64 lde_shift = 0; DevBuf d_cubic;
  size_t validity_len_() const override { return have_validity ? validity_len : 0; }
  int mix_cubic(u64 r, const int* spec, const u64* sc, int ncons) override;
  // trimmed length of a base-field coefficient vector
  int degree_launch1(const T* poly, size_t n, unsigned long long** dres_out);
  int validity_read(u64* out) override;

  // evaluate `npoly` polynomials (views) at ext point z into dst as [npoly][E] T
  template <int EC>
  int eval_views(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst);
  template <int EC, int ITEMS>
  int eval_views_i(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst, size_t maxc);
  static bool load_ext(const u64* v, XE* out) { for (int l = 0; l < E; l++) { if (v[l] >= F::P) return false; out->c[l] = F::from_u64(v[l]); } return true; }

  // sharded proof: out[i] = sum_r part_r[i] * zstep^r for n extension elements of the all-gathered partials (rank r's payload rank_stride limbs apart, the elements from `off` on)
  int shard_combine_launch(size_t off, size_t rank_stride, u32 n, const XE& zstep, T* out);
  // DEEP-ALI evaluations by coefficient range (r04): rank k evaluates the coefficients [k*Sx, (k+1)*Sx) of every polynomial at every point, ONE all-gather of the partial
  // sums, and every rank combines them with z^Sx
  bool dist_eval() const { return sh_on && shard_dist && N >= shard_min_leaves && N >= (size_t)sh_world * 2; }
  // the polynomials the DEEP-ALI kernels evaluate: those whose coefficients exist, then the validity polynomial (index npolys); the lazily defined ones get their
  // values on the host as the combination their provenance names (eval_finish)
  std::vector<int> eval_set() const { std::vector<int> ev; for (int i = 0; i < npolys; i++) if (poly_mat[i]) ev.push_back(i); ev.push_back(npolys); return ev; }
  // page-locked results [q][ev.size()][E] -> out [q][npolys + 1][E] (u64)
  int eval_finish(int q, const std::vector<int>& ev, u64* out);
  int eval_ext_sharded(const u64* z, int q, u64* out);

  // ------------------------------------------------------------------ starks.rs:124-151
  int eval_ext(const u64* z, int q, u64* out) override;

  // ------------------------------------------------------------------ FRI rounds (fri.rs:314-352)
  Round* round_slot(size_t i) { while (rounds.size() <= i) rounds.push_back(new Round()); return rounds[i]; }
  // codeword + tree of rounds[i] from its coefficient limbs (ncoef_in valid coefficients)
  // `nonzero_limbs`: limbs >= this are identically zero (round 0: extend_poly embeds base coefficients), so their
  // transform is all zeros and is not computed
  // `prev` != nullptr: the codeword is folded out of prev's codeword in the evaluation domain (FriFoldEvalKernel) instead of
  // transforming the round polynomial — same values, a quarter of the arithmetic
  int round_commit(Round* r, size_t ncoef_in, int nonzero_limbs = E, const Round* prev = nullptr, const XE* alpha = nullptr);
  // trimmed length of a round polynomial (DegreeKernel) into a zeroed device word
  int degree_launch(const T* poly, size_t limb_stride, size_t n, unsigned long long** dres_out);
  int read_degree_and_root(const T* poly, size_t limb_stride, size_t n, Round* r, size_t* ncoef, u8* root);
  // fri.rs:73-82
  int fri_begin(size_t blowup_, size_t nrounds, u8* root0) override;
  // fri.rs:89-94
  int fri_deep(const u64* z, u64* B) override;

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
                 const T* ext_carry = nullptr, T* top_agg = nullptr, bool out_h0 = false);
  // z^(m - P) = (1/z)^(P - m): moves a top-level carry-in from the padded position P to the job's end m (0 for z = 0: nothing then carries over)
  static XE carry_scale(const XE& z, size_t m, size_t P) { return e_pow<F, E>(e_inv<F>(z), (u64)(P - m)); }
  int sh_launch_inline(const SHJ& j, int final_mode);
  int sh_launch_table(const SHJ* d_jobs, size_t njobs, size_t maxnb, int final_mode);
  int shard_carry_table(const mspoly::CarryJob<F, E>* d_jobs, size_t njobs);
  // one logical job, launched level by level with the job inline in the kernel arguments
  int suffix_horner(const T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const XE& z,
                    T* out, size_t out_limb_stride, size_t out_off, size_t out_stride, T* h0);

  // all-gather of a distributed round polynomial's parts into one replicated vector (dst: E limbs, dst_stride apart, `count` coefficients)
  int gather_poly(const T* local, size_t S, size_t count, T* dst, size_t dst_stride);
  // fri.rs:96-101 on a DISTRIBUTED round polynomial (r04): rank k folds its own coefficient pairs, runs the suffix Horner of (folded - B(alpha)) / (x - z) over its
  // own range with the sum over the higher ranks as carry-in (one all-gather of [first folded element | aggregate] per rank, ShardCarryKernel), and ends up with
  // its range [k*S', (k+1)*S') of the quotient = the next round polynomial, S' = S/2.  If the next round is too small to stay distributed the parts are
  // all-gathered into a replicated polynomial.  H_j for j in (lo, hi) comes from the rank's own job over f[lo+1 .. hi); H_hi = q_(hi-1) IS the carry-in.
  int fold_dist(Round* pr, Round* nr, const XE& a, size_t* nq_coef_out);
  // fri.rs:96-109
  int fri_fold_commit(const u64* alpha, u8* root) override;
  // MS_FRI_TAIL_MAX: rounds folding a domain of at most this many points run as ONE launch (fri_tail.hpp; 0: never).  Same-box A/B: DESIGN.md
  size_t fri_tail_max = (size_t)1 << 15;   // (8192 / 32768 / 131072: within noise of each other alone and with eight in flight, profiles/r05_fri_tail_ab.log; 2^15 = every replicated round of a sharded proof at the default MS_SHARD_MIN_LEAVES)
  // MS_FLAG_LATENCY (r05): the coefficient side of a round's ms_fri_fold_commit (fold, DEEP quotient scan, trimmed length: ~55 us of small launches) on a SIDE stream
  // while the evaluation side (pointwise codeword, leaf hashing, tree) runs on the context's stream - the two are independent (the codeword never reads the
  // quotient); the launch that forwards root and length word to the host waits for the side stream's event.  The fused tail rounds do the same inside one kernel.
  int fri_overlap = 0;   // MS_FLAG_LATENCY at ms_create (MS_FRI_OVERLAP=0/1 overrides: A/B)
  msrt::Stream* side_stream = nullptr; msrt::Event* ev_side = nullptr; bool side_pending = false;
  struct StreamScope { Ctx* c; msrt::Stream* keep; StreamScope(Ctx* c_, msrt::Stream* s) : c(c_), keep(c_->stream) { c->stream = s; } ~StreamScope() { c->stream = keep; } };
  int join_side() { if (side_pending) { side_pending = false; CK(msrt::stream_wait_event(stream, ev_side)); } return 0; }   // the context's stream goes on only behind the side stream's work
  int fri_tail_round(Round* pr, Round* nr, const XE& a, bool* done);
  int fri_round_info(int r, u64* ncoef, u64* D) override;
  int fri_round_poly_read(int r, u64* out) override;
  int fri_round_codeword_read(int r, u64* out) override;

  // ------------------------------------------------------------------ fri.rs:115-189
  // The whole query phase is a fixed handful of batched launches, whatever the number of
  // rounds and queries: every per-(window, query) step is a job in a device-side table.
  // ext_out != nullptr (ms_fri_query_into): the query-phase kernels write the MSFP blob straight into the caller's buffer - page-locked host
  // memory (ms_pinned_alloc; hipHostMalloc memory is mapped into the device's address space) or device memory - instead of d_blob:
  // no read-back copy afterwards (r02: the runtime executed most of the 64 MiB read-back as shader copies, -15 % on the I/O-inclusive rate)
  bool blob_external = false;
  int fri_query(const u64* betas, int nq, u8* ext_out, size_t ext_cap, size_t* ext_len) override;
  // what the steps of one query phase share (fri_query.cpp): blob layout, host-side job tables, their offsets in the one uploaded table area
  struct QueryPlan {
    typedef mspoly::CarryJob<F, E> CJ; typedef mspoly::FindJob<F, E> FJ; typedef msmerkle::PathJob<F, E> PJ; typedef msmerkle::ShardPathJob<F, E> SPJ;
    size_t W = 0; int nq = 0; size_t pos = 0; u8* blob = nullptr;
    T* d_h0 = nullptr; T* d_tg = nullptr; unsigned long long* d_ix = nullptr;          // H_0 outputs of the scans | find-first targets | found leaf indices
    std::vector<size_t> rec_off, path_off; std::vector<u64> qlen; std::vector<T> x1h;
    std::vector<std::vector<SHJ>> tables; std::vector<int> table_mode, table_part; size_t n_agg_tables = 0;   // launch order: every aggregate launch, then - behind the carry exchange - every final launch
    std::vector<CJ> carry_jobs; bool any_dist = false;                                 // sharded proofs: carry-ins of the ranks' scan jobs
    std::vector<msmerkle::CopyJob> unp; std::vector<size_t> unp_first, unp_cnt, chunk_c0, chunk_len;   // ... and the slices' way into the blob, chunk by chunk
    std::vector<FJ> fjobs; std::vector<PJ> pjobs; std::vector<SPJ> sjobs; std::vector<msmerkle::CopyJob> cjobs; size_t stage_bytes = 0;
    std::vector<size_t> toff; size_t off_f = 0, off_p = 0, off_sp = 0, off_cj = 0, off_rec = 0, off_ql = 0, off_x1 = 0, off_cr = 0, off_un = 0; const u8* dt = nullptr;
  };
  int query_layout(QueryPlan& q, int nq);
  int query_build(QueryPlan& q, const u64* betas);
  int query_scans(QueryPlan& q);
  int query_shard_assembly(QueryPlan& q);
  int query_openings(QueryPlan& q);
  size_t fri_proof_size() const override { return blob_size; }
  int fri_proof_read(u8* out) override;
  // The same copy on the context's COPY stream, ordered behind the query phase by an event: the call returns at once and the next proof's
  // stages run while the ~64 MiB travel; the next ms_fri_query waits (on the device, not on the host) for the copy before it rewrites the blob.
  msrt::Stream* copy_stream = nullptr; msrt::Event* ev_blob = nullptr; msrt::Event* ev_copy = nullptr; bool copy_pending = false;
  // r04: the read-back as an explicit SDMA copy through the HSA runtime (rt.hpp, msrt::Sdma) - queued on a copy engine whatever the engines' load, never a
  // blit kernel; completion is an HSA signal the host waits on (ms_fri_proof_wait, or the next ms_fri_query before it rewrites the blob).  MS_READBACK=hip keeps
  // hipMemcpyAsync on the copy stream (A/B); io_engine() reports which path the last read-back took.
  int readback_sdma = 2, sdma_gpu = -1, sdma_state = 0 /* 0 untried, 1 bound, 2 unavailable */, last_io_engine = 0;
  msrt::Sdma::Signal sdma_sig{0}, sdma_up_sig{0}; bool sdma_pending = false; int upload_sdma = 1;
  u8* readback_dst = nullptr; size_t readback_bytes = 0;   // destination and size of the read-back in flight (a FAILED engine copy is redone through the HIP runtime)
  double sdma_timeout_s = 20.0;        // MS_SDMA_TIMEOUT_S: how long a stage waits for a copy engine before it poisons the context
  bool sdma_ready();
  int poison(const char* what);        // -> MS_ERR_HIP now, MS_ERR_STATE from every later entry point
  int io_engine() const override { return last_io_engine; }
  int fri_proof_read_async(u8* out) override;
  int fri_proof_wait() override;
  // ---- the trace's way in (r05): two device buffers; ms_trace_upload_async queues the SDMA copy of the NEXT proof's trace into the one the current proof does not
  // use, ms_trace_commit of that trace finds it there (it waits for the signal - long since 0 - instead of for the whole transfer)
  struct Upload { const u64* src = nullptr; size_t N = 0, w = 0; int slot = 0; bool pending = false; } up;
  int trace_slot = 0;                  // the buffer the last ms_trace_commit read
  int trace_upload_async(const u64* trace, size_t N_, size_t w_) override;
  int trace_to_device(const u64* trace, const u64** dsrc);
  int upload_wait(const char* what);   // completes the upload in flight: 0 arrived, MS_ERR_HIP poisoned (timeout), 1 = the engine reported failure (the caller copies again)

  // ------------------------------------------------------------------ standalone entry points
  int merkle_commit(const u64* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, u8* nodes_out, size_t cap, size_t* nn, u8* root) override;
  // src/merkle.rs:272-288 on a standalone binary tree: leaves de-interleaved to SoA limbs, then the same
  // LeafHash / InnerHash / FindFirst / MerklePath kernels the FRI query phase uses
  template <int EL>
  int merkle_prove_t(const u64* leafs, size_t leaf_num, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len);
  int merkle_prove(const u64* leafs, size_t leaf_num, int ext, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) override;
  int ntt(u64* data, size_t n, size_t batch, int inverse) override;
  int coset_lde(const u64* coeffs, size_t ncoef, size_t batch, u64 shift, u64* out, size_t L_) override;
};

inline CtxBase* B(ms_ctx* c) { return reinterpret_cast<CtxBase*>(c); }
inline const CtxBase* B(const ms_ctx* c) { return reinterpret_cast<const CtxBase*>(c); }

}  // namespace msctx
