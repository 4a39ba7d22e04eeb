// abi.cpp — the extern "C" entry points of include/ministark.h: one thin guarded call per stage function.
#include "ctx.hpp"

using namespace msctx;

// ---- "never unwind".  The stage functions build std::vector job tables, std::string error texts and a std::map of NTT plans, and ms_create
// constructs the context: every entry point that can reach an allocation runs inside this guard - std::bad_alloc becomes MS_ERR_NOMEM, anything else
// MS_ERR_HIP - so that no C++ exception ever crosses the C ABI into a Rust / C caller (undefined behaviour there).  The error text is set without
// allocating (literals short enough for the small-string buffer).  tests/test_emu_parity.py::test_c_abi_never_unwinds drives a whole proof with the N-th
// allocation of the emulation build failing, for every N.
static int on_exception(const ms_ctx* ctx, bool oom) noexcept {
  try { if (ctx) const_cast<CtxBase*>(B(ctx))->err = oom ? "out of memory" : "C++ exception"; } catch (...) {}
  return oom ? MS_ERR_NOMEM : MS_ERR_HIP;
}
#define MS_ENTRY(ctx, expr) do { if (!(ctx)) return MS_ERR_ARG; if (B(ctx)->poisoned) return MS_ERR_STATE; try { B(ctx)->bind_device(); return (expr); } \
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
int ms_trace_upload_async(ms_ctx* ctx, const uint64_t* t, size_t N, size_t w) { MS_ENTRY(ctx, B(ctx)->trace_upload_async(t, N, w)); }
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
const char* ms_io_runtime_path(void) { return msrt::Sdma::get().runtime_path(); }
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
