// session.cpp — context life cycle, device pools, I/O staging, per-kernel profile, and the first stages of Stark::prove: trace commitment (starks.rs:68-73),
// interpolation (air.rs:147-160), constraint polynomials (air.rs:127-144).
#include "ctx.hpp"

namespace msctx {

template <class F>
int Ctx<F>::profile_end(char* out, size_t cap) {
  static const char* names[K_COUNT] = {"ntt_pass", "scale_pow", "leaf_hash", "inner_hash", "transpose_in", "io_copy", "lincomb", "mix", "eval", "eval_reduce",
                                       "fold", "suffix_horner", "degree", "find_first", "merkle_path", "query_points", "fri_tail"};
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

template <class F>
int Ctx<F>::tabs_host(size_t bytes, u8** out) {
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

template <class F>
int Ctx<F>::zero_alloc(size_t bytes, void** out) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (!d_zero.p) { if (d_zero.ensure(256 << 10)) return fail(MS_ERR_NOMEM, "zero pool"); zero_cap = 256 << 10; zero_used = zero_cap; }
  if (bytes > zero_cap) return fail(MS_ERR_NOMEM, "zero pool request too large");
  if (zero_used + bytes > zero_cap) { CK(msrt::memset_dev(d_zero.p, 0, zero_cap, stream)); zero_used = 0; }
  *out = d_zero.as<u8>() + zero_used; zero_used += bytes;
  return 0;
}

template <class F>
int Ctx<F>::materialize(int i) {
  if (poly_mat[i]) return 0;
  const Lin& li = poly_lin[i];
  for (int ix : li.idx) RQ(materialize(ix));
  RQ(lincomb_into(d_polys.as<T>(), N, N, li.s.data(), li.idx.data(), (int)li.idx.size(), i, d_polys.as<T>() + (size_t)i * N));
  poly_mat[i] = 1;
  return 0;
}

template <class F>
int Ctx<F>::ensure_polys(size_t count) {
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

template <class F>
int Ctx<F>::init(int dev, u32 flags) {
  device = dev; zae = (flags & MS_FLAG_ZERO_DISPLAY_EMPTY) ? 1 : 0; trace_mont = (flags & MS_FLAG_TRACE_MONT64) ? 1 : 0; fri_overlap = (flags & MS_FLAG_LATENCY) ? 1 : 0; poll_sync = fri_overlap;
  if (const char* e = getenv("MS_NTT_KMAX")) { int v = atoi(e); if (v >= 5 && v <= msntt::MAX_LOG_R) ntt_kmax = v; }
  if (const char* e = getenv("MS_NTT_FAST")) ntt_fast = atoi(e);
  if (const char* e = getenv("MS_NTT_V2")) ntt_v2 = atoi(e);
  if (const char* e = getenv("MS_NTT_V2_REGPASS")) ntt_v2_regpass = atoi(e);
  if (const char* e = getenv("MS_NTT_SHARE")) ntt_share = atoi(e);
  if (const char* e = getenv("MS_NTT_COOP_WGS")) { int v = atoi(e); if (v >= 8 && v <= 65536) ntt_coop_wgs = v & ~7; }
  if (const char* e = getenv("MS_LDE_LINEAR")) lde_linear = atoi(e);
  if (const char* e = getenv("MS_LAZY_LINCOMB")) lazy_lin = atoi(e);
  if (const char* e = getenv("MS_LDE_MULTI")) lde_multi = atoi(e);
  if (const char* e = getenv("MS_LDE_VIRTUAL")) lde_virtual = atoi(e);
  if (const char* e = getenv("MS_FRI_POINTWISE")) fri_pointwise = atoi(e);
  if (const char* e = getenv("MS_FOLD_SMALL_MAX")) fold_small_max = (size_t)atol(e);
  if (const char* e = getenv("MS_FRI_TAIL_MAX")) fri_tail_max = (size_t)atol(e);
  if (const char* e = getenv("MS_FRI_OVERLAP")) fri_overlap = atoi(e);
  if (const char* e = getenv("MS_SYNC_POLL")) poll_sync = atoi(e);
  if (const char* e = getenv("MS_EVAL_SMALL_MAX")) eval_small_max = (size_t)atol(e);
  if (const char* e = getenv("MS_TREE_SUBTREE_PARENTS")) subtree_parents = (size_t)atol(e);
  // the boundary's bulk copies (r04): page-locked trace in / FRI proof out on SDMA engines through the HSA runtime by default (measured with 8 provers in flight,
  // tools/io_probe3.py: resident 251 proofs/s; upload by hipMemcpyAsync 238, by SDMA 251; read-back by hipMemcpyAsync 217-223, by SDMA 242-248; both by SDMA 245.5 = 0.978)
  if (const char* e = getenv("MS_UPLOAD")) upload_sdma = !strcmp(e, "hip") ? 0 : 1;
  if (const char* e = getenv("MS_SDMA_TIMEOUT_S")) { const double v = atof(e); if (v > 0) sdma_timeout_s = v; }
  if (const char* e = getenv("MS_READBACK")) readback_sdma = !strcmp(e, "hip") ? 0 : (!strcmp(e, "sdma-async") ? 1 : 2);   // sdma-async: only ms_fri_proof_read_async on the engine
  if (const char* e = getenv("MS_LEAF_LAZY_MIN")) leaf_lazy_min = atoi(e);
  if (const char* e = getenv("MS_TREE_TOP")) { int v = atoi(e); if (v >= 1 && v <= 65536) tree_top_parents = v; }
  if (const char* e = getenv("MS_SHARD_MIN_LEAVES")) { long v = atol(e); if (v >= 1) shard_min_leaves = (size_t)v; }
  if (const char* e = getenv("MS_SHARD_DIST")) shard_dist = atoi(e);
  if (const char* e = getenv("MS_SHARD_WORLD1")) allow_w1 = atoi(e);
  if (const char* e = getenv("MS_SHARD_STUB")) shard_stub = atoi(e);
  if (const char* e = getenv("MS_SHARD_GATHER_CHUNK")) { long v = atol(e); if (v >= 64) shard_gather_chunk = (size_t)v & ~(size_t)63; }
  if (const char* e = getenv("MS_SHARD_SLICES")) { int v = atoi(e); if (v >= 1 && v <= 64) { shard_slices = v; shard_slices_set = true; } }
  if (const char* e = getenv("MS_SHARD_SLICE_MIN")) { long v = atol(e); if (v >= 1) shard_slice_min = (size_t)v; }
  if (const char* e = getenv("MS_RCCL_MAX_PIECE")) { long long v = atoll(e); if (v >= 64 && v <= ((long long)1 << 30)) rccl_max_piece = (size_t)v & ~(size_t)63; }
  CK(msrt::set_device(dev));
  CK(msrt::stream_create(&own_stream));
  stream = own_stream;
  pinned_cap = 1 << 16;
  CK(msrt::malloc_host_coherent(&pinned, pinned_cap + 64));   // (+ the polled sequence word, host_seq())
  *host_seq() = 0;
  if (d_small.ensure(4096)) return fail(MS_ERR_NOMEM, "small");
  return 0;
}

// staged copies between the u64 ABI and device storage
template <class F>
int Ctx<F>::upload_narrow(const u64* host, size_t n, T* dst) {
  if (d_io.ensure(n * 8)) return fail(MS_ERR_NOMEM, "io staging");
  CK(msrt::h2d(d_io.p, host, n * 8, stream));
  typename mspoly::NarrowKernel<F>::Params p{d_io.as<u64>(), dst, n};
  CK(run<mspoly::NarrowKernel<F>>(K_IO, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
  return 0;
}

template <class F>
int Ctx<F>::download_widen(const T* src, size_t n, size_t limb_stride, u32 e, u64* host) {
  if (n == 0) return 0;
  if (d_io.ensure(n * e * 8)) return fail(MS_ERR_NOMEM, "io staging");
  typename mspoly::WidenKernel<F>::Params p{src, d_io.as<u64>(), n, limb_stride, e};
  CK(run<mspoly::WidenKernel<F>>(K_IO, grid1(n * e, mspoly::THREADS), 1, mspoly::THREADS, 0, p));
  CK(msrt::d2h(host, d_io.p, n * e * 8, stream));
  CK(msrt::sync(stream));
  return 0;
}

// ------------------------------------------------------------------ starks.rs:68-73
// row-major u64 matrix (rows x cols, device) -> column-major columns of T, `rows` apart: the range check (bad: a zeroed device word, or null) and the
// Montgomery conversion of MS_FLAG_TRACE_MONT64 ride along
template <class F>
int Ctx<F>::transpose_in(const u64* src, T* dst, size_t rows, size_t cols, T rinv, int mont, u32* bad) {
  typename mspoly::TransposeInKernel<F>::Params tp{src, dst, rows, cols, rows, rinv, mont, bad};
  if (cols >= 16 && rows >= 64) {   // wide traces: 64 x 64 tiles through LDS (coalesced both ways)
    typedef mspoly::TransposeInTiledKernel<F> TK;
    CK(run<TK>(K_TRANSPOSE, (unsigned)(rows / TK::TILE), (unsigned)((cols + TK::TILE - 1) / TK::TILE), TK::THREADS, TK::lds_bytes(), tp));
  } else
  CK(run<mspoly::TransposeInKernel<F>>(K_TRANSPOSE, grid1(rows * cols, mspoly::THREADS), 1, mspoly::THREADS, 0, tp));
  return 0;
}

template <class F>
int Ctx<F>::trace_commit(const u64* trace, bool on_device, size_t N_, size_t w_, size_t lpn, u8* root) {
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
    RQ(trace_to_device(trace, &dsrc));   // io.cpp: an SDMA engine for page-locked sources (a prefetched trace is already there), the HIP runtime's copy otherwise
  }
  RQ(ensure_polys(w + 1));
  // 2^-64 mod p: arkworks stores Montgomery representatives (R = 2^64 for the one-limb Fp of both fields)
  const T rinv = f_inv<F>(F::from_u64((u64)(((unsigned __int128)1 << 64) % F::P)));
  void* badw;
  RQ(zero_alloc(8, &badw));  // device input cannot be range-checked on the host: the kernel flags elements >= p
  RQ(transpose_in(dsrc, d_polys.as<T>(), N, w, rinv, trace_mont, reinterpret_cast<u32*>(badw)));
  // the flag word rides to the host on the tree's last launch, like a round's length word (r05: the 4-byte copy behind the tree cost ~30 us of launch latency per proof)
  pending_aux = reinterpret_cast<unsigned long long*>(badw); aux_on_host = false;
  auto clear = scope_exit([this] { pending_aux = nullptr; });
  // element f of trace.get_data() = column f % w, row f / w of the column-major copy
  // one proof over several ranks: every rank holds the whole trace, so rank k hashes the contiguous leaf groups [k*M/W, (k+1)*M/W) and only the W subtree roots travel (r04)
  if (sh_on && shard_dist && shardable(ts.leaf_num / ts.lpn)) RQ((tree_build_sharded_contiguous<1>(d_polys.as<T>(), N, 1, 0, (u32)w, ts, d_trace_nodes)));
  else
  RQ((tree_build<1>(d_polys.as<T>(), N, 1, 0, (u32)w, ts, d_trace_nodes)));
  trace_ts = ts;
  if (!aux_on_host) { pending_aux = nullptr; seq_armed = 0; CK(msrt::d2h(pinned, badw, 8, stream)); if (root_on_host) CK(msrt::sync(stream)); }   // (a tree of one leaf group: no launch took it along)
  RQ(read_root(d_trace_nodes, ts, root));
  if (*reinterpret_cast<const unsigned long long*>(pinned)) return fail(MS_ERR_ARG, "trace element not canonical (>= p)");
  have_trace = true;
  return MS_OK;
}

// ------------------------------------------------------------------ air.rs:147-160
template <class F>
int Ctx<F>::interpolate() {
  if (!have_trace) return fail(MS_ERR_STATE, "interpolate before trace_commit");
  RQ(ntt_run(ctz64(N), true, d_polys.as<T>(), N, N, d_polys.as<T>(), N, w));
  poly_lin.assign(w, Lin());   // (first: if this allocation fails the session keeps no half-set state)
  poly_mat.assign(w, 1);
  npolys = (int)w; have_polys = true; have_lde = have_validity = false;
  return MS_OK;
}

template <class F>
int Ctx<F>::polys_lincomb(const u64* s, const int* idx, int k) {
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

template <class F>
int Ctx<F>::polys_append(const u64* coeffs, size_t n) {
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

template <class F>
int Ctx<F>::poly_read(int i, u64* out) {
  if (!have_polys || i < 0 || i >= npolys || !out) return fail(MS_ERR_ARG, "poly_read");
  RQ(materialize(i));
  return download_widen(d_polys.as<T>() + (size_t)i * N, N, 0, 1, out);
}

template <class F>
int Ctx<F>::arith_selftest(int op, const u64* a, const u64* b, u64* out, size_t n) {
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

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::profile_end(char* out, size_t cap); \
  template int Ctx<FF>::tabs_host(size_t bytes, u8** out); \
  template int Ctx<FF>::zero_alloc(size_t bytes, void** out); \
  template int Ctx<FF>::materialize(int i); \
  template int Ctx<FF>::ensure_polys(size_t count); \
  template int Ctx<FF>::init(int dev, u32 flags); \
  template int Ctx<FF>::upload_narrow(const u64* host, size_t n, Ctx<FF>::T* dst); \
  template int Ctx<FF>::download_widen(const Ctx<FF>::T* src, size_t n, size_t limb_stride, u32 e, u64* host); \
  template int Ctx<FF>::transpose_in(const u64* src, Ctx<FF>::T* dst, size_t rows, size_t cols, Ctx<FF>::T rinv, int mont, u32* bad); \
  template int Ctx<FF>::trace_commit(const u64* trace, bool on_device, size_t N_, size_t w_, size_t lpn, u8* root); \
  template int Ctx<FF>::interpolate(); \
  template int Ctx<FF>::polys_lincomb(const u64* s, const int* idx, int k); \
  template int Ctx<FF>::polys_append(const u64* coeffs, size_t n); \
  template int Ctx<FF>::poly_read(int i, u64* out); \
  template int Ctx<FF>::arith_selftest(int op, const u64* a, const u64* b, u64* out, size_t n);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

}  // namespace msctx
