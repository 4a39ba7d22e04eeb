// air_stages.cpp — constraint columns by linear provenance, coset LDE + commitment (starks.rs:80-95), constraint mixing (starks.rs:108-119; build-defined
// ms_mix_cubic), DEEP-ALI evaluations (starks.rs:124-151).
#include "ctx.hpp"

namespace msctx {

// ------------------------------------------------------------------ starks.rs:80-95
// one lincomb launch chain: dst = sum_t s[t] * base[idx[t]] over n elements (columns `stride` apart)
template <class F>
int Ctx<F>::lincomb_into(const T* base, size_t stride, size_t n, const u64* sc, const int* idx, int k, int self_index, T* dst) {
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
template <class F>
int Ctx<F>::lincomb_linear_columns(T* base, size_t stride, size_t n) {
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
template <class F>
int Ctx<F>::lde_compute(size_t blowup_, u64 shift) {
  const size_t c = (size_t)npolys;
  const size_t L_ = N * blowup_;
  if (d_coef.ensure(c * N * sizeof(T)) || d_lde.ensure(c * L_ * sizeof(T))) return fail(MS_ERR_NOMEM, "lde");
  if (!lde_linear) for (size_t i = 0; i < c; i++) RQ(materialize((int)i));
  const T sh = F::from_u64(shift);
  for (size_t i = 0; i < c;) {  // maximal runs of polynomials that need a transform
    if (lde_linear && !poly_lin[i].idx.empty()) { i++; continue; }
    size_t j = i;
    while (j < c && !(lde_linear && !poly_lin[j].idx.empty())) j++;
    RQ(scale_pow(d_polys.as<T>() + i * N, N, d_coef.as<T>() + i * N, N, N, sh, j - i));
    RQ(ntt_run(ctz64(L_), false, d_coef.as<T>() + i * N, N, N, d_lde.as<T>() + i * L_, L_, j - i));
    i = j;
  }
  if (lde_linear) RQ(finish_linear_columns(L_, L_));
  return 0;
}

template <class F>
int Ctx<F>::finish_linear_columns(size_t stride, size_t n) {
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

// lde_compute for a sharded proof: column i of the local LDE (rows rank + W*j) at d_lde + i*m, m = L/W
template <class F>
int Ctx<F>::lde_compute_sharded(size_t blowup_, u64 shift) {
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

template <class F>
int Ctx<F>::lde_commit(size_t blowup_, u64 shift, size_t lpn, u8* root) {
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

template <class F>
int Ctx<F>::bench_lde(size_t blowup_, u64 shift) {
  if (!have_polys) return fail(MS_ERR_STATE, "bench_lde before interpolate");
  have_lde = false;                              // d_lde is overwritten: whatever ms_lde_commit left there is gone
  const int keep = lde_virtual; lde_virtual = 0; // time the columns as WRITTEN (a virtual column's work would move into a leaf kernel this entry does not run)
  auto restore = scope_exit([this, keep] { lde_virtual = keep; lde_cols_virtual = false; });   // (also when lde_compute throws: ADVICE r4)
  return lde_compute(blowup_, shift);
}

template <class F>
int Ctx<F>::lde_read(u64* out) {
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
template <class F>
int Ctx<F>::mix(u64 r) {
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
  // the trimmed length ms_fri_begin needs (fri.rs:74), found HERE: the word travels to the host on DEEP-ALI's last evaluation launch (ms_eval_ext), which the caller
  // waits for anyway - ms_fri_begin then starts its transform without a launch, a copy and a host round trip of its own (r05)
  validity_len_host = false; validity_len_dev = nullptr;
  if (!dist_eval()) {   // (a sharded proof's DEEP-ALI ends in a collective, not in an evaluation launch: ms_fri_begin keeps its own scan there)
  RQ(ensure_evdone());
  validity_len_dev = reinterpret_cast<unsigned long long*>(d_evdone.as<u8>() + 128);   // (a word of its own, not the zero pool: the pool may be wiped before ms_eval_ext comes)
  CK(msrt::memset_dev(validity_len_dev, 0, 8, stream));
  { typename mspoly::DegreeKernel<F, 1>::Params dp{d_polys.as<T>() + (size_t)npolys * N, 0, N, validity_len_dev, 0};
    CK(run<mspoly::DegreeKernel<F, 1>>(K_DEGREE, grid1(N, mspoly::THREADS), 1, mspoly::THREADS, 0, dp)); }
  }
  have_validity = true; validity_len = N; nrounds_done = 0;
  return MS_OK;
}

template <class F>
int Ctx<F>::mix_cubic(u64 r, const int* spec, const u64* sc, int ncons) {
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
  validity_ncoef = (size_t)(*reinterpret_cast<unsigned long long*>(pinned)); validity_len_host = true; validity_len_dev = nullptr;   // (scaling by shift^-k keeps the trimmed length)
  const T shi = f_inv<F>(sh);
  RQ(scale_pow(coef, 0, d_polys.as<T>() + (size_t)npolys * N, 0, 2 * N, shi, 1));
  have_validity = true; validity_len = 2 * N; nrounds_done = 0;
  return MS_OK;
}

// trimmed length of a base-field coefficient vector
template <class F>
int Ctx<F>::degree_launch1(const T* poly, size_t n, unsigned long long** dres_out) {
  void* zr;
  RQ(zero_alloc(8, &zr));
  typename mspoly::DegreeKernel<F, 1>::Params dp{poly, 0, n, reinterpret_cast<unsigned long long*>(zr), 0};
  CK(run<mspoly::DegreeKernel<F, 1>>(K_DEGREE, grid1(n, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
  *dres_out = reinterpret_cast<unsigned long long*>(zr);
  return 0;
}

template <class F>
int Ctx<F>::validity_read(u64* out) {
  if (!have_validity || !out) return fail(MS_ERR_STATE, "validity_read");
  return download_widen(d_polys.as<T>() + (size_t)npolys * N, validity_len, 0, 1, out);
}

// evaluate `npoly` polynomials (views) at ext point z into dst as [npoly][E] T
template <class F> template <int EC>
int Ctx<F>::eval_views(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst) {
  size_t maxc = 0;
  for (int i = 0; i < npoly; i++) if (count[i] > maxc) maxc = count[i];
  return maxc <= eval_small_max ? eval_views_i<EC, 4>(base, poly_stride, limb_stride, kstride, off, count, npoly, z, dst, maxc)
                                : eval_views_i<EC, 16>(base, poly_stride, limb_stride, kstride, off, count, npoly, z, dst, maxc);
}

template <class F> template <int EC, int ITEMS>
int Ctx<F>::eval_views_i(const T* base, size_t poly_stride, size_t limb_stride, size_t kstride, const size_t* off, const size_t* count, int npoly, const XE& z, T* dst, size_t maxc) {
  typedef mspoly::EvalKernel<F, EC, E, ITEMS> EK;
  const bool arm = arm_next_eval; arm_next_eval = false;   // the caller's LAST evaluation: its results end the stage (sync_results).  (Taken first: no exit below leaves it set)
  const unsigned long long* fwd = fwd_next_eval; fwd_next_eval = nullptr;   // ... and a device word it takes along to host_aux2()
  const size_t chunk = (size_t)EK::THREADS * EK::ITEMS;
  const size_t nblocks = maxc ? (maxc + chunk - 1) / chunk : 1;
  if (nblocks > 1 && d_partials.ensure(nblocks * npoly * E * sizeof(T))) return fail(MS_ERR_NOMEM, "partials");
  typename EK::Params p;
  p.base = base; p.poly_stride = poly_stride; p.limb_stride = limb_stride; p.kstride = kstride; p.npoly = npoly;
  for (int i = 0; i < mspoly::MAX_POLYS; i++) { p.off[i] = i < npoly ? off[i] : 0; p.count[i] = i < npoly ? count[i] : 0; }
  XE sq = z;
  for (int i = 0; i < 9; i++) { p.zpow2[i] = sq; sq = e_mul<F>(sq, sq); }
  p.partials = nblocks > 1 ? d_partials.as<T>() : dst;  // single block: P_0 is the value
  p.flag = msrt::HostFlag{nullptr, 0}; p.aux_src = nullptr; p.aux_dst = nullptr; p.single = nblocks == 1;
  if (nblocks == 1) {
    if (fwd) { p.aux_src = fwd; p.aux_dst = host_aux2(); }
    if (arm) p.flag = arm_flag();
  }
  CK(run_coop<EK>(K_EVAL, (unsigned)nblocks, EK::THREADS, EK::lds_bytes(), p));
  if (nblocks > 1) {
    typedef mspoly::ReducePartialsKernel<F, E> RK;
    typename RK::Params rp;
    memset(&rp, 0, sizeof rp);
    rp.partials = d_partials.as<T>(); rp.nblocks = nblocks; rp.per_thread = (nblocks + RK::THREADS - 1) / RK::THREADS; rp.npoly = npoly; rp.out = dst;
    XE zc = p.zpow2[8];  // z^256
    for (int i = 256; i < (int)chunk; i *= 2) zc = e_mul<F>(zc, zc);  // z^CH
    XE zs = zc;
    for (int i = 0; i < 8; i++) { rp.zs2[i] = zs; zs = e_mul<F>(zs, zs); }   // zc^(2^i)
    rp.zc_step = zs;                                                          // zc^256 = zc^THREADS
    static_assert(RK::THREADS == 256, "zc_step = zc^THREADS");
    if (fwd) { rp.aux_src = fwd; rp.aux_dst = host_aux2(); }
    if (arm) rp.flag = arm_flag();
    CK(run_coop<RK>(K_EVAL_REDUCE, 1, RK::THREADS, RK::lds_bytes(), rp));
  }
  return 0;
}

// sharded proof: out[i] = sum_r part_r[i] * zstep^r for n extension elements of the all-gathered partials (rank r's payload rank_stride limbs apart, the elements from `off` on)
template <class F>
int Ctx<F>::shard_combine_launch(size_t off, size_t rank_stride, u32 n, const XE& zstep, T* out) {
  typedef mspoly::ShardCombineKernel<F, E> CKn;
  typename CKn::Params cp{reinterpret_cast<const T*>(xr) + off, rank_stride, n, (u32)sh_world, zstep, out};
  CK(run<CKn>(K_EVAL_REDUCE, grid1(n, CKn::THREADS), 1, CKn::THREADS, 0, cp));
  return 0;
}

// page-locked results [q][ev.size()][E] -> out [q][npolys + 1][E] (u64)
template <class F>
int Ctx<F>::eval_finish(int q, const std::vector<int>& ev, u64* out) {
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

template <class F>
int Ctx<F>::eval_ext_sharded(const u64* z, int q, u64* out) {
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
template <class F>
int Ctx<F>::eval_ext(const u64* z, int q, u64* out) {
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
      arm_next_eval = t == q - 1 && i0 + mspoly::MAX_POLYS >= nev;
      if (arm_next_eval && validity_len_dev && !validity_len_host) fwd_next_eval = validity_len_dev;
      RQ((eval_views<1>(d_polys.as<T>(), 0, 0, 1, off, cnt, nb, zz, reinterpret_cast<T*>(pinned) + ((size_t)t * nev + i0) * E)));   // results land in page-locked host memory
    }
  }
  if (tot) {
    CK(sync_results());
    if (validity_len_dev && !validity_len_host) { validity_ncoef = (size_t)*host_aux2(); validity_len_host = true; }
    RQ(eval_finish(q, ev, out));
  }
  return MS_OK;
}

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::lincomb_into(const Ctx<FF>::T* base, size_t stride, size_t n, const u64* sc, const int* idx, int k, int self_index, Ctx<FF>::T* dst); \
  template int Ctx<FF>::lincomb_linear_columns(Ctx<FF>::T* base, size_t stride, size_t n); \
  template int Ctx<FF>::lde_compute(size_t blowup_, u64 shift); \
  template int Ctx<FF>::finish_linear_columns(size_t stride, size_t n); \
  template int Ctx<FF>::lde_compute_sharded(size_t blowup_, u64 shift); \
  template int Ctx<FF>::lde_commit(size_t blowup_, u64 shift, size_t lpn, u8* root); \
  template int Ctx<FF>::bench_lde(size_t blowup_, u64 shift); \
  template int Ctx<FF>::lde_read(u64* out); \
  template int Ctx<FF>::mix(u64 r); \
  template int Ctx<FF>::mix_cubic(u64 r, const int* spec, const u64* sc, int ncons); \
  template int Ctx<FF>::degree_launch1(const Ctx<FF>::T* poly, size_t n, unsigned long long** dres_out); \
  template int Ctx<FF>::validity_read(u64* out); \
  template int Ctx<FF>::shard_combine_launch(size_t off, size_t rank_stride, u32 n, const Ctx<FF>::XE& zstep, Ctx<FF>::T* out); \
  template int Ctx<FF>::eval_finish(int q, const std::vector<int>& ev, u64* out); \
  template int Ctx<FF>::eval_ext_sharded(const u64* z, int q, u64* out); \
  template int Ctx<FF>::eval_ext(const u64* z, int q, u64* out);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

// eval_views<EC>: base-field coefficients (DEEP-ALI, EC = 1) here, extension coefficients (a FRI round's even(z) / odd(z)) from fri_commit.cpp
#define MS_INSTANTIATE_EC(FF, EC) \
  template int Ctx<FF>::eval_views<EC>(const Ctx<FF>::T*, size_t, size_t, size_t, const size_t*, const size_t*, int, const Ctx<FF>::XE&, Ctx<FF>::T*);
MS_INSTANTIATE_EC(GL, 1)
MS_INSTANTIATE_EC(GL, 2)
MS_INSTANTIATE_EC(BB, 1)
MS_INSTANTIATE_EC(BB, 4)
#undef MS_INSTANTIATE_EC

}  // namespace msctx
