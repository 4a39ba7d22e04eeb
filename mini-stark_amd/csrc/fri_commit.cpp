// fri_commit.cpp — the FRI commit phase (fri.rs:64-113, 301-377): round commitments, even(z) / odd(z), fold + DEEP quotient (replicated and by coefficient range),
// and the suffix-Horner job planning the query phase shares.
#include "ctx.hpp"
#include "fri_tail.hpp"

namespace msctx {

// codeword + tree of rounds[i] from its coefficient limbs (ncoef_in valid coefficients)
// `nonzero_limbs`: limbs >= this are identically zero (round 0: extend_poly embeds base coefficients), so their
// transform is all zeros and is not computed
// `prev` != nullptr: the codeword is folded out of prev's codeword in the evaluation domain (FriFoldEvalKernel) instead of
// transforming the round polynomial — same values, a quarter of the arithmetic
template <class F>
int Ctx<F>::round_commit(Round* r, size_t ncoef_in, int nonzero_limbs, const Round* prev, const XE* alpha) {
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
      { PartScopeIf part(this, shard_next); CK(total <= fold_small_max ? fold_launch((mspoly::FriFoldEvalKernel<F, E, 1>*)nullptr) : fold_launch((mspoly::FriFoldEvalKernel<F, E, 8>*)nullptr)); }
      if (shard_next) RQ((tree_build_sharded<E>(r->cw.template as<T>(), m_out, 1, 2 * m_out, 2, r->ts, r->nodes)));
      else RQ((tree_build<E>(r->cw.template as<T>(), 0, 1, r->D, 1, r->ts, r->nodes)));
      return 0;
    }
  }
  RQ(join_side());   // from here on the round polynomial itself is transformed: the side stream's quotient must be complete
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
template <class F>
int Ctx<F>::degree_launch(const T* poly, size_t limb_stride, size_t n, unsigned long long** dres_out) {
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

template <class F>
int Ctx<F>::read_degree_and_root(const T* poly, size_t limb_stride, size_t n, Round* r, size_t* ncoef, u8* root) {
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
template <class F>
int Ctx<F>::fri_begin(size_t blowup_, size_t nrounds, u8* root0) {
  if (!have_validity) return fail(MS_ERR_STATE, "fri_begin before mix");
  if (!root0 || nrounds < 1 || !blowup_) return fail(MS_ERR_ARG, "fri_begin");
  nrounds_done = 0; have_deep = false; blob_size = 0;
  fri_rounds = nrounds; fri_blowup = blowup_;
  // the self-clearing length word (and completion counter) of ms_fri_fold_commit: allocated and zeroed HERE, behind this stage's stream synchronisation - zeroing it
  // lazily in the first ms_fri_fold_commit raced with the side stream's length scan (r05: first proof of a context, eight lanes in flight) - and zero again even if
  // an earlier proof was abandoned mid-round
  if (!d_deg.p && d_deg.ensure(256)) return fail(MS_ERR_NOMEM, "degree word");
  CK(msrt::memset_dev(d_deg.p, 0, 256, stream));
  Round* r = round_slot(0);
  const size_t VL = validity_len;   // N (ms_mix) or 2N (ms_mix_cubic)
  r->cap = VL;
  if (r->poly.ensure(VL * E * sizeof(T))) return fail(MS_ERR_NOMEM, "round poly");
  // field.rs:23-32 extend_poly: limb 0 = validity, higher limbs zero
  CK(msrt::memset_dev(r->poly.p, 0, VL * E * sizeof(T), stream));
  CK(msrt::d2d(r->poly.p, d_polys.as<T>() + (size_t)npolys * N, VL * sizeof(T), stream));
  size_t nc;
  if (validity_len_host) nc = validity_ncoef;   // (came to the host with DEEP-ALI's values: ms_mix scans, ms_eval_ext forwards)
  else RQ(read_degree_and_root(r->poly.template as<T>(), r->cap, VL, nullptr, &nc, nullptr));
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
template <class F>
int Ctx<F>::fri_deep(const u64* z, u64* B) {
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
  { arm_next_eval = true; RQ((eval_views<E>(r->poly.template as<T>(), 0, r->cap, 2, off, cnt, 2, cur_z, dst))); }  // fri.rs:354-359
  CK(sync_results());
  const T* h = reinterpret_cast<const T*>(pinned);
  for (int s = 0; s < 2; s++) for (int l = 0; l < E; l++) { cur_B[s].c[l] = h[s * E + l]; B[s * E + l] = F::to_u64(h[s * E + l]); }
  have_deep = true;
  return MS_OK;
}

// `scratch` must hold sh_scratch_elems(m) elements of T
// ext_carry (E limbs, device): carry-in of the top level (a rank of a sharded proof: the suffix sum over the higher ranks, scaled - ShardCarryKernel);
// top_agg (E limbs, device): the job's aggregate over all its elements, stored by one extra AGG launch of the top level; out_h0: see SHJob
template <class F>
typename Ctx<F>::SHPlan Ctx<F>::sh_plan(const T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const XE& z,
               void* out, bool out_u64, size_t out_limb_stride, size_t out_off, size_t out_stride, T* h0, T* scratch,
               const T* ext_carry, T* top_agg, bool out_h0) {
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

template <class F>
int Ctx<F>::sh_launch_inline(const SHJ& j, int final_mode) {
  typename SHK::Params p; p.jobs = nullptr; p.inline_job = j; p.final_mode = final_mode;
  const size_t nb = j.m ? (j.m + mspoly::SH_BS - 1) / mspoly::SH_BS : 1;
  CK(run_coop<SHK>(K_SUFFIX_HORNER, (unsigned)nb, SHK::THREADS, SHK::lds_bytes(), p));
  return 0;
}

// a device table of `njobs` homogeneous jobs (the query phase's batched scans): blockIdx.y = job, `maxnb` = blocks of the longest one
template <class F>
int Ctx<F>::sh_launch_table(const SHJ* d_jobs, size_t njobs, size_t maxnb, int final_mode) {
  typename SHK::Params p; p.jobs = d_jobs; p.final_mode = final_mode;
  memset(&p.inline_job, 0, sizeof p.inline_job);
  CK(run_coop<SHK>(K_SUFFIX_HORNER, (unsigned)maxnb, SHK::THREADS, SHK::lds_bytes(), p, (unsigned)njobs));
  return 0;
}
// the carry-ins (and H_0's) of a device table of jobs from the all-gathered aggregates in the receive buffer (rank r's payload: njobs * E limbs)
template <class F>
int Ctx<F>::shard_carry_table(const mspoly::CarryJob<F, E>* d_jobs, size_t njobs) {
  typedef mspoly::ShardCarryKernel<F, E> CKn;
  typename CKn::Params cp; memset(&cp, 0, sizeof cp);
  cp.jobs = d_jobs; cp.njobs = (u32)njobs; cp.W = (u32)sh_world; cp.rank = (u32)sh_rank;
  cp.gathered = reinterpret_cast<const T*>(xr); cp.rank_stride = njobs * E;
  CK(run<CKn>(K_SUFFIX_HORNER, grid1(njobs, CKn::THREADS), 1, CKn::THREADS, 0, cp));
  return 0;
}
// one logical job, launched level by level with the job inline in the kernel arguments
template <class F>
int Ctx<F>::suffix_horner(const T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const XE& z,
                  T* out, size_t out_limb_stride, size_t out_off, size_t out_stride, T* h0) {
  if (d_sh.ensure(sh_scratch_elems(m) * sizeof(T))) return fail(MS_ERR_NOMEM, "scan levels");
  SHPlan pl = sh_plan(in, in_limb_stride, in_off, in_stride, m, z, out, false, out_limb_stride, out_off, out_stride, h0, d_sh.as<T>());
  for (int l = 0; l + 1 < pl.nl; l++) RQ(sh_launch_inline(pl.agg[l], 0));
  for (int l = pl.nl - 1; l >= 0; l--) RQ(sh_launch_inline(pl.fin[l], 1));
  return 0;
}

// all-gather of a distributed round polynomial's parts into one replicated vector (dst: E limbs, dst_stride apart, `count` coefficients)
template <class F>
int Ctx<F>::gather_poly(const T* local, size_t S, size_t count, T* dst, size_t dst_stride) {
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
template <class F>
int Ctx<F>::fold_dist(Round* pr, Round* nr, const XE& a, size_t* nq_coef_out) {
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
    T* pay = reinterpret_cast<T*>(xs);        // [first folded element (E limbs) | aggregate of the job (E limbs)]: the fold's first thread writes [f_0 | 0], the aggregate launch the rest
    typename mspoly::FoldKernel<F, E>::Params fp{lpoly(pr), lstride(pr), cntp, d_folded.as<T>(), Sn, a, pay};  // fri.rs:361-372
    if (cnt) CK(run<mspoly::FoldKernel<F, E>>(K_FOLD, grid1(cnt, mspoly::THREADS), 1, mspoly::THREADS, 0, fp));
    else CK(msrt::memset_dev(xs, 0, 2 * E * sizeof(T), stream));   // (a rank with no coefficient of this round: an all-zero payload)
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

// fri.rs:96-109 for a round of the latency-bound tail, as ONE launch (fri_tail.hpp): fold, DEEP quotient, trimmed length, pointwise codeword, leaf digests, every tree
// level, root and length word into page-locked memory.  *done = false: the round does not qualify (the caller takes the launch-per-step path).
template <class F>
int Ctx<F>::fri_tail_round(Round* pr, Round* nr, const XE& a, bool* done) {
  *done = false;
  const size_t n = pr->ncoef, m = (n + 1) / 2, Dn = nr->D, M = Dn / 2;
  bool z_outside_base = false;
  for (int l = 1; l < E; l++) z_outside_base = z_outside_base || cur_z.c[l] != 0;
  // replicated rounds only (a sharded proof's tail is replicated on every rank); the pointwise codeword needs y - z != 0 on the base-field domain; one scan block
  if (!fri_tail_max || pr->D > fri_tail_max || pr->dist || pr->ts.sharded || shardable(M) || !fri_pointwise || !z_outside_base || m > (size_t)mspoly::SH_BS || M < 1) return 0;
  typedef msfri::FriTailKernel<F, E> TK;
  if (M > (size_t)TK::WG_GROUPS << msmerkle::InnerSubtreeKernel::MAX_LEVELS) return 0;
  if (ctz64(Dn) > F::TWO_ADICITY) return fail(MS_ERR_SHAPE, "FRI domain larger than the field's two-adicity");
  RQ(tree_shape(Dn, 2, 2, &nr->ts));
  nr->m = 0;
  if (nr->poly.ensure(nr->cap * E * sizeof(T)) || d_folded.ensure((m + 1) * E * sizeof(T)) || nr->cw.ensure(Dn * E * sizeof(T)) || nr->nodes.ensure(nr->ts.nodes * 32) ||
      d_sh.ensure(sh_scratch_elems(m) * sizeof(T)))
    return fail(MS_ERR_NOMEM, "fused FRI round");
  Plan* pl;
  RQ(get_plan(ctz64(pr->D), 0, false, &pl));   // w_D^e tables of the previous domain
  typename TK::Params tp;
  memset(&tp, 0, sizeof tp);
  // ---- coefficient side
  tp.do_coef = m >= 2 ? 1 : 0;
  tp.fold = typename TK::FoldK::Params{pr->poly.template as<T>(), pr->cap, n, d_folded.as<T>(), m, a, nullptr};
  if (tp.do_coef) {
    SHPlan sp = sh_plan(d_folded.as<T>(), m, 0, 1, m, cur_z, nr->poly.template as<T>(), false, nr->cap, 0, 1, nullptr, d_sh.as<T>());
    if (sp.nl != 1) return fail(MS_ERR_STATE, "fused FRI round: scan of more than one block");
    tp.scan.jobs = nullptr; tp.scan.inline_job = sp.fin[0]; tp.scan.final_mode = 1;
  }
  unsigned long long* dres = d_deg.as<unsigned long long>();       // the length word: zero between rounds (the forwarding thread clears it)
  tp.deg = typename TK::DegK::Params{nr->poly.template as<T>(), nr->cap, m >= 2 ? m - 1 : 0, dres, 0};
  // ---- evaluation side
  const size_t per = M < (size_t)TK::WG_GROUPS ? M : (size_t)TK::WG_GROUPS, G = M / per;
  tp.G = (u32)G;
  const XE c = e_add<F, E>(cur_B[0], e_mul<F>(cur_B[1], a));         // B(alpha), fri.rs:99
  tp.eval.src = pr->cw.template as<T>(); tp.eval.src_limb_stride = pr->D; tp.eval.dst = nr->cw.template as<T>(); tp.eval.dst_limb_stride = Dn;
  tp.eval.m_out = Dn; tp.eval.log_m = (u32)ctz64(Dn); tp.eval.groups = 1; tp.eval.shard_W = 0; tp.eval.shard_k = 0;
  tp.eval.tw_lo = pl->tw_lo.template as<T>(); tp.eval.tw_hi = pl->tw_hi.template as<T>(); tp.eval.lo_bits = (u32)pl->lo_bits; tp.eval.log_D = (u32)ctz64(pr->D);
  tp.eval.alpha = a; tp.eval.c2 = e_add<F, E>(c, c); tp.eval.z = cur_z; tp.eval.inv2 = f_inv<F>(F::from_u64(2));
  tp.leaf.base = nr->cw.template as<T>(); tp.leaf.col_stride = 0; tp.leaf.row_stride = 1; tp.leaf.limb_stride = Dn; tp.leaf.width = 1; tp.leaf.lpn = 2;
  tp.leaf.zero_as_empty = zae; tp.leaf.ngroups = M; tp.leaf.nodes = nr->nodes.template as<u32>();   // (ovf* = null: pad-only blocks in place; no runs, no virtual columns)
  u32 nl = 0; while (((size_t)1 << nl) < per) nl++;
  tp.sub.nodes = nr->nodes.template as<u32>(); tp.sub.child_off = 0; tp.sub.nchildren = M; tp.sub.ic = 2; tp.sub.nlevels = nl;
  u32 tl = 0; while (((size_t)1 << tl) < G) tl++;
  tp.top.nodes = nr->nodes.template as<u32>(); tp.top.child_off = 2 * M - 2 * G; tp.top.nchildren = G; tp.top.ic = 2; tp.top.nlevels = tl;
  tp.top.host_root = host_root(); tp.top.aux_src = dres; tp.top.aux_dst = reinterpret_cast<unsigned long long*>(pinned); tp.top.flag = arm_flag();
  tp.root_index = nr->ts.nodes - 1;
  tp.done = reinterpret_cast<u32*>(d_deg.as<u8>() + 128);            // (same zeroed line as the length word; the last workgroup leaves it zero)
  next_bytes = (double)Dn * E * sizeof(T) * 3 + (double)M * 96;
  CK(run_coop<TK>(K_FRI_TAIL, (unsigned)(G + 1), TK::THREADS, TK::lds_bytes(), tp));
  root_on_host = true; aux_on_host = true;
  *done = true;
  return 0;
}

// fri.rs:96-109
template <class F>
int Ctx<F>::fri_fold_commit(const u64* alpha, u8* root) {
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
  bool fused = false, side = false;
  RQ(fri_tail_round(pr, nr, a, &fused));
  if (fused) {   // one launch did the whole round; the length word and the root are in page-locked memory behind the stream synchronisation (or the polled flag)
    CK(sync_results());
    nr->ncoef = (size_t)(*reinterpret_cast<unsigned long long*>(pinned));
    memcpy(root, reinterpret_cast<const u8*>(host_root()), 32);
    nrounds_done++; have_deep = false;
    return MS_OK;
  }
  if (pr->dist) RQ(fold_dist(pr, nr, a, &nq_coef));
  else {
  if (nr->poly.ensure(nr->cap * E * sizeof(T)) || d_folded.ensure((m + 1) * E * sizeof(T))) return fail(MS_ERR_NOMEM, "fold");
  if (m >= 2) {
    // the coefficient side on the side stream (every stage ends with a synchronisation of the context's stream behind a join: the side stream starts on finished data)
    side = fri_overlap && !prof_on;   // (the per-kernel profile times launches one behind the other on ONE stream)
    if (side && !side_stream) { CK(msrt::stream_create(&side_stream)); CK(msrt::event_create(&ev_side)); }
    StreamScope on_side(this, side ? side_stream : stream);
    typename mspoly::FoldKernel<F, E>::Params fp{pr->poly.template as<T>(), pr->cap, n, d_folded.as<T>(), m, a, nullptr};  // fri.rs:361-372
    CK(run<mspoly::FoldKernel<F, E>>(K_FOLD, grid1(m, mspoly::THREADS), 1, mspoly::THREADS, 0, fp));
    // (folded - B(alpha)) / (x - z): quotient coefficients are H_1.. of the suffix Horner in z (fri.rs:99-101)
    RQ(suffix_horner(d_folded.as<T>(), m, 0, 1, m, cur_z, nr->poly.template as<T>(), nr->cap, 0, 1, nullptr));
    nq_coef = m - 1;
  }
  }
  // the degree scan runs BEFORE the commitment and the tree's final launch forwards its 8-byte result, with the root, into page-locked host memory:
  // no copy launch in front of the round's one stream synchronisation (r03; the trace showed a 4 us copyBuffer kernel + its launch gap per round)
  // (the word lives in d_deg, zeroed once: the forwarding thread clears it again - the pool of zero_alloc may be wiped while the tree is being built)
  unsigned long long* dres = d_deg.as<unsigned long long>();   // (zeroed by ms_fri_begin)
  if (nr->dist) {   // this rank's part reports the GLOBAL trimmed length; the maximum over the ranks comes back with the subtree roots (finish_sharded_tree)
    const size_t lc = lcount(nr, nq_coef);
    if (lc) {
      typename mspoly::DegreeKernel<F, E>::Params dp{lpoly(nr), lstride(nr), lc, dres, (size_t)sh_rank * nr->S};
      PartScope part(this);
      CK(run<mspoly::DegreeKernel<F, E>>(K_DEGREE, grid1(lc, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
    }
    shard_aux = dres;
  } else if (nq_coef) {
    StreamScope on_side(this, side ? side_stream : stream);
    typename mspoly::DegreeKernel<F, E>::Params dp{nr->poly.template as<T>(), nr->cap, nq_coef, dres, 0};
    CK(run<mspoly::DegreeKernel<F, E>>(K_DEGREE, grid1(nq_coef, mspoly::THREADS), 1, mspoly::THREADS, 0, dp));
  }
  if (side) { CK(msrt::event_record(ev_side, side_stream)); side_pending = true; }
  pending_aux = dres; aux_on_host = false;
  { auto clear = scope_exit([this] { pending_aux = nullptr; shard_aux = nullptr; });   // (also on an error exit: neither word may ride on the NEXT commitment's launches - ADVICE r4)
    auto join = scope_exit([this] { if (side_pending) { side_pending = false; seq_armed = 0; msrt::stream_wait_event(stream, ev_side); } });   // whatever path the commitment took (or left on): the context's stream ends behind the side stream
    RQ(round_commit(nr, nq_coef, E, pr, &a)); }
  if (!aux_on_host) { seq_armed = 0; CK(msrt::d2h(pinned, dres, 8, stream)); CK(msrt::memset_dev(dres, 0, 8, stream)); }
  if (!root_on_host) { seq_armed = 0; CK(msrt::d2h(reinterpret_cast<u8*>(pinned) + 64, nr->nodes.template as<u8>() + (nr->ts.local_nodes - 1) * 32, 32, stream)); }
  CK(sync_results());   // (polls the flag of the tree's last launch if nothing was enqueued behind it)
  nr->ncoef = (size_t)(*reinterpret_cast<unsigned long long*>(pinned));
  memcpy(root, root_on_host ? reinterpret_cast<const u8*>(host_root()) : reinterpret_cast<const u8*>(pinned) + 64, 32);
  nrounds_done++; have_deep = false;
  return MS_OK;
}

template <class F>
int Ctx<F>::fri_round_info(int r, u64* ncoef, u64* D) {
  if (r < 0 || (size_t)r >= nrounds_done) return fail(MS_ERR_ARG, "round index");
  if (ncoef) *ncoef = rounds[r]->ncoef;
  if (D) *D = rounds[r]->D;
  return MS_OK;
}

template <class F>
int Ctx<F>::fri_round_poly_read(int r, u64* out) {
  if (r < 0 || (size_t)r >= nrounds_done || !out) return fail(MS_ERR_ARG, "round index");
  if (rounds[r]->local_store) return fail(MS_ERR_STATE, "round_poly_read: the polynomial of a sharded round is distributed over the ranks");
  return download_widen(rounds[r]->poly.template as<T>(), rounds[r]->ncoef, rounds[r]->cap, E, out);
}

template <class F>
int Ctx<F>::fri_round_codeword_read(int r, u64* out) {
  if (r < 0 || (size_t)r >= nrounds_done || !out) return fail(MS_ERR_ARG, "round index");
  if (rounds[r]->ts.sharded) return fail(MS_ERR_STATE, "codeword_read: the codeword of a sharded round is distributed over the ranks");
  return download_widen(rounds[r]->cw.template as<T>(), rounds[r]->D, rounds[r]->D, E, out);
}

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::round_commit(Ctx<FF>::Round* r, size_t ncoef_in, int nonzero_limbs, const Ctx<FF>::Round* prev, const Ctx<FF>::XE* alpha); \
  template int Ctx<FF>::degree_launch(const Ctx<FF>::T* poly, size_t limb_stride, size_t n, unsigned long long** dres_out); \
  template int Ctx<FF>::read_degree_and_root(const Ctx<FF>::T* poly, size_t limb_stride, size_t n, Ctx<FF>::Round* r, size_t* ncoef, u8* root); \
  template int Ctx<FF>::fri_begin(size_t blowup_, size_t nrounds, u8* root0); \
  template int Ctx<FF>::fri_deep(const u64* z, u64* B); \
  template Ctx<FF>::SHPlan Ctx<FF>::sh_plan(const Ctx<FF>::T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const Ctx<FF>::XE& z, void* out, bool out_u64, size_t out_limb_stride, size_t out_off, size_t out_stride, Ctx<FF>::T* h0, Ctx<FF>::T* scratch, const Ctx<FF>::T* ext_carry, Ctx<FF>::T* top_agg, bool out_h0); \
  template int Ctx<FF>::sh_launch_inline(const Ctx<FF>::SHJ& j, int final_mode); \
  template int Ctx<FF>::sh_launch_table(const Ctx<FF>::SHJ* d_jobs, size_t njobs, size_t maxnb, int final_mode); \
  template int Ctx<FF>::shard_carry_table(const mspoly::CarryJob<FF, Ctx<FF>::E>* d_jobs, size_t njobs); \
  template int Ctx<FF>::suffix_horner(const Ctx<FF>::T* in, size_t in_limb_stride, size_t in_off, size_t in_stride, size_t m, const Ctx<FF>::XE& z, Ctx<FF>::T* out, size_t out_limb_stride, size_t out_off, size_t out_stride, Ctx<FF>::T* h0); \
  template int Ctx<FF>::gather_poly(const Ctx<FF>::T* local, size_t S, size_t count, Ctx<FF>::T* dst, size_t dst_stride); \
  template int Ctx<FF>::fold_dist(Ctx<FF>::Round* pr, Ctx<FF>::Round* nr, const Ctx<FF>::XE& a, size_t* nq_coef_out); \
  template int Ctx<FF>::fri_tail_round(Ctx<FF>::Round* pr, Ctx<FF>::Round* nr, const Ctx<FF>::XE& a, bool* done); \
  template int Ctx<FF>::fri_fold_commit(const u64* alpha, u8* root); \
  template int Ctx<FF>::fri_round_info(int r, u64* ncoef, u64* D); \
  template int Ctx<FF>::fri_round_poly_read(int r, u64* out); \
  template int Ctx<FF>::fri_round_codeword_read(int r, u64* out);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

}  // namespace msctx
