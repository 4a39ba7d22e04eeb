// fri_query.cpp — the FRI query phase (fri.rs:115-189, merkle.rs:216-288) as a handful of batched launches over device job tables, and the standalone ms_merkle_prove.  (The proof's way to the host: io.cpp.)
#include "ctx.hpp"

namespace msctx {

// ---- The query phase in five steps over one QueryPlan (ctx.hpp): layout of the MSFP blob -> job tables (host) -> the quotient scans -> the ranks' slices into the
// blob (sharded proofs) -> opened points, leaf lookup by value, Merkle paths.  Whatever the number of rounds and queries it is a fixed handful of batched launches:
// every per-(window, query) step is a job in a device-side table.

// 1. layout of the MSFP blob (include/ministark.h): per window and query a record (points, quotient) and two Merkle paths
template <class F>
int Ctx<F>::query_layout(QueryPlan& q, int nq) {
  const size_t W = fri_rounds - 1;  // windows (previous, round); round W is only evaluated
  q.W = W; q.nq = nq;
  q.rec_off.assign(W * nq, 0); q.path_off.assign(W * nq * 2, 0); q.qlen.assign(W ? W : 1, 0);
  size_t pos = 0;
  for (size_t i = 0; i < W; i++) {
    Round* pr = rounds[i];
    if (pr->D / 2 != rounds[i + 1]->D) return fail(MS_ERR_SHAPE, "round domains do not halve (fri.rs:134-137)");
    q.qlen[i] = pr->ncoef >= 3 ? pr->ncoef - 2 : 0;
    const size_t nlev = pr->ts.levels - 1;
    const size_t path_bytes = 8 + 2 * E * 8 + 8 + nlev * 2 * 32;
    for (int j = 0; j < nq; j++) {
      q.rec_off[i * nq + j] = pos;
      pos += 6 * E * 8 + 8 + q.qlen[i] * E * 8;
      q.path_off[(i * nq + j) * 2] = pos; pos += path_bytes;
      q.path_off[(i * nq + j) * 2 + 1] = pos; pos += path_bytes;
    }
  }
  q.pos = pos;
  return 0;
}

// 2. the job tables, built on the host and uploaded in ONE copy: [scan tables][find jobs][path jobs][record offsets][quotient lengths][x1][carry jobs][slice copies]
template <class F>
int Ctx<F>::query_build(QueryPlan& q, const u64* betas) {
  const size_t W = q.W; const int nq = q.nq; u8* blob = q.blob;
  T* d_h0 = q.d_h0; T* d_tg = q.d_tg; unsigned long long* d_ix = q.d_ix;
  // ---- suffix Horner jobs: (f - g)/((x-x1)(x-x2)) = Qe(x^2) + x Qo(x^2), Qe = (even(y) - even(x3))/(y - x3) and the
  //      same for odd (fri.rs:159-167); the H_0 outputs are even(x3), odd(x3) (fri.rs:151-153)
  q.x1h.assign((W + 1) * nq, 0);
  // ---- sharded proof, distributed rounds (r04): the quotient jobs BY COEFFICIENT RANGE.  Rank k runs every job over its own half-range [k*S/2, (k+1)*S/2) of the even / odd
  // coefficients with the sum over the higher ranks as carry-in (ONE all-gather of the jobs' aggregates, ShardCarryKernel, which also gives every rank the H_0's),
  // and writes its slice of every quotient polynomial - a contiguous range [t0, t1) of the record's interleaved (even, odd) coefficients - into a packed buffer;
  // the slices are then all-gathered (or gathered to rank 0: ms_shard_proof_on_root) and copied into the blob.
  typedef typename QueryPlan::CJ CJ;
  const size_t Wr = (size_t)sh_world;
  std::vector<size_t> sl_t0(Wr * W * nq, 0), sl_len(Wr * W * nq, 0), sl_off(Wr * W * nq, 0), sl_tot(Wr, 0);   // per rank and (window, query): slice start / length (extension elements) / packed byte offset
  q.any_dist = false;
  for (size_t i = 0; i <= W; i++) q.any_dist = q.any_dist || rounds[i]->dist;
  for (size_t i = 0; i < W; i++) {
    Round* pr = rounds[i];
    if (!pr->dist) continue;
    const size_t Sh = pr->S / 2;
    for (size_t r = 0; r < Wr; r++) for (int j = 0; j < nq; j++) {
      const size_t lo_h = r * Sh, hi_h = lo_h + Sh, q_ = q.qlen[i];
      size_t t1 = 2 * (hi_h - 1); if (t1 > q_) t1 = q_;
      size_t t0 = 2 * ((lo_h > 1 ? lo_h : 1) - 1); if (t0 > t1) t0 = t1;
      const size_t k = (r * W + i) * nq + j;
      sl_t0[k] = t0; sl_len[k] = t1 - t0; sl_off[k] = sl_tot[r]; sl_tot[r] += (t1 - t0) * E * 8;
    }
  }
  size_t pack_max = 0;
  for (size_t r = 0; r < Wr; r++) if (sl_tot[r] > pack_max) pack_max = sl_tot[r];
  if (q.any_dist && (d_pack.ensure(pack_max + 64) || d_carry.ensure(((W + 1) * nq * 2 + 1) * E * sizeof(T)))) return fail(MS_ERR_NOMEM, "query slices");
  if (q.any_dist && ((W + 1) * nq * 2 * E * sizeof(T) * Wr > xcap || xcap / Wr < 4096)) return fail(MS_ERR_NOMEM, "exchange buffers too small for the query phase");
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
        q.x1h[i * nq + j] = x1;
        const XE X3 = e_from_base<F, E>(F::mul(x1, x1));
        for (int sgn = 0; sgn < 2; sgn++) {
          void* out = nullptr;
          T* h0 = d_h0 + ((i * nq + j) * 2 + sgn) * E;
          if (pr->dist) {
            const size_t Sh = pr->S / 2, lo_h = (size_t)sh_rank * Sh, cl = mm[sgn] <= lo_h ? 0 : (mm[sgn] - lo_h < Sh ? mm[sgn] - lo_h : Sh);
            const size_t qi = q.carry_jobs.size();
            T* slot_carry = d_carry.as<T>() + qi * E;
            size_t out_off = 0;
            if (i < W) {   // the record's coefficient t sits E u64 words after coefficient t - 1: `out` is where coefficient 0 WOULD be in the packed buffer
              const size_t k = ((size_t)sh_rank * W + i) * nq + j;
              out = reinterpret_cast<u64*>(d_pack.as<u8>() + sl_off[k]) - sl_t0[k] * E;
              out_off = (size_t)sgn * E + lo_h * 2 * E;
            }
            CJ cj; memset(&cj, 0, sizeof cj);
            cj.agg_off = (u32)(qi * E); cj.has_first = 0; cj.zA = e_one<F, E>(); cj.zB = e_pow<F, E>(X3, (u64)Sh);
            cj.carry_out = slot_carry; cj.tail_out = nullptr; cj.h0_out = h0; cj.scale = e_zero<F, E>();
            if (cl) {
              SHPlan pl = sh_plan(lpoly(pr), lstride(pr), sgn, 2, cl, X3, out, true, 1, out_off, 2 * E, nullptr, scr, slot_carry, reinterpret_cast<T*>(xs) + qi * E, sh_rank > 0);
              cj.scale = carry_scale(X3, cl, pl.P);
              pl.dist = true;
              shplans.push_back(pl);
              scr += sh_scratch_elems(cl);
            }
            q.carry_jobs.push_back(cj);
            continue;
          }
          if (i < W) out = blob + q.rec_off[i * nq + j] + (6 * E + 1) * 8;
          shplans.push_back(sh_plan(pr->poly.template as<T>(), pr->cap, sgn, 2, mm[sgn], X3, out, true, 1, (size_t)sgn * E, 2 * E, h0, scr));
          scr += sh_scratch_elems(mm[sgn]);
        }
      }
    }
    int max_nl = 1;
    for (auto& pl : shplans) if (pl.nl > max_nl) max_nl = pl.nl;
    // group by level count so that every launch is homogeneous: aggregates bottom-up (the ranks' top-level aggregates last), finals top-down
    auto add = [&](std::vector<SHJ>& t, int mode, bool part) { if (!t.empty()) { q.tables.push_back(t); q.table_mode.push_back(mode); q.table_part.push_back(part ? 1 : 0); } };
    for (int nl = 1; nl <= max_nl; nl++) {
      for (int l = 0; l + 1 < nl; l++) { std::vector<SHJ> t; bool ad = true; for (auto& pl : shplans) if (pl.nl == nl) { t.push_back(pl.agg[l]); ad = ad && pl.dist; } add(t, 0, ad); }
      { std::vector<SHJ> t; for (auto& pl : shplans) if (pl.nl == nl && pl.has_top_agg) t.push_back(pl.top_agg); add(t, 0, true); }
    }
    q.n_agg_tables = q.tables.size();
    for (int nl = 1; nl <= max_nl; nl++)
      for (int l = nl - 1; l >= 0; l--) { std::vector<SHJ> t; bool ad = true; for (auto& pl : shplans) if (pl.nl == nl) { t.push_back(pl.fin[l]); ad = ad && pl.dist; } add(t, 1, ad); }
  }
  // ---- the slices' way into the blob: chunks of at most xcap / world bytes per rank and exchange
  if (q.any_dist && pack_max) {
    size_t Cb = (pack_max < ((xcap / Wr) & ~(size_t)63)) ? pack_max : ((xcap / Wr) & ~(size_t)63);
    if (shard_gather_chunk && shard_gather_chunk < Cb) Cb = shard_gather_chunk;   // MS_SHARD_GATHER_CHUNK (tests): several exchanges at small sizes
    for (size_t c0 = 0; c0 < pack_max; c0 += Cb) {
      const size_t len = pack_max - c0 < Cb ? pack_max - c0 : Cb;
      q.chunk_c0.push_back(c0); q.chunk_len.push_back(len); q.unp_first.push_back(q.unp.size());
      for (size_t r = 0; r < Wr; r++) for (size_t i = 0; i < W; i++) for (int j = 0; j < nq; j++) {
        const size_t k = (r * W + i) * nq + j, b0 = sl_off[k], b1 = b0 + sl_len[k] * E * 8;
        const size_t p0 = b0 > c0 ? b0 : c0, p1 = b1 < c0 + len ? b1 : c0 + len;
        if (p0 >= p1) continue;
        q.unp.push_back(msmerkle::CopyJob{xr + r * len + (p0 - c0), blob + q.rec_off[i * nq + j] + (6 * E + 1) * 8 + sl_t0[k] * E * 8 + (p0 - b0), p1 - p0});
      }
      q.unp_cnt.push_back(q.unp.size() - q.unp_first.back());
    }
  }
  // ---- find-first and path jobs
  typedef typename QueryPlan::FJ FJ;
  typedef typename QueryPlan::PJ PJ;
  // Sharded proof (ms_set_shard): paths are staged in the exchange buffer — every byte written by exactly one rank
  // (replicated rounds: rank 0), summed over the ranks, then copied into the blob.
  typedef typename QueryPlan::SPJ SPJ;
  const bool shard = sh_on;
  q.fjobs.resize(W);
  q.stage_bytes = 0;
  for (size_t i = 0; i < W; i++) {
    Round* pr = rounds[i];
    const size_t nlev = pr->ts.levels - 1, path_bytes = 8 + 2 * E * 8 + 8 + nlev * 2 * 32;
    if (pr->ts.sharded) {
      q.fjobs[i] = FJ{pr->cw.template as<T>(), 2 * pr->m, 2 * pr->m, d_tg + i * 2 * nq * E, 2 * nq, d_ix + i * 2 * nq, 2, (u32)sh_world, (u32)sh_rank, pr->m};
    } else q.fjobs[i] = FJ{pr->cw.template as<T>(), pr->D, pr->D, d_tg + i * 2 * nq * E, 2 * nq, d_ix + i * 2 * nq, 0, 0, 0, 0};
    for (int t = 0; t < 2 * nq; t++) {
      u8* dst = blob + q.path_off[(i * nq + t / 2) * 2 + (t & 1)];
      u8* out = dst;
      if (shard) { out = xs + q.stage_bytes; q.cjobs.push_back(msmerkle::CopyJob{out, dst, path_bytes}); q.stage_bytes += path_bytes; }
      if (pr->ts.sharded)
        q.sjobs.push_back(SPJ{pr->cw.template as<T>(), 2 * pr->m, pr->m, pr->nodes.template as<u32>(), pr->nodes.template as<u32>() + (2 * pr->ts.Mloc - 1) * 8, pr->ts.Mloc,
                              2, (u32)sh_world, (u32)sh_rank, (u32)nlev, d_ix + i * 2 * nq + t, out});
      else if (!shard || sh_rank == 0)
        q.pjobs.push_back(PJ{pr->cw.template as<T>(), pr->D, pr->nodes.template as<u32>(), pr->D, 2, 2, (u32)nlev, d_ix + i * 2 * nq + t, out});
    }
  }
  if (shard && (q.stage_bytes > xcap || W * nq * 2 * 8 > xcap)) return fail(MS_ERR_NOMEM, "exchange buffers too small for the query phase");
  // ---- one upload
  size_t bytes = 0;
  q.toff.assign(q.tables.size(), 0);
  for (size_t k = 0; k < q.tables.size(); k++) { q.toff[k] = bytes; bytes += q.tables[k].size() * sizeof(SHJ); }
  q.off_f = bytes; bytes += q.fjobs.size() * sizeof(FJ);
  q.off_p = bytes; bytes += q.pjobs.size() * sizeof(PJ);
  q.off_sp = bytes; bytes += q.sjobs.size() * sizeof(SPJ);
  q.off_cj = bytes; bytes += q.cjobs.size() * sizeof(msmerkle::CopyJob);
  q.off_rec = bytes; bytes += q.rec_off.size() * sizeof(size_t);
  q.off_ql = bytes; bytes += q.qlen.size() * 8;
  q.off_x1 = bytes; bytes += q.x1h.size() * sizeof(T);
  bytes = (bytes + 15) & ~(size_t)15;
  q.off_cr = bytes; bytes += q.carry_jobs.size() * sizeof(CJ);
  q.off_un = bytes; bytes += q.unp.size() * sizeof(msmerkle::CopyJob);
  u8* tab;
  RQ(tabs_host(bytes + 8, &tab));
  for (size_t k = 0; k < q.tables.size(); k++) memcpy(tab + q.toff[k], q.tables[k].data(), q.tables[k].size() * sizeof(SHJ));
  if (!q.fjobs.empty()) memcpy(tab + q.off_f, q.fjobs.data(), q.fjobs.size() * sizeof(FJ));
  if (!q.pjobs.empty()) memcpy(tab + q.off_p, q.pjobs.data(), q.pjobs.size() * sizeof(PJ));
  if (!q.sjobs.empty()) memcpy(tab + q.off_sp, q.sjobs.data(), q.sjobs.size() * sizeof(SPJ));
  if (!q.cjobs.empty()) memcpy(tab + q.off_cj, q.cjobs.data(), q.cjobs.size() * sizeof(msmerkle::CopyJob));
  if (!q.rec_off.empty()) memcpy(tab + q.off_rec, q.rec_off.data(), q.rec_off.size() * sizeof(size_t));
  memcpy(tab + q.off_ql, q.qlen.data(), q.qlen.size() * 8);
  memcpy(tab + q.off_x1, q.x1h.data(), q.x1h.size() * sizeof(T));
  if (!q.carry_jobs.empty()) memcpy(tab + q.off_cr, q.carry_jobs.data(), q.carry_jobs.size() * sizeof(CJ));
  if (!q.unp.empty()) memcpy(tab + q.off_un, q.unp.data(), q.unp.size() * sizeof(msmerkle::CopyJob));
  if (d_tabs.ensure(bytes + 8)) return fail(MS_ERR_NOMEM, "query tables");
  CK(msrt::h2d(d_tabs.p, tab, bytes + 8, stream));   // page-locked source: no synchronisation here (this stage ends with one; the area is next written by the next proof)
  q.dt = d_tabs.as<u8>();
  return 0;
}

// 3. the quotient scans: every aggregate launch, (sharded proofs) the carry exchange, every final launch
template <class F>
int Ctx<F>::query_scans(QueryPlan& q) {
  typedef typename QueryPlan::CJ CJ;
  const size_t ncarry = q.carry_jobs.size();
  auto carries = [&]() -> int {   // every aggregate is in the send buffer: one all-gather, then the carry-ins of this rank's jobs and every job's H_0
    RQ(exchange(MS_XCHG_ALL_GATHER, ncarry * E * sizeof(T)));
    return shard_carry_table(reinterpret_cast<const CJ*>(q.dt + q.off_cr), ncarry);
  };
  if (ncarry) CK(msrt::memset_dev(xs, 0, ncarry * E * sizeof(T), stream));   // aggregates of the jobs this rank has no coefficients for
  for (size_t k = 0; k < q.tables.size(); k++) {
    if (k == q.n_agg_tables && ncarry) RQ(carries());
    size_t maxnb = 1;
    for (auto& j : q.tables[k]) { size_t nb = j.m ? (j.m + mspoly::SH_BS - 1) / mspoly::SH_BS : 1; if (nb > maxnb) maxnb = nb; }
    PartScopeIf part(this, q.table_part[k] != 0);
    RQ(sh_launch_table(reinterpret_cast<const SHJ*>(q.dt + q.toff[k]), q.tables[k].size(), maxnb, q.table_mode[k]));
  }
  if (q.n_agg_tables == q.tables.size() && ncarry) RQ(carries());   // (no job on this rank at all: the exchange still has to happen - the other ranks are in it)
  return 0;
}

// 4. sharded proofs: the ranks' slices of the quotient polynomials into the blob (all-gathered, or gathered to rank 0)
template <class F>
int Ctx<F>::query_shard_assembly(QueryPlan& q) {
  for (size_t c = 0; c < q.chunk_c0.size(); c++) {
    CK(msrt::d2d(xs, d_pack.as<u8>() + q.chunk_c0[c], q.chunk_len[c], stream));
    RQ(exchange(proof_root_only ? MS_XCHG_GATHER : MS_XCHG_ALL_GATHER, q.chunk_len[c]));
    if (q.unp_cnt[c] && (!proof_root_only || sh_rank == 0)) {
      size_t maxw = 1;
      for (size_t u = 0; u < q.unp_cnt[c]; u++) { const size_t wgs = (q.unp[q.unp_first[c] + u].bytes / 8 + msmerkle::CopyRangesKernel::WORDS - 1) / msmerkle::CopyRangesKernel::WORDS; if (wgs > maxw) maxw = wgs; }
      msmerkle::CopyRangesKernel::Params up{reinterpret_cast<const msmerkle::CopyJob*>(q.dt + q.off_un) + q.unp_first[c], (u32)q.unp_cnt[c]};
      CK(run<msmerkle::CopyRangesKernel>(K_IO, (unsigned)maxw, (unsigned)q.unp_cnt[c], msmerkle::CopyRangesKernel::THREADS, 0, up));
    }
  }
  return 0;
}

// 5. the opened points (fri.rs:148-154), the leaf lookup BY VALUE (merkle.rs:216-225, quirk Q7) and the Merkle paths (merkle.rs:230-288); ends with the stage's one
// stream synchronisation
template <class F>
int Ctx<F>::query_openings(QueryPlan& q) {
  typedef typename QueryPlan::FJ FJ;
  typedef typename QueryPlan::PJ PJ;
  typedef typename QueryPlan::SPJ SPJ;
  const size_t W = q.W; const int nq = q.nq; const u8* dt = q.dt;
  const bool shard = sh_on;
  if (!W) { CK(msrt::sync(stream)); return 0; }
  typename mspoly::QueryPointsKernel<F, E>::Params qp{q.d_h0, reinterpret_cast<const T*>(dt + q.off_x1), reinterpret_cast<const u64*>(dt + q.off_ql), (int)W, nq, q.blob,
                                                     reinterpret_cast<const size_t*>(dt + q.off_rec), q.d_tg};
  CK(run<mspoly::QueryPointsKernel<F, E>>(K_QUERY_POINTS, grid1(W * nq, 64), 1, 64, 0, qp));
  CK(msrt::memset_dev(q.d_ix, 0xFF, W * nq * 2 * 8, stream));
  // first match: big codewords one launch each, the rest batched
  size_t first_small = W;
  for (size_t i = 0; i < W; i++) if (rounds[i]->D <= ((size_t)1 << 16)) { first_small = i; break; }
  for (size_t i = 0; i < first_small; i++) {
    typename mspoly::FindFirstKernel<F, E>::Params fp; fp.jobs = nullptr; fp.inline_job = q.fjobs[i];
    PartScopeIf part(this, rounds[i]->ts.sharded);
    CK(run<mspoly::FindFirstKernel<F, E>>(K_FIND_FIRST, grid1(rounds[i]->ts.sharded ? 2 * rounds[i]->m : rounds[i]->D, mspoly::THREADS), 1, mspoly::THREADS, 0, fp));
  }
  if (first_small < W) {
    typename mspoly::FindFirstKernel<F, E>::Params fp; fp.jobs = reinterpret_cast<const FJ*>(dt + q.off_f) + first_small; fp.inline_job = q.fjobs[first_small];
    CK(run<mspoly::FindFirstKernel<F, E>>(K_FIND_FIRST, grid1(rounds[first_small]->D, mspoly::THREADS), (unsigned)(W - first_small), mspoly::THREADS, 0, fp));
  }
  if (shard) {  // first match over ALL ranks' parts: minimum global index (replicated rounds: every rank holds the same value)
    CK(msrt::d2d(xs, q.d_ix, W * nq * 2 * 8, stream));
    RQ(exchange(MS_XCHG_ALL_REDUCE_MIN_U64, W * nq * 2 * 8));
    CK(msrt::d2d(q.d_ix, xs, W * nq * 2 * 8, stream));
    CK(msrt::memset_dev(xs, 0, q.stage_bytes, stream));
  }
  if (!q.pjobs.empty()) {
    typename msmerkle::PathKernel<F, E>::Params pk{reinterpret_cast<const PJ*>(dt + q.off_p), (u32)q.pjobs.size()};
    CK(run<msmerkle::PathKernel<F, E>>(K_PATH, (unsigned)q.pjobs.size(), 1, 64, 0, pk));   // one wave per opening
  }
  if (!q.sjobs.empty()) {
    typename msmerkle::ShardPathKernel<F, E>::Params sk{reinterpret_cast<const SPJ*>(dt + q.off_sp), (u32)q.sjobs.size()};
    CK(run<msmerkle::ShardPathKernel<F, E>>(K_PATH, grid1(q.sjobs.size(), 64), 1, 64, 0, sk));
  }
  if (shard) {
    RQ(exchange(MS_XCHG_ALL_REDUCE_SUM_U8, q.stage_bytes));
    msmerkle::CopyJobsKernel::Params ck{reinterpret_cast<const msmerkle::CopyJob*>(dt + q.off_cj), (u32)q.cjobs.size()};
    CK(run<msmerkle::CopyJobsKernel>(K_PATH, (unsigned)q.cjobs.size(), 1, msmerkle::CopyJobsKernel::THREADS, 0, ck));
  }
  CK(msrt::d2h(pinned, q.d_ix, W * nq * 2 * 8, stream));
  CK(msrt::sync(stream));
  const unsigned long long* hidx = reinterpret_cast<const unsigned long long*>(pinned);
  for (size_t t = 0; t < W * nq * 2; t++) if (hidx[t] == ~0ULL) return fail(MS_ERR_LEAF_NOT_FOUND, "leaf is not included in the tree");
  return 0;
}

// ------------------------------------------------------------------ fri.rs:115-189
// ext_out != nullptr (ms_fri_query_into): the query-phase kernels write the MSFP blob straight into the caller's buffer - page-locked host
// memory (ms_pinned_alloc; hipHostMalloc memory is mapped into the device's address space) or device memory - instead of d_blob
template <class F>
int Ctx<F>::fri_query(const u64* betas, int nq, u8* ext_out, size_t ext_cap, size_t* ext_len) {
  if (nrounds_done != fri_rounds || fri_rounds == 0) return fail(MS_ERR_STATE, "fri_query before the commit phase finished");
  if (!betas || nq < 1) return fail(MS_ERR_ARG, "fri_query");
  QueryPlan q;
  RQ(query_layout(q, nq));
  const size_t W = q.W, pos = q.pos;
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
  q.blob = ext_out ? ext_out : d_blob.as<u8>();
  blob_external = ext_out != nullptr;
  q.d_h0 = d_targets.as<T>();
  q.d_tg = q.d_h0 + n_h0;
  q.d_ix = d_idx.as<unsigned long long>();
  RQ(query_build(q, betas));
  RQ(query_scans(q));
  RQ(query_shard_assembly(q));
  RQ(query_openings(q));
  blob_size = (proof_root_only && sh_world > 1 && sh_rank != 0) ? 0 : pos;   // ms_shard_proof_on_root: the blob is whole on rank 0 only
  return MS_OK;
}

// src/merkle.rs:272-288 on a standalone binary tree: leaves de-interleaved to SoA limbs, then the same
// LeafHash / InnerHash / FindFirst / MerklePath kernels the FRI query phase uses
template <class F> template <int EL>
int Ctx<F>::merkle_prove_t(const u64* leafs, size_t leaf_num, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) {
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
    if (e) rc = fail_rt(e, "leaf upload");
    else rc = transpose_in(d_io.as<u64>(), ds.as<T>(), leaf_num, (size_t)EL, F::from_u64(1), 0, nullptr);
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

template <class F>
int Ctx<F>::merkle_prove(const u64* leafs, size_t leaf_num, int ext, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len) {
  if (!leafs || !leaf) return fail(MS_ERR_ARG, "null argument");
  if (ext == 1) return merkle_prove_t<1>(leafs, leaf_num, lpn, leaf, out, cap, len);
  if (ext == E) return merkle_prove_t<E>(leafs, leaf_num, lpn, leaf, out, cap, len);
  return fail(MS_ERR_ARG, "ext must be 1 or the field's extension degree");
}

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::query_layout(Ctx<FF>::QueryPlan& q, int nq); \
  template int Ctx<FF>::query_build(Ctx<FF>::QueryPlan& q, const u64* betas); \
  template int Ctx<FF>::query_scans(Ctx<FF>::QueryPlan& q); \
  template int Ctx<FF>::query_shard_assembly(Ctx<FF>::QueryPlan& q); \
  template int Ctx<FF>::query_openings(Ctx<FF>::QueryPlan& q); \
  template int Ctx<FF>::fri_query(const u64* betas, int nq, u8* ext_out, size_t ext_cap, size_t* ext_len); \
  template int Ctx<FF>::merkle_prove(const u64* leafs, size_t leaf_num, int ext, size_t lpn, const u64* leaf, u8* out, size_t cap, size_t* len);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

}  // namespace msctx
