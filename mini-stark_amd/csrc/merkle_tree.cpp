// merkle_tree.cpp — MerkleTree::new (merkle.rs:81-177) on the SHA-256 kernels of merkle.hpp: replicated, sharded by leaf-group residue (digest all-to-all) and
// sharded by contiguous range; ms_merkle_commit.
#include "ctx.hpp"

namespace msctx {

// src/merkle.rs:89-118 (shape checks and node count)
template <class F>
int Ctx<F>::tree_shape(size_t leaf_num, size_t lpn, size_t ic, TreeShape* ts) {
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
template <class F> template <int EL>
int Ctx<F>::leaf_hash(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, size_t lpn, size_t ngroups, u32* out,
              size_t g_first, u32 run_len, u32 run_stride, const msmerkle::LinColSpec* lin, size_t out_g0) {
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

template <class F> template <int EL>
int Ctx<F>::tree_build(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, const TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin) {
  if (nodes.ensure(ts.nodes * 32)) return fail(MS_ERR_NOMEM, "merkle nodes");
  const size_t ngroups = ts.leaf_num / ts.lpn;
  RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, ngroups, nodes.as<u32>(), 0, 0, 0, lin)));
  RQ(inner_levels(nodes.as<u32>(), ngroups, ts.ic));
  return 0;
}

template <class F>
int Ctx<F>::inner_levels(u32* nodes, size_t nchildren, size_t ic, bool final_levels, u8* rec_out) {
  size_t child_off = 0;
  if (final_levels) root_on_host = false;
  // rec_out (a subtree of a sharded tree, final_levels false): the launch that produces the subtree's root also stores the rank's record of the root all-gather -
  // root at rec_out, the aux word (shard_aux, if any) 32 bytes behind it - instead of a fill and two copy launches behind the subtree (r05)
  bool rec_done = false;
  auto to_rec = [&](msmerkle::InnerHashKernel::Params& ip) {
    ip.host_root = reinterpret_cast<u32*>(rec_out);
    if (shard_aux) { ip.aux_src = shard_aux; ip.aux_dst = reinterpret_cast<unsigned long long*>(rec_out + 32); }
    rec_done = true;
  };
  while (nchildren > 1) {
    msmerkle::InnerHashKernel::Params ip;
    ip.nodes = nodes; ip.child_off = child_off; ip.nchildren = nchildren; ip.ic = (u32)ic; ip.host_root = nullptr; ip.aux_src = nullptr; ip.aux_dst = nullptr; ip.flag = msrt::HostFlag{nullptr, 0};
    const size_t nparents = nchildren / ic;
    if (ic == 2 && nparents <= subtree_parents && (nchildren & (nchildren - 1)) == 0) {   // latency-bound levels: up to 9 of them per launch, children in LDS
      typedef msmerkle::InnerSubtreeKernel SK;
      u32 nl = 0; for (size_t m = nchildren; m > 1 && nl < (u32)SK::MAX_LEVELS; m >>= 1) nl++;
      const size_t left = nchildren >> nl;
      if (final_levels && left == 1) {
        RQ(join_side());   // this launch forwards the length word the side stream's scan produces
        ip.host_root = host_root(); root_on_host = true;
        if (pending_aux) { ip.aux_src = pending_aux; ip.aux_dst = reinterpret_cast<unsigned long long*>(pinned); pending_aux = nullptr; aux_on_host = true; }
        ip.flag = arm_flag();
      } else if (rec_out && left == 1) to_rec(ip);
      ip.nlevels = nl;
      next_bytes = (double)nchildren * 32 * 2;
      CK(run_coop<SK>(K_INNER_HASH, (unsigned)left, SK::THREADS, SK::lds_bytes(), ip));
      for (u32 l = 0; l < nl; l++) { child_off += nchildren; nchildren >>= 1; }
      continue;
    }
    if (final_levels && (nparents == 1 || nparents <= (size_t)tree_top_parents)) {
      RQ(join_side());
      ip.host_root = host_root(); root_on_host = true;
      if (pending_aux) { ip.aux_src = pending_aux; ip.aux_dst = reinterpret_cast<unsigned long long*>(pinned); pending_aux = nullptr; aux_on_host = true; }
      ip.flag = arm_flag();   // (the launch below is the tree's last either way: the fused top, or the level whose one parent is the root)
    } else if (rec_out && (nparents == 1 || nparents <= (size_t)tree_top_parents)) to_rec(ip);
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
  if (rec_out && !rec_done) {   // no launch took the record along (a one-digest "subtree"): by copies
    CK(msrt::d2d(rec_out, reinterpret_cast<u8*>(nodes) + (child_off + nchildren - 1) * 32, 32, stream));
    if (shard_aux) CK(msrt::d2d(rec_out + 32, shard_aux, 8, stream)); else CK(msrt::memset_dev(rec_out + 32, 0, 8, stream));
  }
  return 0;
}

// Sharded MerkleTree::new over a binary tree of M = leaf_num/lpn leaf groups, of which this rank hashes the groups
// j = rank + W*i found at local group index i of the view (base, strides): digest all-to-all, subtree, root all-gather, top.
template <class F> template <int EL>
int Ctx<F>::tree_build_sharded(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes, const msmerkle::LinColSpec* lin) {
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

template <class F>
int Ctx<F>::finish_sharded_tree(TreeShape& ts, DevBuf& nodes, size_t Mloc) {
  const size_t W = (size_t)sh_world, sub_nodes = 2 * Mloc - 1, top_nodes = 2 * W - 1;
  constexpr size_t REC = msmerkle::ShardTopTreeKernel::REC;
  if (W * REC > xcap) return fail(MS_ERR_NOMEM, "exchange buffers too small for the subtree roots");
  // the subtree; its last launch leaves this rank's record (root | aux word) in the send buffer.  (Bytes 40 .. 64 of a record are never read.)
  { PartScope part(this); RQ(inner_levels(nodes.as<u32>(), Mloc, 2, false, xs)); }
  RQ(exchange(MS_XCHG_ALL_GATHER, REC));
  // the W records -> the top's leaves, the maximum aux word, and the levels above them, in one launch; root (and the length word riding on the commitment) to the host
  u8* top = nodes.as<u8>() + sub_nodes * 32;
  typedef msmerkle::ShardTopTreeKernel TT;
  if (W > ((size_t)1 << msmerkle::InnerSubtreeKernel::MAX_LEVELS)) return fail(MS_ERR_STATE, "sharded tree: more ranks than the top kernel's levels");
  typename TT::Params tk;
  memset(&tk, 0, sizeof tk);
  tk.recs = xr; tk.W = (u32)W; tk.aux_max = shard_aux;
  tk.tree.nodes = reinterpret_cast<u32*>(top); tk.tree.child_off = 0; tk.tree.nchildren = W; tk.tree.ic = 2;
  u32 tl = 0; while (((size_t)1 << tl) < W) tl++;
  tk.tree.nlevels = tl;
  RQ(join_side());
  tk.tree.host_root = host_root(); root_on_host = true;
  if (pending_aux) { tk.tree.aux_src = pending_aux; tk.tree.aux_dst = reinterpret_cast<unsigned long long*>(pinned); pending_aux = nullptr; aux_on_host = true; }
  tk.tree.flag = arm_flag();
  shard_aux = nullptr;
  CK(run_coop<TT>(K_INNER_HASH, 1, TT::THREADS, TT::lds_bytes(), tk));
  ts.sharded = true; ts.Mloc = Mloc; ts.local_nodes = sub_nodes + top_nodes;
  return 0;
}

// MerkleTree::new over data EVERY rank holds (the raw trace): rank k hashes the contiguous leaf groups [k*M/W, (k+1)*M/W) - no digest exchange at all -
// builds that subtree, and the ranks all-gather the W subtree roots
template <class F> template <int EL>
int Ctx<F>::tree_build_sharded_contiguous(const T* base, size_t col_stride, size_t row_stride, size_t limb_stride, u32 width, TreeShape& ts, DevBuf& nodes) {
  const size_t W = (size_t)sh_world, M = ts.leaf_num / ts.lpn, Mloc = M / W;
  if (ts.ic != 2 || Mloc == 0 || (Mloc >> 32)) return fail(MS_ERR_STATE, "sharded tree needs a binary tree with at least world leaf groups");
  if (nodes.ensure((2 * Mloc - 1 + 2 * W - 1) * 32)) return fail(MS_ERR_NOMEM, "merkle nodes");
  // group g of the launch is leaf group rank*Mloc + g (one run of Mloc groups); its digest lands at nodes[g] (out_g0 = the rank's first group)
  PartScope part(this);
  RQ((leaf_hash<EL>(base, col_stride, row_stride, limb_stride, width, ts.lpn, Mloc, nodes.as<u32>(), (size_t)sh_rank * Mloc, (u32)Mloc, 0, nullptr, (size_t)sh_rank * Mloc)));
  return finish_sharded_tree(ts, nodes, Mloc);
}

// root of the tree built LAST on this context (every caller reads it right behind the build)
template <class F>
int Ctx<F>::read_root(const DevBuf& nodes, const TreeShape& ts, u8* root) {
  const bool on_host = root_on_host;
  if (!on_host) { CK(msrt::d2h(pinned, nodes.as<u8>() + (ts.local_nodes - 1) * 32, 32, stream)); CK(msrt::sync(stream)); }
  else CK(sync_results());   // (polls the flag the tree's last launch raises, if it carried one)
  memcpy(root, on_host ? reinterpret_cast<const void*>(host_root()) : pinned, 32);
  return 0;
}

// ------------------------------------------------------------------ standalone entry points
template <class F>
int Ctx<F>::merkle_commit(const u64* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, u8* nodes_out, size_t cap, size_t* nn, u8* root) {
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

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::tree_shape(size_t leaf_num, size_t lpn, size_t ic, Ctx<FF>::TreeShape* ts); \
  template int Ctx<FF>::inner_levels(u32* nodes, size_t nchildren, size_t ic, bool final_levels, u8* rec_out); \
  template int Ctx<FF>::finish_sharded_tree(Ctx<FF>::TreeShape& ts, DevBuf& nodes, size_t Mloc); \
  template int Ctx<FF>::read_root(const DevBuf& nodes, const Ctx<FF>::TreeShape& ts, u8* root); \
  template int Ctx<FF>::merkle_commit(const u64* leafs, size_t leaf_num, int ext, size_t lpn, size_t ic, u8* nodes_out, size_t cap, size_t* nn, u8* root);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

// the member templates other units call (trace / LDE commitments: EL = 1; FRI codewords: EL = the extension degree; ms_merkle_prove: both)
#define MS_INSTANTIATE_EL(FF, EL) \
  template int Ctx<FF>::tree_build<EL>(const Ctx<FF>::T*, size_t, size_t, size_t, u32, const Ctx<FF>::TreeShape&, DevBuf&, const msmerkle::LinColSpec*); \
  template int Ctx<FF>::tree_build_sharded<EL>(const Ctx<FF>::T*, size_t, size_t, size_t, u32, Ctx<FF>::TreeShape&, DevBuf&, const msmerkle::LinColSpec*);
MS_INSTANTIATE_EL(GL, 1)
MS_INSTANTIATE_EL(GL, 2)
MS_INSTANTIATE_EL(BB, 1)
MS_INSTANTIATE_EL(BB, 4)
#undef MS_INSTANTIATE_EL
template int Ctx<GL>::tree_build_sharded_contiguous<1>(const Ctx<GL>::T*, size_t, size_t, size_t, u32, Ctx<GL>::TreeShape&, DevBuf&);
template int Ctx<BB>::tree_build_sharded_contiguous<1>(const Ctx<BB>::T*, size_t, size_t, size_t, u32, Ctx<BB>::TreeShape&, DevBuf&);

}  // namespace msctx
