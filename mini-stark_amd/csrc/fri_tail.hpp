// fri_tail.hpp — one FRI round of the latency-bound tail as ONE launch (r05; SURVEY 7 hard part (iv)).
//
// A late round of Fri::commit_phase (fri.rs:96-109) is a few thousand elements: until r04 its ms_fri_fold_commit was ten launches - Fold, SuffixHorner, Degree,
// FriFoldEval, LeafHash, PadOnlyBlock, InnerSubtree (x2) - each 4-18 us long for under a microsecond of arithmetic, one behind the other (3.6 us of dispatch latency
// per dependent launch: tools/latency_probe.hip), on the critical path of a proof alone on the GPU and of every rank of a sharded proof (replicated work).
// Here the same device code runs as the steps of ONE kernel:
//
//   workgroup 0            the coefficient domain:  folded = even + alpha odd  ->  quotient (folded - B(alpha)) / (x - z) by suffix Horner = the next round
//                          polynomial  ->  its trimmed length
//   workgroups 1 .. G      the evaluation domain, 256 leaf groups each: codeword of the next round pointwise from the previous codeword (FriFoldEval: no
//                          transform)  ->  leaf digests (pad-only blocks in place)  ->  the eight tree levels above them, children in LDS
//   the LAST one to finish the levels above the G subtree roots, and root + trimmed length into page-locked host memory
//
// The two sides are independent (the codeword never reads the quotient), so they run side by side.  The hand-over to the last workgroup is the classic
// completion counter: every workgroup's stores are released (device-scope fence) before its thread 0 bumps the counter, the workgroup that reads G back
// acquires and goes on - no workgroup ever waits for another, so the grid drains whatever the scheduling.  Every step is the device function of the
// stand-alone kernel (mspoly::FoldKernel, SuffixHornerKernel, DegreeKernel, FriFoldEvalKernel, msmerkle::LeafHashKernel, InnerSubtreeKernel): same
// arithmetic, same bytes - the parity suite runs with the fused round forced on and off (MS_FRI_TAIL_MAX).
#pragma once
#include "merkle.hpp"
#include "poly.hpp"

namespace msfri {

template <class F, int E> struct FriTailKernel {
  typedef typename F::T T;
  static constexpr int THREADS = 256;
  static constexpr int WG_GROUPS = 256;                 // leaf groups per evaluation-side workgroup (one per thread; 8 tree levels per workgroup)
  typedef mspoly::FoldKernel<F, E> FoldK;
  typedef mspoly::SuffixHornerKernel<F, E> ScanK;
  typedef mspoly::DegreeKernel<F, E> DegK;
  typedef mspoly::FriFoldEvalKernel<F, E, 2> EvalK;      // two outputs per thread: a workgroup's 512 outputs = its 256 leaf groups
  typedef msmerkle::LeafHashKernel<F, E> LeafK;
  typedef msmerkle::InnerSubtreeKernel TreeK;
  static_assert(FoldK::THREADS == THREADS && ScanK::THREADS == THREADS && DegK::THREADS == THREADS && EvalK::THREADS == THREADS && LeafK::THREADS == THREADS &&
                TreeK::THREADS == THREADS, "one workgroup shape for every step");
  struct Params {
    // ---- workgroup 0
    u32 do_coef;                          // 0: fewer than two folded coefficients - no quotient (the next round polynomial is zero, its length word stays 0)
    typename FoldK::Params fold;          // previous round polynomial -> folded (global scratch)
    typename ScanK::Params scan;          // ONE block, final mode, inline job: folded -> quotient = next round polynomial
    typename DegK::Params deg;            // trimmed length of the quotient into the (zero) device word `top.aux_src`
    // ---- workgroups 1 .. G
    u32 G;
    typename EvalK::Params eval;          // previous codeword -> next codeword
    typename LeafK::Params leaf;          // ovf == nullptr
    msmerkle::InnerHashParams sub;        // the levels above a workgroup's leaf digests (nlevels = log2 of its leaf groups; no host forwarding)
    // ---- the last workgroup
    msmerkle::InnerHashParams top;        // G > 1: the levels above the G subtree roots, root and length word forwarded to the host.  G == 1: only nodes / host_root / aux_* are used
    size_t root_index;                    // node index of the root
    u32* done;                            // completion counter (zero at launch; the last workgroup leaves it zero)
  };
  static MS_HD size_t lds_bytes() {
    size_t a = ScanK::lds_bytes(), b = LeafK::lds_bytes(), c = TreeK::lds_bytes();
    return (a > b ? (a > c ? a : c) : (b > c ? b : c)) + 16;
  }
  static MS_DEV void run(const Params& p, int bx, int, int, int tid, unsigned char* lds) {
    if (bx == 0) {
      if (p.do_coef) {
        const size_t m = (p.fold.n + 1) / 2;
        for (int k = 0; (size_t)k * THREADS < m; k++) FoldK::phase(0, p.fold, k, 0, tid, THREADS, lds);
        msrt::wg_barrier_global();                              // the scan loads what other waves folded
        ScanK::run(p.scan, 0, 0, 1, tid, lds);
        msrt::wg_barrier_global();                              // the length scan reads the quotient other waves stored
        for (int k = 0; (size_t)k * THREADS < p.deg.n; k++) DegK::phase(0, p.deg, k, 0, tid, THREADS, lds);
      }
    } else {
      const int b = bx - 1;
      EvalK::phase(0, p.eval, b, 0, tid, THREADS, lds);
      msrt::wg_barrier_global();                                // a leaf group's two elements come from two other threads
      LeafK::phase(0, p.leaf, b, 0, tid, THREADS, lds);
      msrt::wg_barrier_global();                                // level 0 of the subtree reads the digests from global memory; the leaf buffers in LDS are free
      if (p.sub.nlevels) TreeK::run(p.sub, b, 0, (int)p.G, tid, lds);
    }
    // ---- completion: G + 1 workgroups, the last one finishes the tree
    msrt::wg_barrier_global();                                  // every thread's stores of this workgroup are complete
    u32* flag = reinterpret_cast<u32*>(lds);
    if (tid == 0) {
      msrt::fence_device();                                     // release
      const u32 before = msrt::atomic_add_u32(p.done, 1u);
      *flag = (before == p.G) ? 1u : 0u;
      if (before == p.G) { msrt::fence_device(); *p.done = 0; } // acquire; and the counter is zero again for the next round's launch
    }
    msrt::wg_barrier();
    const bool last = *flag != 0;
    msrt::wg_barrier();                                         // (the flag word is part of the tree buffers below)
    if (!last) return;
    msrt::fence_device();
    if (p.G > 1) { TreeK::run(p.top, 0, 0, 1, tid, lds); return; }
    // G == 1: the subtree root (or the lone leaf digest) is the tree's root - forward it, and the length word, to the host
    if (tid == 0) {
      for (int k = 0; k < 8; k++) p.top.host_root[k] = p.top.nodes[p.root_index * 8 + (size_t)k];
      if (p.top.aux_src) { *p.top.aux_dst = *p.top.aux_src; *p.top.aux_src = 0; }
      msrt::raise_host_flag(p.top.flag);
    }
  }
};

}  // namespace msfri
