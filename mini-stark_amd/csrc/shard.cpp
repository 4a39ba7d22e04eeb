// shard.cpp — one proof over several ranks: the exchange primitive (RCCL inside the library, or the caller's callback) and ms_set_shard*.
#include "ctx.hpp"

namespace msctx {

template <class F>
int Ctx<F>::exchange(int op, size_t bytes) {
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
  if (!xfn) {
    // MS_SHARD_STUB (tools/shard_rank_probe.py: what ONE rank of W computes, measured on one GPU): no peers exist - every collective returns this rank's own payload in
    // every peer's place, as stream-ordered device copies, no host synchronisation (like the RCCL path).  Right sizes, wrong values: the proof is garbage by construction.
    const size_t W = (size_t)sh_world;
    if (op == MS_XCHG_ALL_TO_ALL) CK(msrt::d2d(xr, xs, bytes * W, stream));
    else if (op == MS_XCHG_ALL_GATHER || op == MS_XCHG_GATHER) { for (size_t r = 0; r < W; r++) CK(msrt::d2d(xr + r * bytes, xs, bytes, stream)); }
    return 0;   // (the all-reduces leave the buffer as it is)
  }
  CK(msrt::sync(stream));
  if (xfn(xuser, op, bytes)) return fail(MS_ERR_HIP, "exchange callback failed");
  return 0;
}

template <class F>
int Ctx<F>::exchange_slice(size_t off, size_t stride, size_t bytes, int sl, int S) {
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
  if (!xfn) { for (int r = 0; r < sh_world; r++) CK(msrt::d2d(xr + off + (size_t)r * stride, xs + off + (size_t)r * stride, bytes, stream)); return 0; }   // (stub world: see exchange)
  CK(msrt::sync(stream));
  xl_off = off; xl_stride = stride;
  if (xfn(xuser, MS_XCHG_ALL_TO_ALL_SLICE, bytes)) return fail(MS_ERR_HIP, "exchange callback failed");
  return 0;
}

template <class F>
void Ctx<F>::drop_rccl() {
  if (rccl_comm) { msrt::sync(stream); if (comm_stream) msrt::sync(comm_stream); msrt::Rccl::get().comm_destroy(rccl_comm); rccl_comm = nullptr; }   // (work may be queued on either stream)
  rccl_send.release(); rccl_recv.release();
}

template <class F>
int Ctx<F>::set_shard_rccl(int rank, int world, const u8* unique_id, size_t cap) {
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
template <class F>
int Ctx<F>::rccl_selftest() {
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

template <class F>
int Ctx<F>::set_shard(int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user) {
  if (world < 1 || !is_pow2((u64)world) || rank < 0 || rank >= world) return fail(MS_ERR_ARG, "set_shard: world must be a power of two and 0 <= rank < world");
  if (world > 1 && (!d_send || !d_recv || (!fn && !shard_stub) || cap < 4096)) return fail(MS_ERR_ARG, "set_shard: exchange buffers / callback missing");
  drop_rccl();
  sh_rank = rank; sh_world = world; xs = reinterpret_cast<u8*>(d_send); xr = reinterpret_cast<u8*>(d_recv); xcap = cap; xfn = fn; xuser = user;
  sh_on = world > 1 || (allow_w1 && d_send && d_recv && fn && cap >= 4096);
  have_lde = false; nrounds_done = 0; blob_size = 0;
  return MS_OK;
}

// the members this unit defines, for both fields (the other units see declarations only)
#define MS_INSTANTIATE(FF) \
  template int Ctx<FF>::exchange(int op, size_t bytes); \
  template int Ctx<FF>::exchange_slice(size_t off, size_t stride, size_t bytes, int sl, int S); \
  template void Ctx<FF>::drop_rccl(); \
  template int Ctx<FF>::set_shard_rccl(int rank, int world, const u8* unique_id, size_t cap); \
  template int Ctx<FF>::rccl_selftest(); \
  template int Ctx<FF>::set_shard(int rank, int world, void* d_send, void* d_recv, size_t cap, ms_exchange_fn fn, void* user);
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

}  // namespace msctx
