// io.cpp — the boundary's bulk transfers (Stark::prove takes a host trace and returns a host proof, starks.rs:59-69, 160-168): the trace's way in and the FRI
// proof's way out.  Page-locked buffers travel on SDMA copy engines through the HSA runtime the process already runs on (msrt::Sdma, rt.hpp: never a blit kernel
// that takes issue slots from the provers), anything else - and every fall-back - through the HIP runtime's copies.  Failure handling:
//   * an engine that REFUSES a copy, or a runtime that does not know the buffers: this context stays on the HIP runtime's copies from then on;
//   * a copy the engine reports as FAILED (negative completion signal): it is done again through the HIP runtime, the call succeeds if that does;
//   * a copy that does not COMPLETE within MS_SDMA_TIMEOUT_S (20 s): the engine may still be reading or writing - the pending flag stays set, the context is
//     poisoned (MS_ERR_HIP now, MS_ERR_STATE from every later entry point), and ms_destroy waits for the transfer without limit before it frees anything.
#include "ctx.hpp"

namespace msctx {

template <class F>
bool Ctx<F>::sdma_ready() {
  if (!readback_sdma) return false;
  if (sdma_state == 0) {
    msrt::Sdma& S = msrt::Sdma::get();
    if (!S.bind_device(device, &sdma_gpu)) sdma_state = (!S.signal_create(&sdma_sig) && !S.signal_create(&sdma_up_sig)) ? 1 : 2;
    else if (S.unavailable()) sdma_state = 2;   // no HSA runtime to bind: this context stays on the HIP runtime's copies.  (Engines merely busy: asked again at the next copy.)
  }
  return sdma_state == 1;
}

template <class F>
int Ctx<F>::poison(const char* what) {
  poisoned = true;
  err = std::string(what) + ": the copy engine may still be accessing the buffers, the context is poisoned (every entry point returns MS_ERR_STATE; ms_destroy waits for the transfer, then frees)";
  return MS_ERR_HIP;
}

// ------------------------------------------------------------------ the trace, host -> device
template <class F>
int Ctx<F>::upload_wait(const char* what) {
  if (!up.pending) return 0;
  const int wv = msrt::Sdma::get().wait(sdma_up_sig, sdma_timeout_s);
  if (wv == 1) return poison(what);          // (up.pending stays set: the destructor waits for it)
  up.pending = false;
  if (wv < 0) { sdma_state = 2; return 1; }  // the engine reported a failed copy: no more engine copies on this context; the caller redoes this one
  return 0;
}

// ms_trace_upload_async: the copy of a page-locked trace into the device buffer the proof in flight does not use.  A hint: without a usable engine (or for pageable
// memory) nothing is queued and ms_trace_commit uploads as before.  `trace` must stay valid and unchanged until the ms_trace_commit that names it returns.
template <class F>
int Ctx<F>::trace_upload_async(const u64* trace, size_t N_, size_t w_) {
  if (!trace || !N_ || !w_) return fail(MS_ERR_ARG, "trace_upload_async");
  if (up.pending) { const int e = upload_wait("SDMA upload did not complete within the time limit"); if (e < 0) return e; }   // (one in flight per context; a superseded prefetch is simply dropped)
  up.src = nullptr;
  if (!(upload_sdma && sdma_ready() && msrt::is_pinned_host(trace))) return MS_OK;
  const int slot = trace_slot ^ 1;
  if (d_trace[slot].ensure(N_ * w_ * 8)) return fail(MS_ERR_NOMEM, "trace");
  msrt::Sdma& S = msrt::Sdma::get();
  const unsigned eng = S.h2d_engine(sdma_gpu);
  // (the buffer is free: the transposing kernel that read it ran two ms_trace_commit's ago, and every ms_trace_commit ends with a stream synchronisation)
  if (!eng || S.copy_h2d(sdma_gpu, d_trace[slot].p, trace, N_ * w_ * 8, sdma_up_sig, eng)) return MS_OK;   // engine busy / copy refused: no prefetch
  up.src = trace; up.N = N_; up.w = w_; up.slot = slot; up.pending = true;
  return MS_OK;
}

// the host trace of the ms_trace_commit in progress (N, w set) -> *dsrc, a device buffer holding it
template <class F>
int Ctx<F>::trace_to_device(const u64* trace, const u64** dsrc) {
  const size_t bytes = N * w * 8;
  if (up.src == trace && up.N == N && up.w == w) {        // prefetched (ms_trace_upload_async): normally long since complete
    const int slot = up.slot;
    up.src = nullptr;
    const int e = upload_wait("SDMA upload did not complete within the time limit");
    if (e < 0) return e;
    if (e == 0) { trace_slot = slot; *dsrc = d_trace[slot].template as<u64>(); return 0; }
  } else if (up.pending) {                                 // a prefetch of some other trace: it cannot be cancelled, only waited for
    up.src = nullptr;
    const int e = upload_wait("SDMA upload did not complete within the time limit");
    if (e < 0) return e;
  }
  const int slot = trace_slot ^ 1;
  if (d_trace[slot].ensure(bytes)) return fail(MS_ERR_NOMEM, "trace");
  bool sent = false;
  if (upload_sdma && sdma_ready() && msrt::is_pinned_host(trace)) {   // MS_UPLOAD=sdma: the trace on an SDMA engine through the HSA runtime (page-locked sources only); the host waits for it
    msrt::Sdma& S = msrt::Sdma::get();
    CK(msrt::sync(stream));                                           // (the buffer may still be read by an earlier proof's transposing kernel if that stage left on an error path)
    const unsigned eng = S.h2d_engine(sdma_gpu);
    if (eng && !S.copy_h2d(sdma_gpu, d_trace[slot].p, trace, bytes, sdma_up_sig, eng)) {
      up.pending = true;
      const int e = upload_wait("SDMA upload did not complete within the time limit");
      if (e < 0) return e;
      sent = e == 0;
    }
  }
  if (!sent) CK(msrt::h2d(d_trace[slot].p, trace, bytes, stream));
  trace_slot = slot;
  *dsrc = d_trace[slot].template as<u64>();
  return 0;
}

// ------------------------------------------------------------------ the FRI proof, device -> host
template <class F>
int Ctx<F>::fri_proof_read(u8* out) {
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

// The same copy without blocking: on an SDMA engine (completion = an HSA signal the host waits on in ms_fri_proof_wait, or the next ms_fri_query before it rewrites
// the blob), else on the context's copy stream behind an event.  The call returns at once and the next proof's stages run while the ~64 MiB travel.
template <class F>
int Ctx<F>::fri_proof_read_async(u8* out) {
  if (!blob_size || !out) return fail(MS_ERR_STATE, "no FRI proof");
  if (blob_external) return fail(MS_ERR_STATE, "the FRI proof was written to the caller's buffer (ms_fri_query_into)");
  RQ(fri_proof_wait());   // one read-back in flight per context
  if (sdma_ready() && msrt::is_pinned_host(out)) {     // ms_fri_query ended with a stream synchronisation: the blob is complete, the copy needs no dependency (the synchronisation here returns at once)
    CK(msrt::sync(stream));
    msrt::Sdma& S = msrt::Sdma::get();
    const int e = S.copy_d2h(sdma_gpu, out, d_blob.p, blob_size, sdma_sig, S.d2h_engine(sdma_gpu));
    if (!e) { sdma_pending = true; copy_pending = true; readback_dst = out; readback_bytes = blob_size; last_io_engine = 1; return MS_OK; }
    sdma_state = 2;       // refused (engine id / access / a runtime that does not know the buffers): this context stays on the runtime's copy from here on
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

template <class F>
int Ctx<F>::fri_proof_wait() {
  if (sdma_pending) {
    const int wv = msrt::Sdma::get().wait(sdma_sig, sdma_timeout_s);
    if (wv == 1) return poison("SDMA read-back did not complete within the time limit");   // sdma_pending stays set: neither the blob nor the signal is reused
    sdma_pending = false; copy_pending = false;
    if (wv < 0) {   // the engine reported a FAILED copy (negative completion signal): `out` holds garbage or nothing - once more through the HIP runtime
      sdma_state = 2; last_io_engine = 0;
      CK(msrt::d2h(readback_dst, d_blob.p, readback_bytes, stream));   // (readback_bytes, not blob_size: the next ms_fri_query has reset that by the time it waits for this copy)
      CK(msrt::sync(stream));
    }
    return MS_OK;
  }
  if (copy_pending) { CK(msrt::event_sync(ev_copy)); copy_pending = false; }
  return MS_OK;
}

#define MS_INSTANTIATE(FF) \
  template bool Ctx<FF>::sdma_ready(); \
  template int Ctx<FF>::poison(const char* what); \
  template int Ctx<FF>::upload_wait(const char* what); \
  template int Ctx<FF>::trace_upload_async(const u64* trace, size_t N_, size_t w_); \
  template int Ctx<FF>::trace_to_device(const u64* trace, const u64** dsrc); \
  template int Ctx<FF>::fri_proof_read(u8* out); \
  template int Ctx<FF>::fri_proof_read_async(u8* out); \
  template int Ctx<FF>::fri_proof_wait();
MS_INSTANTIATE(GL)
MS_INSTANTIATE(BB)
#undef MS_INSTANTIATE

}  // namespace msctx
