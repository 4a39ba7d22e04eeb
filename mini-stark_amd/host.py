"""ctypes binding of the C++ host mirror (mini-stark_amd/host/stark_host.cpp: ministark::StarkConfig /
Stark::prove / Transcript above the C ABI).  libministark_host.so contains no arithmetic: it calls the
ms_* stage functions of whichever libministark build the Context was created from."""
import ctypes as C
import os
import subprocess

import numpy as np

from ._native import Context, MsError
from .stark import FriProof, StarkProof

_HERE = os.path.dirname(os.path.abspath(__file__))
_HOST = None


def host_library_path():
    return os.path.join(_HERE, "libministark_host.so")


def build_host_library(force=False):
    so, src = host_library_path(), os.path.join(_HERE, "host", "stark_host.cpp")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    deps = [src, os.path.join(inc, "ministark.h"), os.path.join(inc, "ministark_host.h"), os.path.join(_HERE, "csrc", "field.hpp"), os.path.join(_HERE, "csrc", "rt.hpp")]
    if not force and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return so


def _host():
    global _HOST
    if _HOST is None:
        path = host_library_path()
        if not os.path.exists(path):
            raise MsError(-6, f"{path} not found: run __graft_entry__.build()")
        from . import _native
        if not _native._LIBS:      # the mirror resolves ms_* against the libministark build loaded before it: the product's own, unless a test loaded the emulation build
            _native.load_library()
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.msh_stark_new.restype = C.c_void_p
        for n in ("msh_proof_arthur", "msh_proof_evals", "msh_proof_fri_roots", "msh_proof_fri_blob", "msh_proof_challenges", "msh_proof_num_polys", "msh_proof_serialize",
                  "msh_prev_proof_arthur", "msh_prev_proof_fri_roots", "msh_prev_proof_fri_blob"):
            getattr(L, n).restype = C.c_size_t
        L.msh_proof_blob_checksum.restype = C.c_uint64
        L.msh_proof_blob_sample.restype = C.c_uint64
        _HOST = L
    return _HOST


class HostStark:
    """StarkConfig::new + Stark::new + Stark::prove + Stark::verify (src/starks.rs:268-310, 40-57, 59-169, 171-235) in C++."""

    def __init__(self, ctx: Context, security_bits: int, blowup_factor: int, steps: int, trace_columns: int):
        self.H, self.ctx = _host(), ctx
        err = C.c_int(0)
        self.h = C.c_void_p(self.H.msh_stark_new(ctx.h, C.c_int(ctx.field), C.c_uint64(security_bits), C.c_uint64(blowup_factor), C.c_uint64(steps),
                                                 C.c_uint64(trace_columns), C.byref(err)))
        if not self.h.value:
            raise MsError(err.value, "STARK Config: security bits has to be at least 20")
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.H.msh_stark_config(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.rounds, self.constrain_queries, self.fri_queries = a.value, b.value, c.value

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.H.msh_stark_free(self.h)
                self.h = None
        except Exception:
            pass

    def prove_raw(self, trace, trace_device_ptr=None, read_fri_proof=True):
        """Runs the proof; returns the status code only (the bench loop).  read_fri_proof: False (the FRI proof stays in HBM), True (blocking
        read-back), "async" (read-back on the copy stream; wait_proof() completes it), "into" (the query-phase kernels write the blob straight
        into the slot's page-locked buffer: ms_fri_query_into)."""
        if not hasattr(trace, "_lin"):
            k = np.array([len(i) for _, i in trace.transitions], dtype=np.int32)
            sc = np.array([v for s, _ in trace.transitions for v in s], dtype=np.uint64)
            ix = np.array([v for _, i in trace.transitions for v in i], dtype=np.int32)
            trace._lin = (k, sc, ix)
        k, sc, ix = trace._lin
        host_ptr = None if trace_device_ptr is not None else trace.data.ctypes.data_as(C.POINTER(C.c_uint64))
        return self.H.msh_stark_prove(self.h, host_ptr, C.c_void_p(trace_device_ptr), C.c_size_t(trace.length), C.c_size_t(trace.width), C.c_int(len(k)),
                                      k.ctypes.data_as(C.POINTER(C.c_int)), sc.ctypes.data_as(C.POINTER(C.c_uint64)), ix.ctypes.data_as(C.POINTER(C.c_int)),
                                      C.c_int({"async": 2, "into": 3}.get(read_fri_proof, 1 if read_fri_proof else 0)))

    def next_trace(self, host_ptr, N, w):
        """msh_stark_next_trace: the page-locked trace of the proof after the next prove_raw - that call prefetches it (ms_trace_upload_async)."""
        self.H.msh_stark_next_trace(self.h, C.c_void_p(host_ptr), C.c_size_t(N), C.c_size_t(w))

    def blob_checksum(self, which=0):
        """FNV-1a over the FRI blob of the last (0) / previous (1) proof, read in place from its page-locked slot (msh_proof_blob_checksum)."""
        return int(self.H.msh_proof_blob_checksum(self.h, C.c_int(which)))

    def blob_sample(self, which=0, stride_bytes=4096):
        """FNV-1a over one word per `stride_bytes` of the blob (+ the last word): touches every page without reading all of it."""
        return int(self.H.msh_proof_blob_sample(self.h, C.c_int(which), C.c_size_t(stride_bytes)))

    def prev_fri_blob(self) -> bytes:
        """The FRI blob of the proof BEFORE the last one: the mirror keeps two proof slots, so it stays whole while the next proof is computed."""
        return self._bytes(self.H.msh_prev_proof_fri_blob)

    def wait_proof(self):
        """Completes an asynchronous read-back (read_fri_proof="async": the FRI proof travels into the mirror's page-locked buffer while
        the next proof is already being computed)."""
        return self.H.msh_proof_wait(self.h)

    def _bytes(self, fn):
        n = fn(self.h, None, C.c_size_t(0))
        buf = (C.c_uint8 * max(1, n))()
        fn(self.h, buf, C.c_size_t(n))
        return bytes(buf[:n])

    def last_proof(self, read_fri_proof=True) -> StarkProof:
        e = self.ctx.e
        tc, lc = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        self.H.msh_proof_commits(self.h, tc, lc)
        c = int(self.H.msh_proof_num_polys(self.h))
        n = self.H.msh_proof_evals(self.h, None, C.c_size_t(0))
        ev = np.zeros(max(1, n), dtype=np.uint64)
        self.H.msh_proof_evals(self.h, ev.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(n))
        ev = ev[:n].reshape(-1, c + 1, e)
        roots = self._bytes(self.H.msh_proof_fri_roots)
        n = self.H.msh_proof_challenges(self.h, None, C.c_size_t(0))
        ch = np.zeros(max(1, n), dtype=np.uint64)
        self.H.msh_proof_challenges(self.h, ch.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(n))
        self.last_challenges = ch[:n]
        blob = self._bytes(self.H.msh_proof_fri_blob) if read_fri_proof else b""
        return StarkProof(self._bytes(self.H.msh_proof_arthur), bytes(tc), bytes(lc), ev[:, :c, :], ev[:, c, :],
                          FriProof(blob, device_resident=not read_fri_proof), [roots[i:i + 32] for i in range(0, len(roots), 32)])

    def derive_constrains(self, trace):
        """TraceTable::derive_constrains (src/air.rs:127-144) for the verifier's copy of the AIR: the c constraint polynomials
        in coefficient form, [c][N] canonical u64 (computed on the GPU: INTT of the trace columns + the transition closures)."""
        ctx = self.ctx
        rc, _ = ctx.trace_commit(trace.data, trace.constrain_number())
        ctx.check(rc)
        ctx.check(ctx.interpolate())
        for sc, idx in trace.transitions:
            ctx.check(ctx.polys_lincomb(sc, idx))
        return np.stack([ctx.poly_read(i) for i in range(ctx.polys_count())])

    def verify(self, constrains, proof: StarkProof, zero_display_empty=True) -> bool:
        """Stark::verify (src/starks.rs:171-235, with Fri::verify src/fri.rs:191-290 and MerkleRoot::check_proof
        src/merkle.rs:312-338) on the CPU, as in the reference.  Returns True/False; the reason of a rejection is in
        `self.last_verify_error`.  Raises MsError on malformed input.
        A PARITY MIRROR of the reference's verifier, NOT a sound verifier: like the reference it takes round 0's root from the proof
        (it never enters the transcript), does not tie y3 of a window to the next window, does not check the last round polynomial
        and only degree-bounds the shipped quotients.  True means "the reference would accept" (INTEGRATION.md section 8)."""
        cs = np.ascontiguousarray(constrains, dtype=np.uint64)
        c, N = cs.shape
        ev = np.ascontiguousarray(np.concatenate([np.asarray(proof.constrain_queries, dtype=np.uint64).reshape(-1, c, self.ctx.e),
                                                  np.asarray(proof.validity_queries, dtype=np.uint64).reshape(-1, 1, self.ctx.e)], axis=1))
        roots = b"".join(proof.fri_roots)
        blob = proof.fri_proof.blob
        why = C.create_string_buffer(512)
        u8p = C.POINTER(C.c_uint8)

        def b(x):
            return C.cast(C.create_string_buffer(bytes(x), max(1, len(x))), u8p)
        self.H.msh_stark_verify.restype = C.c_int
        rc = self.H.msh_stark_verify(self.h, cs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(c), C.c_size_t(N), b(proof.arthur), C.c_size_t(len(proof.arthur)),
                                     b(proof.trace_commit), b(proof.constrain_trace_commit), ev.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(ev.size),
                                     b(roots), C.c_size_t(len(proof.fri_roots)), b(blob), C.c_size_t(len(blob)), C.c_int(1 if zero_display_empty else 0), why, C.c_size_t(512))
        self.last_verify_error = why.value.decode()
        if rc < 0:
            raise MsError(rc, self.last_verify_error)
        return rc == 1

    def proof_bytes(self) -> bytes:
        """msh_proof_serialize: the last proof in the MSSP wire format (include/ministark_host.h), written by the C++ mirror."""
        n = self.H.msh_proof_serialize(self.h, None, C.c_size_t(0))
        if n == 0:
            raise MsError(-4, "no proof to serialise (none proved yet, or its FRI proof was left in HBM)")
        buf = (C.c_uint8 * n)()
        assert self.H.msh_proof_serialize(self.h, buf, C.c_size_t(n)) == n
        return bytes(buf)

    def verify_bytes(self, constrains, data: bytes, zero_display_empty=True) -> bool:
        """msh_stark_verify_mssp: Stark::verify straight from the wire format."""
        cs = np.ascontiguousarray(constrains, dtype=np.uint64)
        c, N = cs.shape
        why = C.create_string_buffer(512)
        self.H.msh_stark_verify_mssp.restype = C.c_int
        rc = self.H.msh_stark_verify_mssp(self.h, cs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(c), C.c_size_t(N), data, C.c_size_t(len(data)),
                                          C.c_int(1 if zero_display_empty else 0), why, C.c_size_t(512))
        self.last_verify_error = why.value.decode()
        if rc < 0:
            raise MsError(rc, self.last_verify_error)
        return rc == 1

    def prove(self, trace, trace_device_ptr=None, read_fri_proof=True) -> StarkProof:
        self.ctx.check(self.prove_raw(trace, trace_device_ptr, read_fri_proof))
        return self.last_proof(read_fri_proof)


def fibonacci_rows_native(p, length, steps, secret_b=2, pad_seed=0x5EED):
    """msh_fibonacci_rows: the benchmark's synthetic trace in C (same values as mini_stark_amd.synthetic.fibonacci_rows; the Python
    loop takes ~1 us per row, 16 s at 2^24 rows)."""
    out = np.empty((length, 3), dtype=np.uint64)
    rc = _host().msh_fibonacci_rows(C.c_uint64(p), C.c_size_t(length), C.c_size_t(steps), C.c_uint64(secret_b), C.c_uint64(pad_seed), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise MsError(rc, "msh_fibonacci_rows")
    return out


def cubic_rows_native(p, length, w, seed=9):
    """msh_cubic_rows: the synthetic trace of the build-defined degree-3 wide AIR (same values as tests/parity_cases.py:cubic_trace) and its w scalars."""
    out = np.empty((length, w), dtype=np.uint64)
    sc = np.empty(w, dtype=np.uint64)
    rc = _host().msh_cubic_rows(C.c_uint64(p), C.c_size_t(length), C.c_size_t(w), C.c_uint64(seed), out.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise MsError(rc, "msh_cubic_rows")
    return out, sc
