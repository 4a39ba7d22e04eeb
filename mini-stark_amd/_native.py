"""ctypes binding of include/ministark.h (libministark.so, HIP/gfx950).

Plumbing only: numpy arrays in, numpy arrays / bytes out.  The library is the
product; if it is missing or fails to load this module raises — there is no CPU
fallback (tests may pass an explicit `lib_path` to exercise the CPU *emulation
build of the same kernel code*, tests/emu, which the product never loads).
"""
import ctypes as C
import os
import subprocess

import numpy as np

GOLDILOCKS, BABYBEAR = 0, 1
FLAG_ZERO_DISPLAY_EMPTY = 1
FLAG_TRACE_MONT64 = 2
FLAG_LATENCY = 4   # the context proves alone on its GPU: independent chains of a stage on two streams (costs throughput with several contexts in flight)
OK, ERR_SHAPE, ERR_LEAF_NOT_FOUND, ERR_OUT_OF_RANGE, ERR_STATE, ERR_ARG, ERR_HIP, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_LIBS = {}


class MsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ministark error {code}: {msg}")
        self.code = code


def library_path():
    """The product library; MS_LIB_PATH names another build of it (same-box A/Bs of compile-time variants: tools/ab.sh)."""
    return os.environ.get("MS_LIB_PATH") or os.path.join(_HERE, "libministark.so")


def build_library(force=False):
    """hipcc cross-compile for gfx950 (works without a GPU): `make` in csrc/, one object per translation unit, side by side."""
    so = library_path()
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    jobs = str(max(1, min(8, len(os.sched_getaffinity(0)))))
    subprocess.check_call(["make", "-C", csrc, "-j", jobs], stdout=subprocess.DEVNULL)   # (no-op when up to date)
    return so


def load_library(path=None):
    path = path or library_path()   # the product loads its own in-tree HIP build only (tests pass the emulation build explicitly)
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise MsError(ERR_HIP, f"{path} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)  # the C++ host mirror (libministark_host.so) resolves ms_* against it
    L.ms_last_error.restype = C.c_char_p
    L.ms_last_error.argtypes = [C.c_void_p]
    L.ms_fri_proof_size.restype = C.c_size_t
    L.ms_fri_proof_size.argtypes = [C.c_void_p]
    L.ms_root_of_unity.restype = C.c_uint64
    L.ms_ceil_log2_k.restype = C.c_uint64
    L.ms_logarithm_of_two_k.restype = C.c_long
    _LIBS[path] = L
    return L


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_size_t)  # ms_exchange_fn
XCHG_ALL_TO_ALL, XCHG_ALL_GATHER, XCHG_ALL_REDUCE_MIN_U64, XCHG_ALL_REDUCE_SUM_U8 = 0, 1, 2, 3


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_u64p)


class Context:
    """One ms_ctx: a prover session on one GPU (include/ministark.h)."""

    def __init__(self, field=GOLDILOCKS, device=0, flags=FLAG_ZERO_DISPLAY_EMPTY, lib_path=None):
        self.L = load_library(lib_path)
        self.field = field
        h = C.c_void_p()
        rc = self.L.ms_create(C.byref(h), C.c_int(device), C.c_int(field), C.c_uint32(flags))
        if rc != 0:
            raise MsError(rc, "ms_create failed (no usable GPU / HIP runtime?)")
        self.h = h
        self.e = self.L.ms_ext_degree(self.h)
        self.N = self.w = self.Lsize = 0

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.ms_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return (self.L.ms_last_error(self.h) or b"").decode()

    def check(self, rc):
        if rc != 0:
            raise MsError(rc, self.last_error())

    # ---- host-only config math -------------------------------------------------
    def root_of_unity(self, n):
        return int(self.L.ms_root_of_unity(C.c_int(self.field), C.c_uint64(n)))

    def ceil_log2_k(self, n, base=2):
        return int(self.L.ms_ceil_log2_k(C.c_uint64(n), C.c_uint64(base)))

    def num_queries(self, security_bits, blowup, steps):
        a, b = C.c_uint64(0), C.c_uint64(0)
        rc = self.L.ms_num_queries(C.c_int(self.field), C.c_uint64(security_bits), C.c_uint64(blowup), C.c_uint64(steps), C.byref(a), C.byref(b))
        return rc, a.value, b.value

    def arith_selftest(self, op, a, b):
        """ms_arith_selftest: a (op) b element-wise ON THE DEVICE with the arithmetic class of the NTT tiles (ops: include/ministark.h)."""
        a, pa = _u64(a)
        b, pb = _u64(b)
        out = np.zeros(len(a), dtype=np.uint64)
        self.check(self.L.ms_arith_selftest(self.h, C.c_int(op), pa, pb, out.ctypes.data_as(_u64p), C.c_size_t(len(a))))
        return out

    def set_stream(self, hip_stream_ptr):
        self.check(self.L.ms_set_stream(self.h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        self.check(self.L.ms_synchronize(self.h))

    def set_shard(self, rank, world, send_ptr, recv_ptr, cap_bytes, callback):
        """ms_set_shard: one proof over `world` ranks.  `callback(op, nbytes) -> int` runs the collective on the exchange
        buffers (mini_stark_amd.dist.ShardExchange does it with torch.distributed)."""
        self._xchg_cb = EXCHANGE_FN(lambda user, op, nbytes: int(callback(int(op), int(nbytes)))) if callback is not None else EXCHANGE_FN(0)
        self.L.ms_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, EXCHANGE_FN, C.c_void_p]
        self.check(self.L.ms_set_shard(self.h, rank, world, send_ptr, recv_ptr, cap_bytes, self._xchg_cb, None))

    def rccl_unique_id(self) -> bytes:
        """ms_rccl_unique_id: the 128-byte ncclUniqueId rank 0 hands to every rank (any channel) before set_shard_rccl."""
        buf = (C.c_uint8 * 128)()
        rc = self.L.ms_rccl_unique_id(buf)
        if rc != 0:
            raise MsError(rc, "RCCL is not available (librccl.so could not be loaded)")
        return bytes(buf)

    def set_shard_rccl(self, rank, world, unique_id: bytes, cap_bytes: int):
        """ms_set_shard_rccl: one proof over `world` ranks, the collectives run by the library itself with RCCL on its stream."""
        self.L.ms_set_shard_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        self.check(self.L.ms_set_shard_rccl(self.h, rank, world, unique_id, cap_bytes))

    def shard_proof_on_root(self, on=True):
        """ms_shard_proof_on_root: a sharded proof's FRI blob is assembled on rank 0 only (the other ranks send their slices and hold no proof)."""
        self.check(self.L.ms_shard_proof_on_root(self.h, C.c_int(1 if on else 0)))

    def shard_stats(self):
        out = (C.c_uint64 * 8)()
        self.check(self.L.ms_shard_stats(self.h, out))
        return [int(v) for v in out]

    # ---- Stark::prove stages ---------------------------------------------------
    def trace_commit(self, trace, lpn):
        t = np.ascontiguousarray(trace, dtype=np.uint64)
        N, w = t.shape
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_trace_commit(self.h, t.ctypes.data_as(_u64p), C.c_size_t(N), C.c_size_t(w), C.c_size_t(lpn), root)
        if rc == 0:
            self.N, self.w = N, w
        return rc, bytes(root)

    def trace_commit_ptr(self, host_ptr, N, w, lpn):
        """ms_trace_commit on a raw host pointer (page-locked memory from pinned_alloc: the trace then travels on an SDMA engine)."""
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_trace_commit(self.h, C.c_void_p(host_ptr), C.c_size_t(N), C.c_size_t(w), C.c_size_t(lpn), root)
        if rc == 0:
            self.N, self.w = N, w
        return rc, bytes(root)

    def trace_upload_async(self, host_ptr, N, w):
        """ms_trace_upload_async: prefetch of the NEXT proof's page-locked trace while the current proof computes (a hint; MS_OK also when nothing was queued)."""
        return self.L.ms_trace_upload_async(self.h, C.c_void_p(host_ptr), C.c_size_t(N), C.c_size_t(w))

    def pinned_alloc(self, nbytes):
        self.L.ms_pinned_alloc.restype = C.c_void_p
        return self.L.ms_pinned_alloc(C.c_size_t(nbytes))

    def pinned_free(self, ptr):
        self.L.ms_pinned_free.argtypes = [C.c_void_p]
        self.L.ms_pinned_free(C.c_void_p(ptr))

    def io_runtime_path(self):
        self.L.ms_io_runtime_path.restype = C.c_char_p
        return (self.L.ms_io_runtime_path() or b"").decode()

    def trace_commit_device(self, dev_ptr, N, w, lpn):
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_trace_commit_device(self.h, C.c_void_p(dev_ptr), C.c_size_t(N), C.c_size_t(w), C.c_size_t(lpn), root)
        if rc == 0:
            self.N, self.w = N, w
        return rc, bytes(root)

    def interpolate(self):
        return self.L.ms_interpolate(self.h)

    def polys_lincomb(self, scalars, idx):
        s, sp = _u64(scalars)
        i = np.ascontiguousarray(idx, dtype=np.int32)
        return self.L.ms_polys_lincomb(self.h, sp, i.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(len(i)))

    def polys_append(self, coeffs):
        c, cp = _u64(coeffs)
        return self.L.ms_polys_append(self.h, cp, C.c_size_t(c.size))

    def polys_count(self):
        return self.L.ms_polys_count(self.h)

    def poly_read(self, i):
        out = np.zeros(self.N, dtype=np.uint64)
        self.check(self.L.ms_poly_read(self.h, C.c_int(i), out.ctypes.data_as(_u64p)))
        return out

    def lde_commit(self, blowup, shift, lpn):
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_lde_commit(self.h, C.c_size_t(blowup), C.c_uint64(shift), C.c_size_t(lpn), root)
        if rc == 0:
            self.Lsize = self.N * blowup
        return rc, bytes(root)

    def bench_lde(self, blowup, shift):
        return self.L.ms_bench_lde(self.h, C.c_size_t(blowup), C.c_uint64(shift))

    def lde_read(self):
        out = np.zeros((self.Lsize, self.polys_count()), dtype=np.uint64)
        self.check(self.L.ms_lde_read(self.h, out.ctypes.data_as(_u64p)))
        return out

    def mix(self, r):
        return self.L.ms_mix(self.h, C.c_uint64(r))

    def mix_cubic(self, r, spec, scalars):
        """ms_mix_cubic (BUILD-DEFINED degree-3 composition with the true quotient; include/ministark.h): spec = [(j, a, b, c, d), ...]."""
        sp = np.ascontiguousarray(spec, dtype=np.int32).reshape(-1, 5)
        sc, scp = _u64(scalars)
        return self.L.ms_mix_cubic(self.h, C.c_uint64(r), sp.ctypes.data_as(C.POINTER(C.c_int)), scp, C.c_int(len(sp)))

    def validity_read(self):
        self.L.ms_validity_len.restype = C.c_size_t
        out = np.zeros(int(self.L.ms_validity_len(self.h)) or self.N, dtype=np.uint64)
        self.check(self.L.ms_validity_read(self.h, out.ctypes.data_as(_u64p)))
        return out

    def eval_ext(self, z):
        z, zp = _u64(z)
        q = z.size // self.e
        out = np.zeros((q, self.polys_count() + 1, self.e), dtype=np.uint64)
        rc = self.L.ms_eval_ext(self.h, zp, C.c_int(q), out.ctypes.data_as(_u64p))
        return rc, out

    # ---- Fri::prove stages -------------------------------------------------------
    def fri_begin(self, blowup, rounds):
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_fri_begin(self.h, C.c_size_t(blowup), C.c_size_t(rounds), root)
        return rc, bytes(root)

    def fri_deep(self, z):
        z, zp = _u64(z)
        B = np.zeros(2 * self.e, dtype=np.uint64)
        rc = self.L.ms_fri_deep(self.h, zp, B.ctypes.data_as(_u64p))
        return rc, B

    def fri_fold_commit(self, alpha):
        a, ap = _u64(alpha)
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_fri_fold_commit(self.h, ap, root)
        return rc, bytes(root)

    def fri_round_info(self, r):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.check(self.L.ms_fri_round_info(self.h, C.c_int(r), C.byref(a), C.byref(b)))
        return a.value, b.value

    def fri_round_poly(self, r):
        n, _ = self.fri_round_info(r)
        out = np.zeros((n, self.e), dtype=np.uint64)
        if n:
            self.check(self.L.ms_fri_round_poly_read(self.h, C.c_int(r), out.ctypes.data_as(_u64p)))
        return out

    def fri_round_codeword(self, r):
        _, D = self.fri_round_info(r)
        out = np.zeros((D, self.e), dtype=np.uint64)
        self.check(self.L.ms_fri_round_codeword_read(self.h, C.c_int(r), out.ctypes.data_as(_u64p)))
        return out

    def fri_query(self, betas, read=True):
        b, bp = _u64(betas)
        rc = self.L.ms_fri_query(self.h, bp, C.c_int(b.size))
        if rc != 0 or not read:
            return rc, None
        return 0, self.fri_proof_read()

    def fri_proof_size(self):
        return int(self.L.ms_fri_proof_size(self.h))

    def fri_proof_read(self):
        n = self.fri_proof_size()
        if n == 0 and self.L.ms_shard_proof_is_elsewhere(self.h) == 1:
            return b""       # ms_shard_proof_on_root: this rank sent its slices to rank 0 and holds no proof
        buf = np.zeros(max(1, n), dtype=np.uint8)
        self.check(self.L.ms_fri_proof_read(self.h, buf.ctypes.data_as(_u8p)))
        return buf[:n].tobytes()

    # ---- standalone ----------------------------------------------------------------
    def merkle_commit(self, leafs, ext=1, lpn=2, ic=2):
        a, p = _u64(leafs)
        leaf_num = a.size // ext
        cap = max(1, 2 * leaf_num)
        nodes = np.zeros((cap, 32), dtype=np.uint8)
        nn = C.c_size_t(0)
        root = (C.c_uint8 * 32)()
        rc = self.L.ms_merkle_commit(self.h, p, C.c_size_t(leaf_num), C.c_int(ext), C.c_size_t(lpn), C.c_size_t(ic),
                                     nodes.ctypes.data_as(_u8p), C.c_size_t(cap), C.byref(nn), root)
        if rc != 0:
            return rc, None, None
        return 0, nodes[: nn.value].copy(), bytes(root)

    def merkle_prove(self, leafs, leaf, ext=1, lpn=2):
        a, p = _u64(leafs)
        l, lp = _u64(leaf)
        leaf_num = a.size // ext
        cap = 64 + lpn * ext * 8 + 64 * 64
        buf = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.L.ms_merkle_prove(self.h, p, C.c_size_t(leaf_num), C.c_int(ext), C.c_size_t(lpn), lp, buf.ctypes.data_as(_u8p), C.c_size_t(cap), C.byref(n))
        return (rc, None) if rc != 0 else (0, buf[: n.value].tobytes())

    def ntt(self, data, inverse=False):
        a = np.array(data, dtype=np.uint64)
        shape = a.shape
        a = np.ascontiguousarray(a.reshape(-1, shape[-1]))
        rc = self.L.ms_ntt(self.h, a.ctypes.data_as(_u64p), C.c_size_t(a.shape[1]), C.c_size_t(a.shape[0]), C.c_int(1 if inverse else 0))
        return rc, a.reshape(shape)

    def coset_lde(self, coeffs, shift, L):
        c = np.ascontiguousarray(np.atleast_2d(np.asarray(coeffs, dtype=np.uint64)))
        out = np.zeros((c.shape[0], L), dtype=np.uint64)
        rc = self.L.ms_coset_lde(self.h, c.ctypes.data_as(_u64p), C.c_size_t(c.shape[1]), C.c_size_t(c.shape[0]), C.c_uint64(shift),
                                 out.ctypes.data_as(_u64p), C.c_size_t(L))
        return rc, out
