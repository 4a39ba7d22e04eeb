"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under mini-stark_amd/ does.  The library is the CPU
restatement of the reference's LDE + FRI + Merkle path
(oracle/ministark_oracle.cpp, which cites the reference file:line per function).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GOLDILOCKS, BABYBEAR = 0, 1
OK, ERR_SHAPE, ERR_LEAF_NOT_FOUND, ERR_OUT_OF_RANGE, ERR_STATE, ERR_ARG = 0, -1, -2, -3, -4, -5

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ministark_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.or_create.restype = C.c_void_p
        L.or_create_ext.restype = C.c_void_p
        L.or_modulus.restype = C.c_uint64
        L.or_root_of_unity.restype = C.c_uint64
        L.or_mul.restype = C.c_uint64
        L.or_inv.restype = C.c_uint64
        L.or_pow.restype = C.c_uint64
        L.or_ceil_log2_k.restype = C.c_uint64
        L.or_logarithm_of_two_k.restype = C.c_long
        L.or_fri_proof_size.restype = C.c_size_t
        L.or_display.restype = C.c_size_t
        L.or_merkle_prove.restype = C.c_long
        for name in ("or_modulus", "or_root_of_unity", "or_mul", "or_inv", "or_pow"):
            getattr(L, name).argtypes = None
        _LIB = L
    return _LIB


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def set_threads(n):
    """OpenMP threads for the oracle's independent loops (default 1, like the reference)."""
    lib().or_set_threads(C.c_int(n))


def max_threads():
    return int(lib().or_max_threads())


def modulus(field):
    return int(lib().or_modulus(C.c_int(field)))


def root_of_unity(field, n):
    return int(lib().or_root_of_unity(C.c_int(field), C.c_uint64(n)))


def sha256(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().or_sha256(msg, C.c_size_t(len(msg)), out)
    return bytes(out)


def display(ext, limbs, zero_as_empty=1) -> bytes:
    a, p = _u64(limbs)
    count = a.size // ext
    n = lib().or_display(C.c_int(ext), p, C.c_size_t(count), C.c_int(zero_as_empty), None, C.c_size_t(0))
    buf = C.create_string_buffer(n + 1)
    lib().or_display(C.c_int(ext), p, C.c_size_t(count), C.c_int(zero_as_empty), buf, C.c_size_t(n))
    return buf.raw[:n]


def intt(field, a):
    a = np.array(a, dtype=np.uint64)
    rc = lib().or_intt(C.c_int(field), a.ctypes.data_as(u64p), C.c_size_t(a.size))
    assert rc == 0, rc
    return a


def ntt(field, a):
    a = np.array(a, dtype=np.uint64)
    rc = lib().or_ntt(C.c_int(field), a.ctypes.data_as(u64p), C.c_size_t(a.size))
    assert rc == 0, rc
    return a


def coset_lde(field, coeffs, shift, L):
    c, cp = _u64(coeffs)
    out = np.zeros(L, dtype=np.uint64)
    rc = lib().or_coset_lde(C.c_int(field), cp, C.c_size_t(c.size), C.c_uint64(shift), out.ctypes.data_as(u64p), C.c_size_t(L))
    assert rc == 0, rc
    return out


def ext_mul(field, ext, a, b):
    a, ap = _u64(a)
    b, bp = _u64(b)
    out = np.zeros(ext, dtype=np.uint64)
    assert lib().or_ext_mul(C.c_int(field), C.c_int(ext), ap, bp, out.ctypes.data_as(u64p)) == 0
    return out


def ext_inv(field, ext, a):
    a, ap = _u64(a)
    out = np.zeros(ext, dtype=np.uint64)
    assert lib().or_ext_inv(C.c_int(field), C.c_int(ext), ap, out.ctypes.data_as(u64p)) == 0
    return out


def merkle_build(leafs, ext=1, lpn=2, ic=2, zero_as_empty=1):
    """Returns (rc, nodes[n,32] uint8, root bytes)."""
    a, p = _u64(leafs)
    leaf_num = a.size // ext
    cap = max(1, 2 * leaf_num)
    nodes = np.zeros((cap, 32), dtype=np.uint8)
    nn = C.c_size_t(0)
    root = (C.c_uint8 * 32)()
    rc = lib().or_merkle_build(C.c_int(ext), p, C.c_size_t(leaf_num), C.c_size_t(lpn), C.c_size_t(ic), C.c_int(zero_as_empty),
                               nodes.ctypes.data_as(u8p), C.c_size_t(cap), C.byref(nn), root)
    if rc != 0:
        return rc, None, None
    return 0, nodes[: nn.value].copy(), bytes(root)


def merkle_parent_idx(leaf_num, lpn, ic, index):
    out = C.c_size_t(0)
    rc = lib().or_merkle_parent_idx(C.c_size_t(leaf_num), C.c_size_t(lpn), C.c_size_t(ic), C.c_size_t(index), C.byref(out))
    return rc, out.value


def merkle_prove(leafs, leaf, ext=1, lpn=2, ic=2, zero_as_empty=1):
    a, p = _u64(leafs)
    l, lp = _u64(leaf)
    leaf_num = a.size // ext
    cap = 64 + lpn * ext * 8 + 64 * ic * 32
    buf = (C.c_uint8 * cap)()
    n = lib().or_merkle_prove(C.c_int(ext), p, C.c_size_t(leaf_num), C.c_size_t(lpn), C.c_size_t(ic), C.c_int(zero_as_empty), lp, buf, C.c_size_t(cap))
    if n < 0:
        return n, None
    return 0, bytes(buf[:n])


def merkle_check_proof(root, path, ext=1, lpn=2, ic=2, zero_as_empty=1):
    return bool(lib().or_merkle_check_proof(root, path, C.c_size_t(len(path)), C.c_int(ext), C.c_size_t(lpn), C.c_size_t(ic), C.c_int(zero_as_empty)))


def num_queries(field, security_bits, blowup, steps):
    a, b = C.c_uint64(0), C.c_uint64(0)
    rc = lib().or_num_queries(C.c_int(field), C.c_uint64(security_bits), C.c_uint64(blowup), C.c_uint64(steps), C.byref(a), C.byref(b))
    return rc, a.value, b.value


def fri_verify(field, ext, rounds, betas, zs, Bs, alphas, roots, proof, zero_as_empty=1):
    betas, bp = _u64(betas)
    zs, zp = _u64(zs)
    Bs, Bp = _u64(Bs)
    alphas, ap = _u64(alphas)
    return int(lib().or_fri_verify(C.c_int(field), C.c_int(ext), C.c_int(zero_as_empty), C.c_size_t(rounds), C.c_size_t(betas.size), bp, zp, Bp, ap,
                                   roots, proof, C.c_size_t(len(proof))))


class Session:
    """Stage-by-stage mirror of Stark::prove (src/starks.rs:59-169) on the CPU."""

    def __init__(self, field, zero_as_empty=1, ext=None):
        self.L = lib()
        self.field = field
        if ext is None:
            self.h = C.c_void_p(self.L.or_create(C.c_int(field), C.c_int(zero_as_empty)))
        else:
            self.h = C.c_void_p(self.L.or_create_ext(C.c_int(field), C.c_int(ext), C.c_int(zero_as_empty)))
        assert self.h.value
        self.e = self.L.or_ext_degree(self.h)
        self.N = self.w = self.Lsize = 0

    def close(self):
        if self.h is not None and self.h.value:
            self.L.or_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_commit(self, trace, lpn):
        t = np.ascontiguousarray(trace, dtype=np.uint64)
        N, w = t.shape
        root = (C.c_uint8 * 32)()
        rc = self.L.or_trace_commit(self.h, t.ctypes.data_as(u64p), C.c_size_t(N), C.c_size_t(w), C.c_size_t(lpn), root)
        if rc == 0:
            self.N, self.w = N, w
        return rc, bytes(root)

    def interpolate(self):
        return self.L.or_interpolate(self.h)

    def polys_lincomb(self, scalars, idx):
        s, sp = _u64(scalars)
        i = np.ascontiguousarray(idx, dtype=np.int32)
        return self.L.or_polys_lincomb(self.h, sp, i.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(len(i)))

    def polys_count(self):
        return self.L.or_polys_count(self.h)

    def poly_read(self, i):
        out = np.zeros(self.N, dtype=np.uint64)
        assert self.L.or_poly_read(self.h, C.c_int(i), out.ctypes.data_as(u64p)) == 0
        return out

    def lde_commit(self, blowup, shift, lpn):
        root = (C.c_uint8 * 32)()
        rc = self.L.or_lde_commit(self.h, C.c_size_t(blowup), C.c_uint64(shift), C.c_size_t(lpn), root)
        if rc == 0:
            self.Lsize = self.N * blowup
        return rc, bytes(root)

    def lde_read(self):
        c = self.polys_count()
        out = np.zeros((self.Lsize, c), dtype=np.uint64)
        assert self.L.or_lde_read(self.h, out.ctypes.data_as(u64p)) == 0
        return out

    def mix(self, r):
        return self.L.or_mix(self.h, C.c_uint64(r))

    def validity_read(self):
        out = np.zeros(self.N, dtype=np.uint64)
        assert self.L.or_validity_read(self.h, out.ctypes.data_as(u64p)) == 0
        return out

    def eval_ext(self, z):
        z, zp = _u64(z)
        q = z.size // self.e
        c = self.polys_count()
        out = np.zeros((q, c + 1, self.e), dtype=np.uint64)
        rc = self.L.or_eval_ext(self.h, zp, C.c_int(q), out.ctypes.data_as(u64p))
        return rc, out

    def verify_ood(self, r, z, evals):
        """src/starks.rs:204-225 against this session's polynomials; evals: [q][c+1][E]."""
        z, zp = _u64(z)
        ev, ep = _u64(evals)
        return int(self.L.or_verify_ood(self.h, C.c_uint64(r), zp, C.c_int(z.size // self.e), ep))

    def fri_begin(self, blowup, rounds):
        root = (C.c_uint8 * 32)()
        rc = self.L.or_fri_begin(self.h, C.c_size_t(blowup), C.c_size_t(rounds), root)
        return rc, bytes(root)

    def fri_deep(self, z):
        z, zp = _u64(z)
        B = np.zeros(2 * self.e, dtype=np.uint64)
        rc = self.L.or_fri_deep(self.h, zp, B.ctypes.data_as(u64p))
        return rc, B

    def fri_fold_commit(self, alpha):
        a, ap = _u64(alpha)
        root = (C.c_uint8 * 32)()
        rc = self.L.or_fri_fold_commit(self.h, ap, root)
        return rc, bytes(root)

    def fri_round_info(self, r):
        a, b = C.c_uint64(0), C.c_uint64(0)
        rc = self.L.or_fri_round_info(self.h, C.c_int(r), C.byref(a), C.byref(b))
        assert rc == 0
        return a.value, b.value

    def fri_round_poly(self, r):
        n, _ = self.fri_round_info(r)
        out = np.zeros((n, self.e), dtype=np.uint64)
        if n:
            assert self.L.or_fri_round_poly_read(self.h, C.c_int(r), out.ctypes.data_as(u64p)) == 0
        return out

    def fri_round_codeword(self, r):
        _, D = self.fri_round_info(r)
        out = np.zeros((D, self.e), dtype=np.uint64)
        assert self.L.or_fri_round_codeword_read(self.h, C.c_int(r), out.ctypes.data_as(u64p)) == 0
        return out

    def fri_query(self, betas):
        b, bp = _u64(betas)
        rc = self.L.or_fri_query(self.h, bp, C.c_int(b.size))
        if rc != 0:
            return rc, None
        n = self.L.or_fri_proof_size(self.h)
        buf = (C.c_uint8 * max(1, n))()
        self.L.or_fri_proof_read(self.h, buf)
        return 0, bytes(buf[:n])
