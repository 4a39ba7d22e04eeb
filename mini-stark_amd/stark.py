"""Host-side mirror of the reference's prover API for the accelerated path.

Same names and argument meaning as the reference (`StarkConfig::new`,
`Stark::prove`, `TraceTable`, `StarkProof`, `FriProof` — src/starks.rs:21-57,
238-333, src/air.rs:63-161, src/fri.rs:17-30), sequencing the C-ABI stage
functions of include/ministark.h exactly where `Stark::prove`
(src/starks.rs:59-169) and `Fri::commit_phase/query_phase` (src/fri.rs:64-189)
call arkworks and `MerkleTree::new`.  All arithmetic runs on the GPU inside
libministark.so; this file only moves challenges and commitments between the
transcript and the library.

Transcript: the reference uses nimue (`IOPattern`/`Merlin` over
`DigestBridge<Sha256>`, src/fiatshamir.rs) whose source is not available here,
so `Transcript` below is a BUILD-DEFINED SHA-256 hash chain that follows the
same message ORDER (src/fiatshamir.rs:48-64,100-116) but not nimue's bytes.
A Rust caller keeps nimue and passes the challenges in.
"""
import hashlib
import struct
from dataclasses import dataclass, field as _field
from typing import List

import numpy as np

from ._native import Context, MsError, GOLDILOCKS, BABYBEAR, ERR_SHAPE

_MODULUS = {GOLDILOCKS: 2**64 - 2**32 + 1, BABYBEAR: 2013265921}


class Transcript:
    """Build-defined Fiat–Shamir sponge (NOT nimue): state' = SHA256(state || tag || data)."""

    def __init__(self, domsep: str):
        self.state = hashlib.sha256(b"mini-stark_amd/transcript/v0" + domsep.encode()).digest()
        self.prover_bytes = bytearray()  # what nimue calls the transcript ("arthur", starks.rs:160)
        self.ops = []                    # ("absorb", nbytes) / ("squeeze", nbytes): the IOPattern the reference declares (fiatshamir.rs:48-64,100-116)

    def add_bytes(self, data: bytes):
        self.ops.append(("absorb", len(data)))
        self.prover_bytes += data
        self.state = hashlib.sha256(self.state + b"A" + data).digest()

    def add_scalars(self, limbs):
        self.add_bytes(b"".join(struct.pack("<Q", int(v)) for v in limbs))

    def challenge_bytes(self, n: int, _scalars=None) -> bytes:
        self.ops.append(("squeeze_scalars", _scalars) if _scalars is not None else ("squeeze_bytes", n))
        out = b""
        ctr = 0
        while len(out) < n:
            out += hashlib.sha256(self.state + b"C" + struct.pack("<I", ctr)).digest()
            ctr += 1
        self.state = hashlib.sha256(self.state + b"R").digest()
        return out[:n]

    def challenge_scalars(self, count: int, p: int):
        raw = self.challenge_bytes(16 * count, _scalars=count)
        return [int.from_bytes(raw[16 * i:16 * i + 16], "little") % p for i in range(count)]


class TraceTable:
    """src/air.rs:63-161.  `data` is the N x w row-major matrix incl. the padding rows."""

    def __init__(self, ctx: Context, steps: int, registers: int):
        self.ctx, self.steps, self.width = ctx, steps, registers
        self.length = 1
        while self.length < steps + 1:  # Radix2EvaluationDomain::new(steps + 1), air.rs:74
            self.length <<= 1
        self.omega = ctx.root_of_unity(self.length)  # air.rs:75
        self.data = np.zeros((self.length, registers), dtype=np.uint64)
        self.transitions = []  # (scalars, idx) linear combinations of trace polys

    def add_row(self, index, row):  # air.rs:106-112
        assert len(row) == self.width and index < self.steps
        self.data[index] = row

    def add_transition_lincomb(self, scalars, idx):  # closures of tests/e2e_goldilocks.rs:48-59
        self.transitions.append((list(scalars), list(idx)))

    def constrain_number(self):  # air.rs:123-125
        return self.width + len(self.transitions)


@dataclass
class FriProof:  # src/fri.rs:17-22, serialised ("MSFP", include/ministark.h)
    blob: bytes = b""
    device_resident: bool = False


@dataclass
class StarkProof:  # src/starks.rs:21-28
    arthur: bytes
    trace_commit: bytes
    constrain_trace_commit: bytes
    constrain_queries: np.ndarray  # [q][c][E]
    validity_queries: np.ndarray   # [q][E]
    fri_proof: FriProof
    fri_roots: List[bytes] = _field(default_factory=list)

    # ---- wire format "MSSP" (build-defined: the reference derives no Serialize for StarkProof/FriProof/MerklePath)
    #   u32 magic 'MSSP' | u32 version 1 | u32 E | u32 c | u32 q | u32 rounds | u64 len(arthur) | u64 len(fri blob)
    #   trace_commit[32] | constrain_trace_commit[32] | constrain_queries q*c*E u64 | validity_queries q*E u64
    #   fri_roots rounds*32 (round 0 first; not part of the reference's StarkProof: round 0's root never reaches the transcript)
    #   arthur bytes | FriProof in the MSFP layout of include/ministark.h            (all little-endian)
    def to_bytes(self) -> bytes:
        cq = np.ascontiguousarray(self.constrain_queries, dtype="<u8")
        vq = np.ascontiguousarray(self.validity_queries, dtype="<u8")
        q, c, e = cq.shape
        if self.fri_proof.device_resident:
            raise ValueError("the FRI proof was left in HBM (read_fri_proof=False): nothing to serialise")
        head = struct.pack("<4sIIIIIQQ", b"MSSP", 1, e, c, q, len(self.fri_roots), len(self.arthur), len(self.fri_proof.blob))
        return b"".join([head, self.trace_commit, self.constrain_trace_commit, cq.tobytes(), vq.tobytes(), b"".join(self.fri_roots), self.arthur, self.fri_proof.blob])

    @staticmethod
    def from_bytes(data: bytes) -> "StarkProof":
        hs = struct.calcsize("<4sIIIIIQQ")
        magic, ver, e, c, q, rounds, la, lb = struct.unpack_from("<4sIIIIIQQ", data, 0)
        if magic != b"MSSP" or ver != 1:
            raise ValueError("not an MSSP v1 proof")
        need = hs + 64 + 8 * q * (c + 1) * e + 32 * rounds + la + lb
        if len(data) != need:
            raise ValueError(f"MSSP proof has {len(data)} bytes, header says {need}")
        pos = hs
        tc, lc = data[pos:pos + 32], data[pos + 32:pos + 64]; pos += 64
        cq = np.frombuffer(data, dtype="<u8", count=q * c * e, offset=pos).reshape(q, c, e).copy(); pos += 8 * q * c * e
        vq = np.frombuffer(data, dtype="<u8", count=q * e, offset=pos).reshape(q, e).copy(); pos += 8 * q * e
        roots = [data[pos + 32 * i:pos + 32 * i + 32] for i in range(rounds)]; pos += 32 * rounds
        arthur = data[pos:pos + la]; pos += la
        return StarkProof(arthur, tc, lc, cq, vq, FriProof(data[pos:pos + lb], device_resident=False), roots)


class StarkConfig:
    """src/starks.rs:238-333."""

    def __init__(self, ctx: Context, security_bits: int, blowup_factor: int, steps: int, trace_columns: int):
        rc, cq, fq = ctx.num_queries(security_bits, blowup_factor, steps)  # starks.rs:274-275, 312-332
        if rc != 0:
            raise MsError(rc, "STARK Config: security bits has to be at least 20")  # starks.rs:317-320
        self.ctx = ctx
        self.security_bits, self.blowup_factor, self.steps = security_bits, blowup_factor, steps
        self.constrain_queries, self.fri_queries = cq, fq
        self.degree = steps - 1                                        # starks.rs:276
        self.rounds = ctx.ceil_log2_k(steps * blowup_factor + 1, 2)    # starks.rs:277
        self.trace_columns = trace_columns                             # merkle_config.leafs_per_node, starks.rs:297-302
        self.domsep = "\U0001F43A"                                     # starks.rs:307


class Stark:
    def __init__(self, config: StarkConfig):
        self.cfg = config

    def prove(self, trace: TraceTable, trace_device_ptr=None, read_fri_proof=True) -> StarkProof:
        """src/starks.rs:59-169.  `trace_device_ptr`: the same matrix already resident in HBM."""
        cfg, ctx = self.cfg, self.cfg.ctx
        p, e = _MODULUS[ctx.field], ctx.e
        t = Transcript(cfg.domsep)
        # 1.1 commit to the raw trace (starks.rs:68-73)
        if trace_device_ptr is not None:
            rc, trace_commit = ctx.trace_commit_device(trace_device_ptr, trace.length, trace.width, cfg.trace_columns)
        else:
            rc, trace_commit = ctx.trace_commit(trace.data, cfg.trace_columns)
        ctx.check(rc)
        t.add_bytes(trace_commit)
        # 1.2 coset LDE of the constraint polynomials + commit (starks.rs:80-95)
        (shift,) = t.challenge_scalars(1, p)
        shift = shift or 1
        ctx.check(ctx.interpolate())
        for sc, idx in trace.transitions:
            ctx.check(ctx.polys_lincomb(sc, idx))
        rc, lde_commit = ctx.lde_commit(cfg.blowup_factor, shift, cfg.trace_columns)
        ctx.check(rc)
        t.add_bytes(lde_commit)
        # 1.3 mix (starks.rs:108-119)
        (r,) = t.challenge_scalars(1, p)
        ctx.check(ctx.mix(r))
        # 2. DEEP-ALI queries (starks.rs:124-151)
        z = t.challenge_scalars(cfg.constrain_queries * e, p)
        rc, ev = ctx.eval_ext(z)
        ctx.check(rc)
        c = ctx.polys_count()
        # 3. FRI (starks.rs:155-156 -> fri.rs:53-62)
        roots = []
        rc, root0 = ctx.fri_begin(cfg.blowup_factor, cfg.rounds)  # fri.rs:73-82
        ctx.check(rc)
        roots.append(root0)
        for _ in range(1, cfg.rounds):                            # fri.rs:85-110
            zq = t.challenge_scalars(e, p)
            rc, B = ctx.fri_deep(zq)
            ctx.check(rc)
            t.add_scalars(B)
            alpha = t.challenge_scalars(e, p)
            rc, root = ctx.fri_fold_commit(alpha)
            ctx.check(rc)
            t.add_bytes(root)
            roots.append(root)
        raw = t.challenge_bytes(8 * cfg.fri_queries)              # fri.rs:121-126
        betas = [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(cfg.fri_queries)]
        rc, blob = ctx.fri_query(betas, read=read_fri_proof)
        ctx.check(rc)
        self.last_challenges = dict(shift=shift, r=r, z=z, betas=betas)
        self.last_transcript_ops = list(t.ops)
        return StarkProof(bytes(t.prover_bytes), trace_commit, lde_commit, ev[:, :c, :], ev[:, c, :],
                          FriProof(blob or b"", device_resident=not read_fri_proof), roots)


def fibonacci_air(ctx: Context, steps: int, secret_b: int = 2, pad_seed: int = 0x5EED) -> TraceTable:
    """The Fibonacci AIR of tests/e2e_goldilocks.rs:20-63 (w = 3, three transition closures)."""
    p = _MODULUS[ctx.field]
    tt = TraceTable(ctx, steps, 3)
    try:   # C helper of the host mirror (16 s -> 0.1 s at 2^24 rows); same values as synthetic.fibonacci_rows (tests/test_host_mirror.py)
        from .host import fibonacci_rows_native, host_library_path
        import os
        if not os.path.exists(host_library_path()):
            raise ImportError
        tt.data[:] = fibonacci_rows_native(p, tt.length, steps, secret_b, pad_seed)
    except (ImportError, OSError):
        from .synthetic import fibonacci_rows
        tt.data[:] = fibonacci_rows(p, tt.length, steps, secret_b, pad_seed)
    m1 = p - 1
    tt.add_transition_lincomb([tt.omega, m1], [0, 1])      # e2e_goldilocks.rs:48-51
    tt.add_transition_lincomb([tt.omega, m1], [0, 1])      # e2e_goldilocks.rs:53-56 (identical, quirk Q2)
    tt.add_transition_lincomb([1, m1, m1], [2, 0, 1])      # e2e_goldilocks.rs:57-59
    return tt
