"""C++ host mirror (mini-stark_amd/host/stark_host.cpp) vs the Python mirror (mini-stark_amd/stark.py): same
transcript, same stage order, byte-identical proofs; and the reference's own e2e configuration
(tests/e2e_goldilocks.rs:65-75,98-114: steps 9, blowup 2, 20 bits; e2e_babybear.rs: steps 7) proves and its FRI
part verifies under the oracle's verifier restatement."""
import os
import subprocess

import numpy as np
import pytest

import mini_stark_amd as ms
from mini_stark_amd.host import HostStark, build_host_library
from mini_stark_amd.stark import Stark, StarkConfig, fibonacci_air
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "libministark_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    build_host_library()
    return EMU


def check_pair(ctx, steps, blowup):
    tt = fibonacci_air(ctx, steps)
    cols = tt.constrain_number()
    py = Stark(StarkConfig(ctx, 20, blowup, steps, cols)).prove(tt)
    hs = HostStark(ctx, 20, blowup, steps, cols)
    cc = hs.prove(tt)
    assert (hs.rounds, hs.constrain_queries, hs.fri_queries) == (StarkConfig(ctx, 20, blowup, steps, cols).rounds,) + tuple(ctx.num_queries(20, blowup, steps)[1:])
    assert cc.arthur == py.arthur and cc.trace_commit == py.trace_commit and cc.constrain_trace_commit == py.constrain_trace_commit
    assert (cc.constrain_queries == py.constrain_queries).all() and (cc.validity_queries == py.validity_queries).all()
    assert cc.fri_roots == py.fri_roots and cc.fri_proof.blob == py.fri_proof.blob
    return hs, cc


@pytest.mark.parametrize("field,steps,blowup", [(0, 9, 2), (1, 7, 2), (0, 63, 8), (1, 31, 4)])
def test_host_mirror_matches_python_mirror(emu, field, steps, blowup):
    ctx = ms.Context(field, lib_path=emu)
    hs, proof = check_pair(ctx, steps, blowup)
    # FRI part of the proof under the verifier restatement (fri.rs:191-245), challenges as recorded by the prover
    e, ch = ctx.e, [int(v) for v in hs.last_challenges]
    q = hs.constrain_queries
    pos = 2 + q * e
    zs, als = [], []
    for _ in range(1, hs.rounds):
        zs += ch[pos:pos + e]; als += ch[pos + e:pos + 2 * e]; pos += 2 * e
    betas = ch[pos:]
    assert len(betas) == hs.fri_queries
    # B values are the prover messages in the transcript: arthur = trace_root | lde_root | per round (B (2e u64) | root)
    ar = proof.arthur
    Bs, off = [], 64
    for _ in range(1, hs.rounds):
        Bs += list(np.frombuffer(ar[off:off + 16 * e], dtype=np.uint64)); off += 16 * e + 32
    assert orc.fri_verify(field, e, hs.rounds, betas, zs, [int(b) for b in Bs], als, b"".join(proof.fri_roots), proof.fri_proof.blob) == 1


def test_low_security_bits_rejected(emu):
    ctx = ms.Context(0, lib_path=emu)
    with pytest.raises(ms.MsError):
        HostStark(ctx, 19, 2, 9, 6)  # starks.rs:341-346 should_panic
    with pytest.raises(ms.MsError):
        StarkConfig(ctx, 1, 4, 128, 6)


@pytest.mark.parametrize("field", [0, 1])
def test_stark_verify_deep_ali_and_mont64_input(emu, field):
    """(i) The OOD values of a proof satisfy the DEEP-ALI half of Stark::verify (src/starks.rs:204-225) restated in
    the oracle, and a tampered value is rejected.  (ii) A trace handed over in arkworks' Montgomery form
    (MS_FLAG_TRACE_MONT64) gives the same proof as the canonical one."""
    from common import MODULUS
    p = MODULUS[field]
    ctx = ms.Context(field, lib_path=emu)
    steps, blowup = 63, 8
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, blowup, steps, tt.constrain_number())
    proof = hs.prove(tt)
    ch = [int(v) for v in hs.last_challenges]
    r, z = ch[1], ch[2:2 + hs.constrain_queries * ctx.e]
    o = orc.Session(field)
    assert o.trace_commit(tt.data, tt.constrain_number())[0] == 0 and o.interpolate() == 0
    for sc, idx in tt.transitions:
        assert o.polys_lincomb(sc, idx) == 0
    ev = np.concatenate([proof.constrain_queries, proof.validity_queries[:, None, :]], axis=1)
    assert o.verify_ood(r, z, ev) == 1
    bad = ev.copy(); bad[0, 2, 0] ^= np.uint64(1)
    assert o.verify_ood(r, z, bad) == 0
    assert o.verify_ood((r + 1) % p, z, ev) == 0
    # Montgomery-form input
    mctx = ms.Context(field, flags=ms.FLAG_ZERO_DISPLAY_EMPTY | ms.FLAG_TRACE_MONT64, lib_path=emu)
    mt = fibonacci_air(mctx, steps)
    R = (1 << 64) % p
    mt.data[:] = np.array([[int(v) * R % p for v in row] for row in tt.data], dtype=np.uint64)
    mp = HostStark(mctx, 20, blowup, steps, mt.constrain_number()).prove(mt)
    assert mp.arthur == proof.arthur and mp.fri_proof.blob == proof.fri_proof.blob and mp.trace_commit == proof.trace_commit


@pytest.mark.parametrize("field,steps,blowup", [(0, 9, 2), (1, 7, 2), (0, 63, 8), (1, 31, 4), (0, 255, 8)])
def test_prove_then_verify_roundtrip(emu, field, steps, blowup):
    """The reference's integration tests (tests/e2e_goldilocks.rs:98-114, tests/e2e_babybear.rs): derive_constrains on the verifier's
    copy, prove, verify — here with the product's own CPU verifier (stark_host.cpp: Stark::verify / Fri::verify / check_proof),
    no oracle involved; then every part of the proof is tampered with in turn and must be rejected."""
    import copy
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, blowup, steps, tt.constrain_number())
    constrains = hs.derive_constrains(tt)
    proof = hs.prove(tt)
    assert hs.verify(constrains, proof), hs.last_verify_error
    p = 2**64 - 2**32 + 1 if field == 0 else 2013265921

    def rejected(pr, cs=constrains):
        ok = hs.verify(cs, pr)
        assert not ok and hs.last_verify_error
        return hs.last_verify_error

    bad = copy.deepcopy(proof); bad.trace_commit = bytes([proof.trace_commit[0] ^ 1]) + proof.trace_commit[1:]
    assert "trace commit" in rejected(bad)
    bad = copy.deepcopy(proof); bad.constrain_queries = proof.constrain_queries.copy(); bad.constrain_queries[0, 1, 0] = (int(bad.constrain_queries[0, 1, 0]) + 1) % p
    assert "evaluation" in rejected(bad)
    bad = copy.deepcopy(proof); bad.validity_queries = proof.validity_queries.copy(); bad.validity_queries[0, 0] = (int(bad.validity_queries[0, 0]) + 1) % p
    assert "validity" in rejected(bad)
    bad = copy.deepcopy(proof); a = bytearray(proof.arthur); a[70] ^= 1; bad.arthur = bytes(a)          # a DEEP coefficient of round 1
    rejected(bad)
    bad = copy.deepcopy(proof); r = bytearray(proof.fri_roots[1]); r[5] ^= 1; bad.fri_roots = [proof.fri_roots[0], bytes(r)] + proof.fri_roots[2:]
    assert "root" in rejected(bad)
    blob = proof.fri_proof.blob
    e = ctx.e
    for off, what in ((8 * e, "linearity"), (8 * 5 * e, "linearity"), (len(blob) - 1, "Merkle")):
        bad = copy.deepcopy(proof); b2 = bytearray(blob); b2[off] ^= 1
        bad.fri_proof = type(proof.fri_proof)(bytes(b2), device_resident=False)
        assert what in rejected(bad)
    # the shipped quotient polynomials are only degree-bounded by the reference's verifier (`let _ = quotient / vanishing_poly`,
    # fri.rs:219-225): a changed low coefficient is accepted there and here; a non-canonical one is malformed input
    bad = copy.deepcopy(proof); b2 = bytearray(blob); b2[(6 * e + 1) * 8] ^= 1
    bad.fri_proof = type(proof.fri_proof)(bytes(b2), device_resident=False)
    assert hs.verify(constrains, bad)
    bad = copy.deepcopy(proof); bad.fri_proof = type(proof.fri_proof)(blob[:-8], device_resident=False)
    rejected(bad)
    # a different AIR instance (other witness) does not verify against this proof
    other = hs.derive_constrains(fibonacci_air(ctx, steps, secret_b=3))
    assert "evaluation" in rejected(proof, other)
    # round 0's root is NOT bound by the transcript (fri.rs:73-82 never sends it): a wrong root0 is only caught by the Merkle paths
    bad = copy.deepcopy(proof); r = bytearray(proof.fri_roots[0]); r[0] ^= 1; bad.fri_roots = [bytes(r)] + proof.fri_roots[1:]
    assert "Merkle" in rejected(bad)


@pytest.mark.parametrize("field,steps", [(0, 63), (1, 31)])
def test_proof_wire_format_roundtrip(emu, field, steps):
    """StarkProof.to_bytes / from_bytes ("MSSP", mini-stark_amd/stark.py): the deserialised proof verifies, truncation is refused."""
    from mini_stark_amd.stark import StarkProof
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, 8, steps, tt.constrain_number())
    constrains = hs.derive_constrains(tt)
    proof = hs.prove(tt)
    wire = proof.to_bytes()
    back = StarkProof.from_bytes(wire)
    assert back.to_bytes() == wire and back.fri_roots == proof.fri_roots and back.fri_proof.blob == proof.fri_proof.blob
    assert hs.verify(constrains, back), hs.last_verify_error
    with pytest.raises(ValueError):
        StarkProof.from_bytes(wire[:-1])
    with pytest.raises(ValueError):
        StarkProof.from_bytes(b"XXXX" + wire[4:])


@pytest.mark.parametrize("field,steps", [(0, 63), (1, 31)])
def test_mssp_in_the_host_library(emu, field, steps):
    """the whole-proof wire format lives in libministark_host.so (msh_proof_serialize / msh_proof_parse /
    msh_stark_verify_mssp, include/ministark_host.h): byte-identical to the Python mirror's StarkProof.to_bytes, verifies from the
    bytes alone, tampering and truncation are refused."""
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, 8, steps, tt.constrain_number())
    constrains = hs.derive_constrains(tt)
    proof = hs.prove(tt)
    wire = hs.proof_bytes()
    assert wire == proof.to_bytes()
    assert hs.verify_bytes(constrains, wire), hs.last_verify_error
    bad = bytearray(wire); bad[-1] ^= 1
    assert not hs.verify_bytes(constrains, bytes(bad))
    with pytest.raises(ms.MsError):
        hs.verify_bytes(constrains, wire[:-3])


def test_native_trace_generator_matches_python():
    from mini_stark_amd.host import fibonacci_rows_native
    from mini_stark_amd.synthetic import fibonacci_rows
    for p in (2**64 - 2**32 + 1, 2013265921):
        for length, steps in ((16, 9), (64, 63), (128, 100), (1024, 1023)):
            assert (fibonacci_rows_native(p, length, steps, 5, 77) == fibonacci_rows(p, length, steps, 5, 77)).all()


@pytest.mark.parametrize("field,steps,blowup", [(0, 9, 2), (1, 7, 2), (0, 255, 8)])
def test_transcript_message_order_equals_the_reference_io_pattern(emu, field, steps, blowup):
    """SURVEY 8(f) rank 2, the checkable part: the mirrors' transcript is build-defined (nimue's source is unavailable), but its
    absorb / squeeze ORDER and COUNTS must equal the IOPattern the reference declares, label by label
    (src/fiatshamir.rs:53-63 new_stark, 100-116 add_fri).  The C++ mirror is byte-identical to the Python one (test above)."""
    from mini_stark_amd.stark import Stark, StarkConfig
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    cfg = StarkConfig(ctx, 20, blowup, steps, tt.constrain_number())
    st = Stark(cfg)
    st.prove(tt)
    e = ctx.e
    want = [("absorb", 32),                                     # add_digest(1, "commit to original trace")                      fiatshamir.rs:54
            ("squeeze_scalars", 1),                             # challenge_scalars(1, "ZK: pick random shift of domain")         :55
            ("absorb", 32),                                     # add_digest(1, "commit to quotients")                            :56
            ("squeeze_scalars", 1),                             # challenge_scalars(1, "batching: retrieve random scalar r")      :57
            ("squeeze_scalars", cfg.constrain_queries * e)]     # challenge_scalars(constrain_queries * ext_degree, "DEEP ALI")   :58-61
    for _ in range(cfg.rounds - 1):                             # add_fri                                                          :100-108
        want += [("squeeze_scalars", e),                        # challenge_scalars(1 extension element, "(DEEP) FRI: pick random z")
                 ("absorb", 2 * e * 8),                         # add_scalars(2, "(DEEP) FRI: degree one B polynomial")
                 ("squeeze_scalars", e),                        # challenge_scalars(1, "FRI COMMIT Phase: random scalar challenge")
                 ("absorb", 32)]                                # add_digest(1, "FRI COMMIT Phase: commit to folded codeword")
    want.append(("squeeze_bytes", 8 * cfg.fri_queries))        # challenge_bytes(8 * queries, "FRI QUERY Phase ...")              :110-113
    assert st.last_transcript_ops == want


def test_mssp_roundtrip_across_processes_emulation(emu, tmp_path):
    """The cross-process MSSP flow of the GPU suite, rehearsed on the emulation build."""
    import subprocess
    import sys
    ctx = ms.Context(0, lib_path=emu)
    tt = fibonacci_air(ctx, 63)
    hs = HostStark(ctx, 20, 8, 63, tt.constrain_number())
    hs.prove(tt)
    f = tmp_path / "proof.mssp"
    f.write_bytes(hs.proof_bytes())
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "verify_worker.py"), str(f), "0", "63", "8", emu],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "VERIFY accepted tampered-rejected" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize("field", [0, 1])
def test_async_proof_readback_matches_blocking(emu, field):
    """read_fri_proof="async" (ms_fri_proof_read_async on the context's copy stream, the next prove issued before the bytes are awaited)
    delivers the same FRI proof bytes as the blocking read-back, proof after proof on one context."""
    ctx = ms.Context(field, lib_path=emu)
    steps, blowup = 63, 8
    tts = [fibonacci_air(ctx, steps, secret_b=b) for b in (2, 5, 9)]
    hs = HostStark(ctx, 20, blowup, steps, tts[0].constrain_number())
    want = [hs.prove(tt).fri_proof.blob for tt in tts]
    assert len(set(want)) == 3
    for k, tt in enumerate(tts):   # asynchronous: the accessor waits
        ctx.check(hs.prove_raw(tt, read_fri_proof="async"))
        assert hs.last_proof().fri_proof.blob == want[k]
    for tt in tts:                 # back to back without looking at the bytes in between, explicit wait at the end
        ctx.check(hs.prove_raw(tt, read_fri_proof="async"))
    assert hs.wait_proof() == 0
    assert hs.last_proof().fri_proof.blob == want[-1]
    ctx.close()


@pytest.mark.parametrize("field", [0, 1])
def test_proof_written_into_pinned_memory_and_two_proof_slots(emu, field):
    check_into_and_slots(ms.Context(field, lib_path=emu), 63)


def check_into_and_slots(ctx, steps):
    """read_fri_proof="into" (ms_fri_query_into: the query-phase kernels write the MSFP blob straight into the mirror's page-locked slot, no
    read-back copy) delivers the bytes of the blocking read-back; and the mirror's TWO proof slots keep proof k whole - transcript,
    roots and FRI blob - while proof k + 1 is computed, in every read-back mode (in async mode the next prove starts before the bytes arrived)."""
    blowup = 8
    tts = [fibonacci_air(ctx, steps, secret_b=b) for b in (2, 5, 9, 11)]
    hs = HostStark(ctx, 20, blowup, steps, tts[0].constrain_number())
    want = [hs.prove(tt).fri_proof.blob for tt in tts]
    assert len(set(want)) == 4
    for mode in ("into", "async", True):
        sums = []
        for k, tt in enumerate(tts):
            ctx.check(hs.prove_raw(tt, read_fri_proof=mode))
            if k:   # proof k - 1 is still there, untouched by proof k (which may still be arriving: nothing has waited for it yet)
                assert hs.prev_fri_blob() == want[k - 1], (mode, k)
                assert hs.blob_checksum(1) == sums[-1]
            sums.append(hs.blob_checksum(0))
            assert hs.last_proof().fri_proof.blob == want[k], (mode, k)
        assert len(set(sums)) == 4
    # the library refuses a read-back of a proof it wrote into the caller's buffer, and a buffer that is too small
    import ctypes as C
    ctx.check(hs.prove_raw(tts[0], read_fri_proof="into"))
    buf = (C.c_uint8 * 16)()
    assert ctx.L.ms_fri_proof_read(ctx.h, buf) == ms.ERR_STATE
    n = C.c_size_t(0)
    betas = (C.c_uint64 * hs.fri_queries)(*([3] * hs.fri_queries))
    assert ctx.L.ms_fri_query_into(ctx.h, betas, C.c_int(hs.fri_queries), None, C.c_size_t(0), C.byref(n)) == 0 and n.value == len(want[0])
    assert ctx.L.ms_fri_query_into(ctx.h, betas, C.c_int(hs.fri_queries), buf, C.c_size_t(16), C.byref(n)) == -5
    ctx.close()


@pytest.mark.parametrize("field,steps", [(0, 63), (1, 31)])
def test_mssp_crafted_length_fields_are_refused(emu, field, steps):
    """a single wrapping sum used to accept arthur_len = 2^64 - 1000 with fri_blob_len enlarged to compensate, and the
    verifier then died in std::vector::assign (an exception across the C boundary).  Every length field is now checked against the
    bytes left, and q / rounds must equal the handle's configuration; a malformed proof is MS error -1, never a crash."""
    import struct
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, 8, steps, tt.constrain_number())
    constrains = hs.derive_constrains(tt)
    hs.prove(tt)
    wire = hs.proof_bytes()
    assert hs.verify_bytes(constrains, wire)
    la, lb = struct.unpack_from("<QQ", wire, 24)

    def with_header(**kw):
        b = bytearray(wire)
        for off, fmt, key in ((8, "<I", "e"), (12, "<I", "c"), (16, "<I", "q"), (20, "<I", "rounds"), (24, "<Q", "la"), (32, "<Q", "lb")):
            if key in kw:
                struct.pack_into(fmt, b, off, kw[key] & ((1 << (32 if fmt == "<I" else 64)) - 1))
        return bytes(b)
    crafted = [with_header(la=(1 << 64) - 1000, lb=lb + la + 1000),     # the reported wrap: the sum comes back to len
               with_header(la=la + (1 << 63), lb=lb + (1 << 63)),       # both huge, sum wraps to the same total
               with_header(la=la + 8, lb=lb - 8),                       # boundary moved: parses, must not verify
               with_header(q=0xFFFFFFFF), with_header(q=hs.constrain_queries + 1), with_header(rounds=0xFFFFFFFF), with_header(rounds=hs.rounds + 1),
               with_header(c=0xFFFFFFFF), with_header(c=0), with_header(e=0), with_header(e=3), with_header(e=0x80000002),
               with_header(lb=(1 << 64) - 1), with_header(la=(1 << 64) - 1), wire + b"\0", wire[:40], b"MSSP"]
    for i, bad in enumerate(crafted):
        try:
            ok = hs.verify_bytes(constrains, bad)
        except ms.MsError:
            continue
        assert not ok, f"crafted proof {i} was accepted"


@pytest.mark.parametrize("field,steps", [(0, 63), (1, 31)])
def test_fri_proof_parse_walks_the_msfp_blob(emu, field, steps):
    """msh_fri_proof_parse (the compiled twin of the Rust shim's FriProof::from_msfp): record count, points, quotient lengths and path
    depths of a real proof; truncation / trailing bytes / wrong shape are refused."""
    import ctypes as C
    from mini_stark_amd.host import _host
    ctx = ms.Context(field, lib_path=emu)
    tt = fibonacci_air(ctx, steps)
    hs = HostStark(ctx, 20, 8, steps, tt.constrain_number())
    proof = hs.prove(tt)
    blob = proof.fri_proof.blob
    e, W, nq = ctx.e, hs.rounds - 1, hs.fri_queries

    class Path(C.Structure):
        _fields_ = [("leaf_index", C.c_uint64), ("nlevels", C.c_uint64), ("leaf_neighbours", C.POINTER(C.c_uint64)), ("levels", C.POINTER(C.c_uint8))]

    class Rec(C.Structure):
        _fields_ = [("points", C.POINTER(C.c_uint64)), ("qlen", C.c_uint64), ("quotient", C.POINTER(C.c_uint64)), ("path", Path * 2)]
    H = _host()
    H.msh_fri_proof_parse.restype = C.c_int
    recs = (Rec * (W * nq))()

    def parse(b, e_=e, W_=W, nq_=nq):
        return H.msh_fri_proof_parse(b, C.c_size_t(len(b)), C.c_uint32(e_), C.c_uint32(W_), C.c_uint32(nq_), recs, C.c_size_t(W * nq))
    assert parse(blob) == W * nq
    L = 8 * (steps + 1)
    for i in range(W):
        D = L >> i
        for j in range(nq):
            r = recs[i * nq + j]
            x1, x2, x3 = r.points[0], r.points[2 * e], r.points[4 * e]
            P = 2**64 - 2**32 + 1 if field == 0 else 2013265921
            assert (x1 + x2) % P == 0 and x3 == x1 * x1 % P                       # fri.rs:148-150: x2 = -x1, x3 = x1^2
            for s in range(2):
                assert r.path[s].nlevels == max(0, (D // 2).bit_length() - 1)      # binary tree over D/2 leaf groups (merkle.rs:272-288)
                assert r.path[s].leaf_index < D                                     # index of the opened LEAF (element), merkle.rs:216-225
    assert parse(blob[:-1]) == -1 and parse(blob + b"\0") == -1 and parse(blob, W_=W - 1) == -1 and parse(blob, nq_=nq + 1) == -1 and parse(blob, e_=3) == -1
    assert parse(b"") == -1 if W * nq else parse(b"") == 0
