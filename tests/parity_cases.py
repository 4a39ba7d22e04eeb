"""Parity cases shared by the CPU-emulation suite (tests/test_emu_parity.py) and
the GPU suite (tests/test_gpu_parity.py): the C-ABI library (HIP build, or the
emulation build of the same kernel code) against the CPU oracle on the same
seeded inputs, bit-exact.  `mk(field)` returns a mini_stark_amd.Context."""
import ctypes as C

import numpy as np

from common import MODULUS, EXT, SplitMix64, fibonacci_trace, fibonacci_trace_fast, fibonacci_closures
from oracle import oracle as orc
ERR_SHAPE = -1


def rand_field(field, shape, seed):
    rng = np.random.default_rng(seed)
    p = MODULUS[field]
    a = rng.integers(0, 2**63, shape, dtype=np.uint64)
    b = rng.integers(0, 2, shape, dtype=np.uint64)
    return ((a << np.uint64(1)) | b) % np.uint64(p)


def case_ntt(mk, field, log_n, batch=2):
    ctx = mk(field)
    n = 1 << log_n
    a = rand_field(field, (batch, n), seed=log_n)
    rc, out = ctx.ntt(a)
    assert rc == 0, ctx.last_error()
    for i in range(batch):
        assert (out[i] == orc.ntt(field, a[i])).all()
    rc, inv = ctx.ntt(out, inverse=True)
    assert rc == 0 and (inv == a).all()
    rc, out2 = ctx.ntt(a, inverse=True)
    assert rc == 0
    assert (out2[0] == orc.intt(field, a[0])).all()


def case_coset_lde(mk, field, log_n, blowup):
    ctx = mk(field)
    n = 1 << log_n
    coeffs = rand_field(field, (3, n), seed=100 + log_n)
    shift = int(rand_field(field, (1,), seed=5)[0]) or 3
    rc, out = ctx.coset_lde(coeffs, shift, n * blowup)
    assert rc == 0, ctx.last_error()
    for i in range(3):
        assert (out[i] == orc.coset_lde(field, coeffs[i], shift, n * blowup)).all()


def case_merkle(mk, field, leaf_num, ext, lpn, ic, special=False):
    ctx = mk(field)
    leafs = rand_field(field, (leaf_num * ext,), seed=leaf_num + lpn)
    if special:  # zeros, small values, p-1, digit-count boundaries
        p = MODULUS[field]
        vals = [0, 1, p - 1, 2**32 % p, 12345678901234567890 % p, 1000100010001 % p, 10203040506070809 % p]
        vals += [v % p for k in range(1, 20) for v in (10**k - 1, 10**k, 10**k + 1, 7 * 10**k)]  # every digit-count boundary
        # chunk boundaries of the kernels' decimal conversion (estimated quotients by 10^16 / 10^8 / 10^4 with fix-ups)
        for a in (0, 1, 2, 999, 1000, 1843, 1844):
            for b in (0, 1, 9999, 10000, 99999999):
                for c in (0, 1, 9999, 10000, 99999999):
                    vals.append(a * 10**16 + b * 10**8 + c)
        vals += [p - 2, 2**63 % p, (2**64 - 2**32) % p, 2**54 - 1, 2**54, 10**16 - 2**22, 10**16 + 2**22 - 1]
        for i, v in enumerate(vals):
            leafs[i % leafs.size] = v % p
    rc, nodes, root = ctx.merkle_commit(leafs, ext, lpn, ic)
    orc_rc, onodes, oroot = orc.merkle_build(leafs, ext, lpn, ic)
    assert (rc == 0) == (orc_rc == 0), (rc, orc_rc, ctx.last_error())
    if rc == 0:
        assert (nodes == onodes).all()
        assert root == oroot


def drive(sess, field, trace, blowup, nq_fri, seed, rounds=None, q_ood=2, read_big=True, base_field_z_rounds=(), fixed_betas=None):
    """Runs one full prove on `sess` (oracle Session or mini_stark_amd Context) with challenges
    drawn from SplitMix64(seed); returns the list of stage outputs."""
    p, e = MODULUS[field], EXT[field]
    N, w = trace.shape
    rng = SplitMix64(seed)
    omega = orc.root_of_unity(field, N)
    out = []
    rc, root = sess.trace_commit(trace, 2 * w)
    assert rc == 0
    out.append(("trace_root", root))
    assert sess.interpolate() == 0
    for sc, idx in fibonacci_closures(field, N, omega):
        assert sess.polys_lincomb(sc, idx) == 0
    if read_big:
        for i in range(sess.polys_count()):
            out.append((f"poly{i}", sess.poly_read(i).tolist()))
    shift = rng.nonzero(p)
    rc, root = sess.lde_commit(blowup, shift, 2 * w)
    assert rc == 0
    out.append(("lde_root", root))
    if read_big:
        out.append(("lde", sess.lde_read().tolist()))
    assert sess.mix(rng.field(p)) == 0
    if read_big:
        out.append(("validity", sess.validity_read().tolist()))
    zs = np.array([[rng.field(p) for _ in range(e)] for _ in range(q_ood)], dtype=np.uint64)
    rc, ev = sess.eval_ext(zs)
    assert rc == 0
    out.append(("ood", ev.tolist()))
    if rounds is None:
        rounds = int(orc.lib().or_ceil_log2_k(C.c_uint64((N - 1) * blowup + 1), C.c_uint64(2)))
    rc, root = sess.fri_begin(blowup, rounds)
    assert rc == 0
    out.append(("fri_root0", root))
    for i in range(1, rounds):
        z = [rng.field(p) for _ in range(e)]
        if i in base_field_z_rounds:  # a DEEP point in the base field: the kernels' evaluation-domain fold must fall back to the transform
            z = [z[0]] + [0] * (e - 1)
        rc, B = sess.fri_deep(z)
        assert rc == 0
        out.append((f"B{i}", B.tolist()))
        rc, root = sess.fri_fold_commit([rng.field(p) for _ in range(e)])
        assert rc == 0
        out.append((f"fri_root{i}", root))
    for i in range(rounds):
        out.append((f"round_info{i}", tuple(sess.fri_round_info(i))))
        if read_big:
            out.append((f"round_poly{i}", sess.fri_round_poly(i).tolist()))
            out.append((f"round_cw{i}", sess.fri_round_codeword(i).tolist()))
    betas = [rng.next() for _ in range(nq_fri)] + list([3, 2 * N * blowup] if fixed_betas is None else fixed_betas)
    rc, proof = sess.fri_query(betas)
    assert rc == 0
    out.append(("fri_proof", proof))
    return out


def case_merkle_length_sweep(mk, field, lpn=6, ext=1):
    """Leaf groups whose decimal strings sweep every message length around the SHA-256 block boundaries (55/56, 119/120, ...):
    exercises the kernels' on-the-fly padding and the deferred pad-only block."""
    ctx = mk(field)
    p = MODULUS[field]
    maxd = len(str(p - 1))
    groups = 256
    leafs = np.zeros(groups * lpn * ext, dtype=np.uint64)
    per = lpn * ext
    for g in range(groups):
        total = per + (g * 7) % (per * (maxd - 1) + 1)      # total digits of the group, from all-1-digit to all-max-digit
        digs = [1] * per
        left = total - per
        k = 0
        while left > 0:
            add = min(maxd - 1, left)
            digs[k % per] += add
            left -= add
            k += 1
        for j, d in enumerate(digs):
            v = 10 ** (d - 1) + (g * 31 + j) % 9
            leafs[g * per + j] = v if v < p else p - 1 - j
    rc, nodes, root = ctx.merkle_commit(leafs, ext, lpn, 2)
    orc_rc, onodes, oroot = orc.merkle_build(leafs, ext, lpn, 2)
    assert rc == 0 and orc_rc == 0, ctx.last_error()
    assert (nodes == onodes).all() and root == oroot


def case_prove(mk, field, log_n, blowup, nq_fri=2, seed=77, read_big=True, steps=None):
    N = 1 << log_n
    if steps is None:
        trace = fibonacci_trace_fast(field, N)
    else:  # several padding rows (air.rs:79-83): steps + 1 <= N, steps >= N / 2
        from mini_stark_amd.synthetic import fibonacci_rows
        trace = fibonacci_rows(MODULUS[field], N, steps, secret_b=seed, pad_seed=seed * 3 + 1)
    a = drive(mk(field), field, trace, blowup, nq_fri, seed, read_big=read_big)
    b = drive(orc.Session(field), field, trace, blowup, nq_fri, seed, read_big=read_big)
    assert len(a) == len(b)
    for (ka, va), (kb, vb) in zip(a, b):
        assert ka == kb
        assert va == vb, f"stage output {ka} differs"


def case_prove_base_field_deep_points(mk, field, log_n=6, blowup=8):
    """FRI rounds whose DEEP point z lies in the base field (and rounds mixing both kinds)."""
    trace = fibonacci_trace_fast(field, 1 << log_n)
    a = drive(mk(field), field, trace, blowup, 1, seed=3, base_field_z_rounds=(1, 3, 4))
    b = drive(orc.Session(field), field, trace, blowup, 1, seed=3, base_field_z_rounds=(1, 3, 4))
    assert a == b


def case_errors(mk, field):
    ctx = mk(field, fresh=True)
    import mini_stark_amd as ms
    t = fibonacci_trace(field, 8)
    assert ctx.interpolate() == ms.ERR_STATE
    assert ctx.trace_commit(t[:6], 6)[0] == ms.ERR_SHAPE          # air.rs:23 length not a power of two
    assert ctx.trace_commit(t, 5)[0] == ms.ERR_SHAPE              # merkle.rs:93-104 tree not full
    bad = t.copy(); bad[0, 0] = MODULUS[field]
    assert ctx.trace_commit(bad, 6)[0] == ms.ERR_ARG
    assert ctx.trace_commit(t, 6)[0] == 0
    assert ctx.mix(1) == ms.ERR_STATE
    assert ctx.interpolate() == 0
    assert ctx.lde_commit(3, 5, 3)[0] == ms.ERR_ARG               # blowup not a power of two
    assert ctx.lde_commit(2, 0, 3)[0] == ms.ERR_ARG
    assert ctx.lde_commit(2, 5, 5)[0] == ms.ERR_SHAPE
    assert ctx.fri_begin(2, 3)[0] == ms.ERR_STATE
    assert ctx.mix(3) == 0
    assert ctx.fri_fold_commit([1] * ctx.e)[0] == ms.ERR_STATE
    assert ctx.fri_begin(2, 4)[0] == 0
    assert ctx.fri_query([1])[0] == ms.ERR_STATE                  # commit phase unfinished
    assert ctx.merkle_commit(np.arange(3, dtype=np.uint64), 1, 2, 2)[0] == ms.ERR_SHAPE
    assert ctx.merkle_commit(np.zeros(0, dtype=np.uint64), 1, 2, 2)[0] == ms.ERR_SHAPE
    assert ctx.num_queries(1, 4, 128)[0] == ms.ERR_SHAPE          # starks.rs:341-346
    assert ctx.num_queries(20, 4, 129) == (0, 1, 3) if field == 0 else True
    # the reference panics (division by zero / overflow) where these used to SIGFPE or wrap across the C ABI
    assert ctx.num_queries(20, 4, 2**63 if field == 1 else 2**64 - 1)[0] == ms.ERR_SHAPE   # log_steps >= modulus_bits
    assert ctx.num_queries(20, 2**40, 2**30)[0] == ms.ERR_SHAPE                              # steps * blowup overflows u64


def case_prove_wide(mk, field, log_n=8, w=64, blowup=8, seed=5):
    """BASELINE configs[4] shape: 64 trace columns + 64 transition polynomials (c = 128), random trace; the transition
    polynomials are linear (quirk Q1: degree-3 constraints are not expressible in the reference)."""
    p, e = MODULUS[field], EXT[field]
    N = 1 << log_n
    trace = rand_field(field, (N, w), seed=seed)
    rng = SplitMix64(seed)
    combos = [([rng.nonzero(p), rng.field(p), p - 1], [j, (j + 1) % w, (j + 7) % w]) for j in range(w)]
    outs = []
    for sess in (mk(field), orc.Session(field)):
        r2 = SplitMix64(seed + 1)
        o = []
        rc, root = sess.trace_commit(trace, 2 * w)
        assert rc == 0
        o.append(root)
        assert sess.interpolate() == 0
        for sc, idx in combos:
            assert sess.polys_lincomb(sc, idx) == 0
        assert sess.polys_count() == 2 * w
        rc, root = sess.lde_commit(blowup, r2.nonzero(p), 2 * w)
        assert rc == 0
        o.append(root)
        assert sess.mix(r2.field(p)) == 0
        rc, ev = sess.eval_ext(np.array([r2.field(p) for _ in range(e)], dtype=np.uint64))
        assert rc == 0
        o.append(ev.tolist())
        rounds = log_n + 3
        rc, root = sess.fri_begin(blowup, rounds)
        assert rc == 0
        o.append(root)
        for _ in range(1, rounds):
            rc, B = sess.fri_deep([r2.field(p) for _ in range(e)])
            assert rc == 0
            rc, root = sess.fri_fold_commit([r2.field(p) for _ in range(e)])
            assert rc == 0
            o.append((B.tolist(), root))
        rc, proof = sess.fri_query([r2.next()])
        assert rc == 0
        o.append(proof)
        outs.append(o)
    assert outs[0] == outs[1]


def case_general_closure(mk, field, log_n=6):
    """A transition closure evaluated on the HOST (air.rs:130-134: closures see coefficient vectors): read the trace
    polynomials back, combine them, ms_polys_append — must equal the on-device ms_polys_lincomb of the same closure."""
    p = MODULUS[field]
    N = 1 << log_n
    trace = fibonacci_trace(field, N)
    omega = orc.root_of_unity(field, N)
    a, b = mk(field), mk(field, fresh=True)
    for s in (a, b):
        assert s.trace_commit(trace, 6)[0] == 0 and s.interpolate() == 0
    assert a.polys_lincomb([omega, p - 1], [0, 1]) == 0
    P0, P1 = [int(v) for v in b.poly_read(0)], [int(v) for v in b.poly_read(1)]
    host = np.array([(omega * x - y) % p for x, y in zip(P0, P1)], dtype=np.uint64)
    assert b.polys_append(host) == 0
    assert (a.poly_read(3) == b.poly_read(3)).all()
    assert a.lde_commit(4, 3, 4)[1] == b.lde_commit(4, 3, 4)[1]
    import mini_stark_amd as ms
    assert b.polys_append(np.zeros(N + 1, dtype=np.uint64)) == ms.ERR_SHAPE   # more than N coefficients: starks.rs:118-119 would panic


def case_merkle_prove(mk, field, leaf_num=64, ext=1, lpn=2):
    """MerkleTree::generate_proof by leaf value (merkle.rs:272-288) incl. duplicates (first match wins, quirk Q7)."""
    import mini_stark_amd as ms
    ctx = mk(field)
    leafs = rand_field(field, (leaf_num, ext), seed=leaf_num + ext)
    leafs[leaf_num // 2] = leafs[3]            # duplicate value: index 3 must be reported
    flat = leafs.reshape(-1)
    rc, nodes, root = orc.merkle_build(flat, ext, lpn, 2)
    assert rc == 0
    for idx in (0, 3, leaf_num // 2, leaf_num - 1):
        rc, path = ctx.merkle_prove(flat, leafs[idx], ext, lpn)
        orc_rc, opath = orc.merkle_prove(flat, leafs[idx], ext, lpn, 2)
        assert rc == 0 and orc_rc == 0, ctx.last_error()
        assert path == opath
        assert orc.merkle_check_proof(root, path, ext, lpn, 2)
    missing = (leafs[0] + np.uint64(1)) % np.uint64(MODULUS[field])
    if not (leafs == missing).all(axis=1).any():
        assert ctx.merkle_prove(flat, missing, ext, lpn)[0] == ms.ERR_LEAF_NOT_FOUND


def case_lincomb_many_terms(mk_fresh, field, k, linear, log_n=5, blowup=4):
    """ms_polys_lincomb with k terms (k >= MAX_TERMS = 8 needs several kernel launches chained through the partial result):
    the constraint polynomial AND its LDE column, with the linear-provenance shortcut on (MS_LDE_LINEAR=1) and off (0), against the
    oracle.  `mk_fresh(field)` must create a NEW context (the knob is read at ms_create)."""
    import os
    p = MODULUS[field]
    N = 1 << log_n
    w = 4
    trace = rand_field(field, (N, w), seed=1000 + k)
    rng = SplitMix64(500 + k)
    sc = [rng.nonzero(p) for _ in range(k)]
    idx = [int(rng.next() % w) for _ in range(k)]
    old = os.environ.get("MS_LDE_LINEAR")
    os.environ["MS_LDE_LINEAR"] = str(linear)
    try:
        ctx = mk_fresh(field)
    finally:
        if old is None:
            del os.environ["MS_LDE_LINEAR"]
        else:
            os.environ["MS_LDE_LINEAR"] = old
    o = orc.Session(field)
    lpn = 2 * (w + 2)
    for s in (ctx, o):
        assert s.trace_commit(trace, w)[0] == 0 and s.interpolate() == 0
        assert s.polys_lincomb(sc, idx) == 0
        assert s.polys_lincomb(sc[::-1] + [1], idx[::-1] + [w]) == 0   # a second one that also combines the first combination
    for i in (w, w + 1):
        assert (ctx.poly_read(i) == o.poly_read(i)).all(), f"constraint polynomial {i} with k={k} differs"
    ra, rb = ctx.lde_commit(blowup, 7, w + 2), o.lde_commit(blowup, 7, w + 2)
    assert ra[0] == 0 and rb[0] == 0 and ra[1] == rb[1]
    assert (ctx.lde_read() == o.lde_read()).all()


def case_lincomb_shared_sweep(mk_fresh, field, log_n=6, blowup=8):
    """Seven linear constraint polynomials over a 10-column trace: the LDE stage computes their columns in shared sweeps
    (LincombMultiKernel: up to 4 outputs over up to 8 distinct sources per launch) - groups that fill up on outputs, a group that fills up on
    sources, a column named twice, a coefficient 1, and one polynomial whose source is itself a linear column (must keep its order).
    Polynomials, LDE matrix and root against the oracle; MS_LDE_MULTI=0 (one launch per column) must give the same bytes."""
    import os
    p = MODULUS[field]
    N = 1 << log_n
    w = 10
    trace = rand_field(field, (N, w), seed=4242)
    rng = SplitMix64(77)
    combos = [
        ([rng.nonzero(p), rng.nonzero(p)], [0, 1]),
        ([1, rng.nonzero(p), rng.nonzero(p)], [1, 2, 1]),                 # column 1 twice, coefficient 1
        ([rng.nonzero(p)] * 3, [3, 4, 5]),
        ([rng.nonzero(p), p - 1], [6, 7]),                                # the fourth output: sources 0..7 = 8 distinct, group full
        ([rng.nonzero(p), rng.nonzero(p)], [8, 9]),                       # does not fit the sources any more: new group
        ([rng.nonzero(p), rng.nonzero(p)], [w + 1, 2]),                   # a source that is a linear column
        ([rng.nonzero(p) for _ in range(8)], [0, 2, 4, 6, 8, 1, 3, 5]),   # eight distinct sources on its own
    ]
    outs = []
    for multi in ("1", "0"):
        old = os.environ.get("MS_LDE_MULTI")
        os.environ["MS_LDE_MULTI"] = multi
        try:
            ctx = mk_fresh(field)
        finally:
            if old is None:
                del os.environ["MS_LDE_MULTI"]
            else:
                os.environ["MS_LDE_MULTI"] = old
        assert ctx.trace_commit(trace, w)[0] == 0 and ctx.interpolate() == 0
        for sc, idx in combos:
            assert ctx.polys_lincomb(sc, idx) == 0
        r = ctx.lde_commit(blowup, 5, w + len(combos))
        assert r[0] == 0
        outs.append((r[1], ctx.lde_read()))
    o = orc.Session(field)
    assert o.trace_commit(trace, w)[0] == 0 and o.interpolate() == 0
    for sc, idx in combos:
        assert o.polys_lincomb(sc, idx) == 0
    rb = o.lde_commit(blowup, 5, w + len(combos))
    ref = o.lde_read()
    for root, lde in outs:
        assert root == rb[1]
        assert (lde == ref).all()


def case_device_trace_range_check(mk, field, to_device):
    """ms_trace_commit_device validates the canonical range on the device: a trace element >= p gives MS_ERR_ARG, like
    the host path; a clean trace gives the host path's root.  `to_device(np_u64_array) -> (pointer, keepalive)`."""
    import mini_stark_amd as ms
    ctx = mk(field, fresh=True)
    t = fibonacci_trace(field, 16)
    rc_h, root_h = ctx.trace_commit(t, 6)
    ptr, keep = to_device(np.ascontiguousarray(t))
    rc_d, root_d = ctx.trace_commit_device(ptr, 16, 3, 6)
    assert rc_h == 0 and rc_d == 0 and root_h == root_d
    bad = t.copy(); bad[5, 1] = MODULUS[field] + 3
    ptr, keep2 = to_device(np.ascontiguousarray(bad))
    assert ctx.trace_commit_device(ptr, 16, 3, 6)[0] == ms.ERR_ARG
    assert ctx.interpolate() == ms.ERR_STATE      # the failed commit left no trace behind
    del keep, keep2


def case_arith_selftest(mk, field, nrand=1 << 16):
    """ms_arith_selftest: every operation of the NTT tiles' arithmetic class (Goldilocks on the GPU: GLM, exec-masked inline asm with
    hand-managed wait states - ADVICE r2: the one class no CPU build can execute) against Python big integers, on all pairs of directed
    edge values (operands that force a carry, a borrow, both, or neither; values around p, 2^32 and 2^63) and on random pairs."""
    p = MODULUS[field]
    if field == 0:
        edge = [0, 1, 2, p - 1, p - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 1 << 63, (1 << 63) - 1, 0xFFFFFFFE, 0x1FFFFFFFF,
                0xFFFFFFFE00000001, 0xFFFFFFFE00000002, 0x7FFFFFFF80000001, 0x80000000FFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF00000000 - 1]
    else:
        edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 30, (1 << 30) - 1, 0x77FFFFFF, 0x78000000, 0x78000001 % p, 1172168163]
    edge = [v % p for v in edge]
    a = [x for x in edge for _ in edge]
    b = [y for _ in edge for y in edge]
    rng = np.random.RandomState(12345)
    ra = [int(v) % p for v in (rng.randint(0, 1 << 62, size=nrand, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rng.randint(0, 4, size=nrand).astype(np.uint64))]
    rb = [int(v) % p for v in (rng.randint(0, 1 << 62, size=nrand, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rng.randint(0, 4, size=nrand).astype(np.uint64))]
    a, b = a + ra, b + rb
    ctx = mk(field)
    ops = {0: lambda x, y: (x + y) % p, 1: lambda x, y: (x - y) % p, 2: lambda x, y: x * y % p, 3: lambda x, y: x * y % p}
    if field == 0:
        ops.update({4: lambda x, y: (x << 32) % p, 5: lambda x, y: (x << 64) % p, 6: lambda x, y: (x << (y % 96)) % p, 7: lambda x, y: (x + ((y & 0x7FFFFFFF) << 64)) % p})
    for op, f in ops.items():
        got = ctx.arith_selftest(op, a, b)
        want = np.array([f(x, y) for x, y in zip(a, b)], dtype=np.uint64)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, f"field {field} op {op}: {len(bad)} mismatches, first a={a[bad[0]]:#x} b={b[bad[0]]:#x} got {int(got[bad[0]]):#x} want {int(want[bad[0]]):#x}"
    # shift op: every exponent 0..95 on the edge values
    if field == 0:
        aa = [x for x in edge for _ in range(96)]
        ss = [s_ for _ in edge for s_ in range(96)]
        got = ctx.arith_selftest(6, aa, ss)
        want = np.array([(x << s_) % p for x, s_ in zip(aa, ss)], dtype=np.uint64)
        assert (got == want).all()
    assert ctx.L.ms_arith_selftest(ctx.h, 9, None, None, None, 0) != 0


def cubic_trace(field, N, w, seed=9):
    """A trace that satisfies the build-defined degree-3 transitions col_j[i+1] = col_j[i] * col_{j+1}[i] * col_{j+2}[i] + s_j * col_{j+3}[i] (indices mod w) on rows
    0 .. N-2: random first row, the recurrence below it."""
    p = MODULUS[field]
    rng = SplitMix64(seed)
    sc = [rng.field(p) for _ in range(w)]
    rows = [[rng.field(p) for _ in range(w)]]
    for _ in range(N - 1):
        prev = rows[-1]
        rows.append([(prev[j] * prev[(j + 1) % w] * prev[(j + 2) % w] + sc[j] * prev[(j + 3) % w]) % p for j in range(w)])
    spec = [(j, j, (j + 1) % w, (j + 2) % w, (j + 3) % w) for j in range(w)]
    return np.array(rows, dtype=np.uint64), spec, sc


def case_mix_cubic(mk, field, log_n=4, w=4, blowup=8, seed=9):
    """ms_mix_cubic (BASELINE configs[4]'s "degree-3 constraints", build-defined: the reference cannot express them) against a big-integer restatement by
    the DEFINITION: C_t(x) = P_j(w x) - P_a P_b P_c - s P_d with schoolbook products, validity = (sum r^t C_t)(x - w^(N-1)) / (x^N - 1) by long division
    (exact), 2N coefficients; then the DEEP-ALI identity validity(z) (z^N - 1) = (z - w^(N-1)) sum r^t C_t(z) from ms_eval_ext's values at an extension
    point z and at w z; a full FRI over the 2N-coefficient validity polynomial (proof accepted by the verifier restatement's FRI part); and a trace with one
    violated row must be refused (MS_ERR_SHAPE)."""
    p, e = MODULUS[field], EXT[field]
    N = 1 << log_n
    trace, spec, sc = cubic_trace(field, N, w, seed)
    ctx = mk(field)
    rng = SplitMix64(seed + 1)
    omega = orc.root_of_unity(field, N)
    rc, _ = ctx.trace_commit(trace, w)
    assert rc == 0, ctx.last_error()
    assert ctx.interpolate() == 0
    polys = [[int(v) for v in ctx.poly_read(j)] for j in range(w)]
    shift = rng.nonzero(p)
    rc, _ = ctx.lde_commit(blowup, shift, w)
    assert rc == 0, ctx.last_error()
    r = rng.field(p)
    assert ctx.mix_cubic(r, spec, sc) == 0, ctx.last_error()
    got = [int(v) for v in ctx.validity_read()]
    assert len(got) == 2 * N

    # ---- the definition, in Python integers
    def pmul(a, b):
        out = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            if x:
                for k, y in enumerate(b):
                    out[i + k] = (out[i + k] + x * y) % p
        return out

    def padd(a, b, sb=1):
        n = max(len(a), len(b))
        return [((a[i] if i < len(a) else 0) + sb * (b[i] if i < len(b) else 0)) % p for i in range(n)]
    mixed, rp = [0], 1
    for (j, a, b, c_, d), s_ in zip(spec, sc):
        shifted = [polys[j][k] * pow(omega, k, p) % p for k in range(N)]            # P_j(w x)
        C_t = padd(padd(shifted, pmul(pmul(polys[a], polys[b]), polys[c_]), -1), [s_ * v % p for v in polys[d]], -1)
        mixed = padd(mixed, [rp * v % p for v in C_t])
        rp = rp * r % p
    num = pmul(mixed, [(-pow(omega, N - 1, p)) % p, 1])
    quo = [0] * max(1, len(num) - N)
    rem = list(num)
    for k in range(len(num) - 1, N - 1, -1):                                         # divide by x^N - 1
        q = rem[k]
        quo[k - N] = q
        rem[k] = 0
        rem[k - N] = (rem[k - N] + q) % p
    assert not any(rem), "the restatement's own division must be exact for a valid trace"
    want = (quo + [0] * (2 * N))[:2 * N]
    assert got == want
    # ---- DEEP-ALI identity at an out-of-domain extension point (values at z and at w z)
    from pyref import Tower
    T = Tower(field, e)
    z = tuple(rng.field(p) for _ in range(e))
    wz = T.mul(z, T.from_base(omega))
    rc, ev = ctx.eval_ext(np.array([z, wz], dtype=np.uint64))
    assert rc == 0
    E_ = lambda v: tuple(int(x) for x in v)
    Pz, Pwz, Vz = [E_(ev[0][j]) for j in range(w)], [E_(ev[1][j]) for j in range(w)], E_(ev[0][w])
    acc, rp = T.zero(), 1
    for (j, a, b, c_, d), s_ in zip(spec, sc):
        C_t = T.sub(T.sub(Pwz[j], T.mul(T.mul(Pz[a], Pz[b]), Pz[c_])), T.mul(Pz[d], T.from_base(s_)))
        acc = T.add(acc, T.mul(C_t, T.from_base(rp)))
        rp = rp * r % p
    assert T.mul(Vz, T.sub(T.pow(z, N), T.one())) == T.mul(acc, T.sub(z, T.from_base(pow(omega, N - 1, p))))
    # ---- FRI over the 2N-coefficient validity polynomial: commit phase + query phase run, round 0 has (2N - 1 trimmed) * blowup leaves
    rounds = (2 * N * blowup).bit_length() - 1
    rc, _root0 = ctx.fri_begin(blowup, rounds)
    assert rc == 0, ctx.last_error()
    nc0, D0 = ctx.fri_round_info(0)
    assert nc0 <= 2 * N and D0 == 2 * N * blowup
    for i in range(1, rounds):
        rc, _B = ctx.fri_deep([rng.field(p) for _ in range(e)])
        assert rc == 0
        rc, _root = ctx.fri_fold_commit([rng.field(p) for _ in range(e)])
        assert rc == 0, ctx.last_error()
    rc, proof = ctx.fri_query([rng.next(), 5])
    assert rc == 0 and len(proof) > 0
    # ---- a violated row is refused, like the reference's assert on the first output of divide_by_vanishing_poly (starks.rs:119)
    bad = trace.copy()
    bad[N // 2, 0] = (int(bad[N // 2, 0]) + 1) % p
    ctx2 = mk(field, fresh=True) if "fresh" in mk.__code__.co_varnames else mk(field)
    assert ctx2.trace_commit(bad, w)[0] == 0 and ctx2.interpolate() == 0 and ctx2.lde_commit(blowup, shift, w)[0] == 0
    assert ctx2.mix_cubic(r, spec, sc) == ERR_SHAPE
    # blowup 2 cannot hold the 3N-coefficient composition
    ctx3 = mk(field, fresh=True) if "fresh" in mk.__code__.co_varnames else mk(field)
    assert ctx3.trace_commit(trace, w)[0] == 0 and ctx3.interpolate() == 0 and ctx3.lde_commit(2, shift, w)[0] == 0
    assert ctx3.mix_cubic(r, spec, sc) == ERR_SHAPE
