#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: wide synthetic AIR (64 trace columns + 64 transition polynomials, c = 128),
Goldilocks, 2^22 rows, blowup 8.  The transition polynomials are linear combinations of trace polynomials (quirk Q1:
degree-3 constraints cannot be expressed in the reference); challenges come from SplitMix64.  Times the stage calls of
one proof after a warm-up proof (trace resident in HBM) and prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mini_stark_amd as ms
from mini_stark_amd.synthetic import SplitMix64

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, default=22)
ap.add_argument("--width", type=int, default=64)
ap.add_argument("--blowup", type=int, default=8)
ap.add_argument("--proofs", type=int, default=2)
a = ap.parse_args()
P = 2**64 - 2**32 + 1
N, w = 1 << a.log_rows, a.width
rs = np.random.RandomState(7)
trace = (rs.randint(0, 2**63, size=(N, w), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(N, w)).astype(np.uint64)) % np.uint64(P)
d_trace = torch.from_numpy(trace.view(np.int64)).cuda()
ctx = ms.Context(ms.GOLDILOCKS)
rng = SplitMix64(5)
combos = [([rng.next() % P or 1, rng.next() % P, P - 1], [j, (j + 1) % w, (j + 7) % w]) for j in range(w)]
rounds = a.log_rows + 3
times = {}


def stage(name, fn):
    t = time.perf_counter(); r = fn(); ctx.synchronize(); times[name] = times.get(name, 0.0) + time.perf_counter() - t
    return r


for it in range(a.proofs + 1):
    if it == 1:
        times.clear(); t0 = time.perf_counter()
    r2 = SplitMix64(6)
    assert stage("trace_commit", lambda: ctx.trace_commit_device(d_trace.data_ptr(), N, w, 2 * w))[0] == 0
    assert stage("interpolate", ctx.interpolate) == 0
    stage("lincomb", lambda: [ctx.check(ctx.polys_lincomb(sc, idx)) for sc, idx in combos])
    assert stage("lde_commit", lambda: ctx.lde_commit(a.blowup, r2.next() % P or 3, 2 * w))[0] == 0
    assert stage("mix", lambda: ctx.mix(r2.next() % P)) == 0
    assert stage("eval_ext", lambda: ctx.eval_ext(np.array([r2.next() % P, r2.next() % P], dtype=np.uint64)))[0] == 0
    assert stage("fri_begin", lambda: ctx.fri_begin(a.blowup, rounds))[0] == 0
    for _ in range(1, rounds):
        assert stage("fri_deep", lambda: ctx.fri_deep([r2.next() % P, r2.next() % P]))[0] == 0
        assert stage("fri_fold_commit", lambda: ctx.fri_fold_commit([r2.next() % P, r2.next() % P]))[0] == 0
    assert stage("fri_query", lambda: ctx.fri_query([r2.next()], read=False))[0] == 0
el = time.perf_counter() - t0
import ctypes as C
buf = C.create_string_buffer(1 << 15)
ctx.check(ctx.L.ms_profile_begin(ctx.h))
r2 = SplitMix64(6)
ctx.trace_commit_device(d_trace.data_ptr(), N, w, 2 * w); ctx.interpolate()
[ctx.check(ctx.polys_lincomb(sc, idx)) for sc, idx in combos]
ctx.lde_commit(a.blowup, r2.next() % P or 3, 2 * w)
ctx.check(ctx.L.ms_profile_end(ctx.h, buf, C.c_size_t(len(buf))))
prof = json.loads(buf.value.decode())
prof.pop("shard", None)
variants = prof.pop("ntt_pass_variants")
print(json.dumps({"workload": f"wide AIR w={w} c={2 * w}, Goldilocks, 2^{a.log_rows} rows, blowup {a.blowup}", "proofs": a.proofs, "s_per_proof": el / a.proofs,
                  "proofs_per_s": a.proofs / el, "stage_ms_per_proof": {k: round(v / a.proofs * 1e3, 2) for k, v in times.items()},
                  "hbm_gib_allocated": round(torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30, 1),
                  "kernel_ms_trace_commit_to_lde_commit": {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]},
                  "ntt_variants": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "alg_GBps": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in variants.items()}}))
