#!/usr/bin/env python3
"""Wall time of every stage call of ONE proof at a time (latency view): python tools/stage_times.py [--log-rows 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mini_stark_amd as ms
from mini_stark_amd.stark import fibonacci_air
from mini_stark_amd.synthetic import SplitMix64

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, default=20)
ap.add_argument("--field", type=int, default=0)
ap.add_argument("--proofs", type=int, default=5)
ap.add_argument("--lib", default=None, help="another build of libministark.so (experiments)")
a = ap.parse_args()
ctx = ms.Context(a.field, lib_path=a.lib)
P = 2**64 - 2**32 + 1 if a.field == 0 else 2013265921
N = 1 << a.log_rows
tt = fibonacci_air(ctx, N - 1)
d_trace = torch.from_numpy(tt.data.view(np.int64)).cuda()
rounds = a.log_rows + 3
e = ctx.e
T = {}; per_round = [[0.0, 0.0] for _ in range(rounds)]


def st(name, fn):
    t = time.perf_counter(); r = fn(); dt = time.perf_counter() - t
    T[name] = T.get(name, 0.0) + dt
    return r, dt


for it in range(a.proofs + 1):
    if it == 1:
        T.clear(); per_round = [[0.0, 0.0] for _ in range(rounds)]; t0 = time.perf_counter()
    rng = SplitMix64(9)
    st("trace_commit", lambda: ctx.trace_commit_device(d_trace.data_ptr(), N, 3, 6))
    st("interpolate", ctx.interpolate)
    st("lincomb", lambda: [ctx.polys_lincomb(sc, idx) for sc, idx in tt.transitions])
    st("lde_commit", lambda: ctx.lde_commit(8, rng.next() % P or 3, 6))
    st("mix", lambda: ctx.mix(rng.next() % P))
    st("eval_ext", lambda: ctx.eval_ext(np.array([rng.next() % P for _ in range(e)], dtype=np.uint64)))
    st("fri_begin", lambda: ctx.fri_begin(8, rounds))
    for i in range(1, rounds):
        _, d1 = st("fri_deep", lambda: ctx.fri_deep([rng.next() % P for _ in range(e)]))
        _, d2 = st("fri_fold_commit", lambda: ctx.fri_fold_commit([rng.next() % P for _ in range(e)]))
        per_round[i][0] += d1; per_round[i][1] += d2
    st("fri_query", lambda: ctx.fri_query([rng.next(), rng.next()], read=False))
el = (time.perf_counter() - t0) / a.proofs
print(json.dumps({"ms_per_proof": round(el * 1e3, 3), "stage_ms": {k: round(v / a.proofs * 1e3, 3) for k, v in T.items()},
                  "round_us_deep_fold": [[round(x[0] / a.proofs * 1e6), round(x[1] / a.proofs * 1e6)] for x in per_round[1:]]}))
