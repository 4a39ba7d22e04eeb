#!/usr/bin/env python3
"""N proving threads, one context each, all proving the same trace over and over: every stage output of every proof is compared with a reference proof made by one
thread alone.  Finds data races that only show with several contexts in flight (the parity suite proves on one context at a time).
  python3 tools/multi_lane_check.py --lanes 8 --log-rows 18 --proofs 6"""
import argparse, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_stark_amd as ms
import parity_cases as pc
from common import fibonacci_trace_fast

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=8)
ap.add_argument("--log-rows", type=int, default=18)
ap.add_argument("--proofs", type=int, default=6)
ap.add_argument("--field", type=int, default=0)
ap.add_argument("--latency", action="store_true", help="contexts created with MS_FLAG_LATENCY (side streams)")
a = ap.parse_args()
trace = fibonacci_trace_fast(a.field, 1 << a.log_rows)
FLAGS = ms.FLAG_ZERO_DISPLAY_EMPTY | (ms.FLAG_LATENCY if a.latency else 0)
ref_ctx = ms.Context(a.field)
want = pc._digest_big(pc.drive(ref_ctx, a.field, trace, 8, 2, seed=13, read_big=False))
again = pc._digest_big(pc.drive(ref_ctx, a.field, trace, 8, 2, seed=13, read_big=False))
assert want == again, "the single-lane proof is not reproducible"
ref_ctx.close()
bad = []
lock = threading.Lock()


def lane(i):
    ctx = ms.Context(a.field, flags=FLAGS)
    for k in range(a.proofs):
        try:
            got = pc._digest_big(pc.drive(ctx, a.field, trace, 8, 2, seed=13, read_big=False))
        except AssertionError as e:
            with lock:
                bad.append((i, k, "stage failed: " + str(e)[:200]))
            return
        for (x, y) in zip(got, want):
            if x != y:
                with lock:
                    bad.append((i, k, "first differing output: " + str(x[0])))
                break
    ctx.close()


ts = [threading.Thread(target=lane, args=(i,)) for i in range(a.lanes)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print({"lanes": a.lanes, "latency_flag": a.latency, "log_rows": a.log_rows, "proofs_per_lane": a.proofs, "mismatches": sorted(bad)[:16], "ok": not bad})
sys.exit(1 if bad else 0)
