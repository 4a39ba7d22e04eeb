#!/usr/bin/env python3
"""One-off bit-exactness runs at sizes beyond the test-suite's budget: GPU proof vs the CPU oracle (OpenMP over its independent loops).
   python tools/fullsize_parity.py FIELD LOG_ROWS [THREADS [wide]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_stark_amd as ms
import parity_cases as pc
from oracle import oracle as orc
field, log_n = int(sys.argv[1]), int(sys.argv[2])
orc.set_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 32)
cache = {}
def mk(f, fresh=False):
    if f not in cache: cache[f] = ms.Context(f)
    return cache[f]
t = time.time()
if len(sys.argv) > 4 and sys.argv[4] == "wide":   # BASELINE configs[4] shape: 64 trace columns, c = 128
    pc.case_prove_wide(mk, field, log_n=log_n, w=64)
    print(f"wide AIR (w=64, c=128), field {field}, 2^{log_n} rows, blowup 8: commitments, DEEP values, FRI rounds and FRI proof bit-exact vs the oracle ({time.time() - t:.0f} s)", flush=True)
    sys.exit(0)
pc.case_prove(mk, field, log_n, 8, nq_fri=0, read_big=False)
print(f"field {field} 2^{log_n} rows, blowup 8: every commitment, DEEP value, FRI round and the FRI proof bit-exact vs the oracle ({time.time() - t:.0f} s)", flush=True)
