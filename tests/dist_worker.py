"""Worker for tests/test_dist_gloo.py: one rank of the N>1 bench path on CPU (gloo), proving on the
kernel-emulation library."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mini_stark_amd as ms  # noqa: E402
from mini_stark_amd.dist import Group  # noqa: E402
from mini_stark_amd.stark import Stark, StarkConfig, fibonacci_air  # noqa: E402

grp = Group("gloo")
ctx = ms.Context(ms.GOLDILOCKS, lib_path=os.path.join(ROOT, "tests", "emu", "libministark_emu.so"))
tt = fibonacci_air(ctx, 63, secret_b=2 + grp.rank)
cfg = StarkConfig(ctx, 20, 8, 63, tt.constrain_number())
grp.barrier()
t0 = time.perf_counter()
proof = Stark(cfg).prove(tt)
grp.barrier()
elapsed = grp.max_over_ranks(time.perf_counter() - t0)
roots = grp.all_gather_bytes(proof.fri_roots[-1] + proof.trace_commit)
if grp.rank == 0:
    print(json.dumps({"world": grp.world, "elapsed": elapsed, "roots": [r.hex() for r in roots]}), flush=True)
grp.close()
