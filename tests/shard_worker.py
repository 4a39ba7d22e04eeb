"""Worker for tests/test_shard_gloo.py (and the GPU rehearsal in tests/test_gpu_parity.py): one rank of ONE proof
sharded over the ranks (ms_set_shard), on the kernel-emulation library (or, with "gpu", the HIP library) with gloo; every rank checks all stage outputs and the FRI proof bit-for-bit against
the single-process oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_stark_amd as ms  # noqa: E402
from mini_stark_amd.dist import Group, ShardExchange  # noqa: E402
import parity_cases as pc  # noqa: E402
from common import fibonacci_trace_fast  # noqa: E402
from oracle import oracle as orc  # noqa: E402

field, log_n, blowup = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
os.environ["MS_SHARD_MIN_LEAVES"] = sys.argv[4] if len(sys.argv) > 4 else "16"
mode = sys.argv[5] if len(sys.argv) > 5 else ""
modes = set(mode.split(","))           # several at once: "gpu,root-only"
on_gpu = bool(modes & {"gpu", "gpu-self"})   # the real HIP library, all ranks on GPU 0, payloads staged through host tensors for gloo
self_check = "gpu-self" in modes       # sizes beyond the oracle's reach: compare with an UNSHARDED proof of the same library instead
grp = Group("gloo")
N = 1 << log_n
cap = 32 * N * blowup // grp.world + (1 << 20)  # leaf digests of the largest commitment / world + query-phase paths
if on_gpu:
    import torch
    ctx = ms.Context(field)
    xchg = ShardExchange(grp, ctx, cap, staged=True, buffer_device=torch.device("cuda", 0))
else:
    ctx = ms.Context(field, lib_path=os.environ.get("MS_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libministark_emu.so"))  # MS_EMU_LIB: sanitizer builds
    xchg = ShardExchange(grp, ctx, cap)
trace = fibonacci_trace_fast(field, N)
if "low-degree" in modes:   # trace columns of degree N/2 + 2: the validity polynomial fills only half of round 0's coefficient ranges - ranks above world / 2 hold (almost) nothing, a middle rank holds the ragged top
    import numpy as np
    from common import MODULUS
    cols = []
    for c in range(3):
        coef = np.zeros(N, dtype=np.uint64)
        coef[: N // 2 + 3] = pc.rand_field(field, (N // 2 + 3,), seed=40 + c)
        cols.append(orc.ntt(field, coef))
    trace = np.ascontiguousarray(np.stack(cols, axis=1))
if log_n >= 16:  # full-size comparisons: OpenMP over the oracle's independent loops (every rank runs its own oracle)
    orc.set_threads(max(1, min(8, len(os.sched_getaffinity(0)) // grp.world)))
root_only = "root-only" in modes        # ms_shard_proof_on_root: only rank 0 ends up with the FRI proof
base_z = (1, 2) if "base-z" in modes else ()   # DEEP points in the base field in the first rounds (distributed polynomials: the transform fallback gathers them)
if root_only:
    ctx.shard_proof_on_root(True)
got = pc.drive(ctx, field, trace, blowup, 2, seed=11, read_big=False, base_field_z_rounds=base_z)
if self_check:
    want = pc.drive(ms.Context(field), field, trace, blowup, 2, seed=11, read_big=False, base_field_z_rounds=base_z)
else:
    want = pc.drive(orc.Session(field), field, trace, blowup, 2, seed=11, read_big=False, base_field_z_rounds=base_z)
assert len(got) == len(want)
for (ka, va), (kb, vb) in zip(got, want):
    if root_only and ka == "fri_proof" and grp.rank != 0:
        assert va == b"", f"rank {grp.rank} holds a proof although it is assembled on rank 0 only"
        continue
    assert ka == kb and va == vb, f"rank {grp.rank}: stage output {ka} differs from the oracle"
# the distributed parts are not readable in shard mode
assert ctx.L.ms_lde_read(ctx.h, None) != 0
grp.barrier()
if grp.rank == 0:
    import ctypes as C
    dr = sum(1 for i in range(64) if ctx.L.ms_shard_round_is_distributed(ctx.h, C.c_int(i)) == 1)
    print(json.dumps({"world": grp.world, "calls": xchg.calls, "bytes": xchg.bytes, "stages": len(got), "slices": xchg.slices, "dist_rounds": dr, "root_only": root_only}), flush=True)
xchg.close()
grp.close()
