#!/usr/bin/env python3
"""What ONE rank of a W-rank sharded proof computes, measured on one GPU: the context is rank W-1 of W with a STUB exchange (every collective returns the rank's own
payload in every peer's place - no peers exist), so the kernels run on operands of the right SIZES but wrong VALUES; the proof itself is garbage (the query phase
ends with "leaf not found", after all its work).  Gives, per world size: wall time of the rank's proof, kernel time partitioned / replicated (HIP events per launch,
ms_profile), collective calls - next to the unsharded proof on the same GPU.  Strong-scaling bound = unsharded time / (rank time + link time of the exchanges).
  python3 tools/shard_rank_probe.py --log-rows 24 --worlds 2 4 8"""
import argparse, ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mini_stark_amd as ms
from mini_stark_amd.stark import StarkConfig, fibonacci_air
from mini_stark_amd.host import HostStark

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, default=20)
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dist", type=int, nargs="+", default=[1, 0], help="MS_SHARD_DIST values to compare (1: coefficient-domain work partitioned, 0: replicated as in r03)")
ap.add_argument("--python-stub", action="store_true")
args = ap.parse_args()
if not args.python_stub:
    os.environ["MS_SHARD_STUB"] = "1"
dev = torch.device("cuda", 0)
N = 1 << args.log_rows
blowup = 8


class Stub:
    """A world without peers.  Default (r05): the library's own stub (MS_SHARD_STUB=1: stream-ordered device copies, no host synchronisation - what the in-library RCCL
    path costs a rank on the host side); --python-stub: the r04 form, a Python callback with torch copies and a device synchronisation per exchange."""
    def __init__(self, ctx, rank, world, cap):
        self.ctx, self.W = ctx, world
        self.send = torch.zeros(cap, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(cap, dtype=torch.uint8, device=dev)
        self.calls = {}
        ctx.set_shard(rank, world, self.send.data_ptr(), self.recv.data_ptr(), cap, self.cb if args.python_stub else None)

    def cb(self, op, n):
        self.calls[op] = self.calls.get(op, 0) + 1
        W = self.W
        if op == 0:
            self.recv[: n * W].copy_(self.send[: n * W])
        elif op == 1:
            self.recv[: n * W].view(W, n).copy_(self.send[:n].unsqueeze(0).expand(W, n))
        elif op == 4:
            off, stride = C.c_size_t(0), C.c_size_t(0)
            self.ctx.L.ms_shard_slice_layout(self.ctx.h, C.byref(off), C.byref(stride))
            for p in range(W):
                a = off.value + p * stride.value
                self.recv[a: a + n].copy_(self.send[a: a + n])
        torch.cuda.synchronize()
        return 0


def fingerprint(hs):
    """Everything a proof outputs, compactly: transcript, both trace commitments, DEEP values, FRI roots, SHA-256 of the FRI blob (read back from HBM)."""
    p = hs.last_proof(read_fri_proof=True)
    return {"arthur": hashlib.sha256(p.arthur).hexdigest(), "trace_root": p.trace_commit.hex(), "lde_root": p.constrain_trace_commit.hex(),
            "deep_values": hashlib.sha256(p.constrain_queries.tobytes() + p.validity_queries.tobytes()).hexdigest(),
            "fri_roots": hashlib.sha256(b"".join(p.fri_roots)).hexdigest(), "fri_blob": hashlib.sha256(p.fri_proof.blob).hexdigest(), "fri_blob_bytes": len(p.fri_proof.blob)}


UNSHARDED = {}


def one(world, dist):
    os.environ["MS_SHARD_DIST"] = str(dist)
    ctx = ms.Context(0)
    stub = None
    if world > 1:
        stub = Stub(ctx, world - 1, world, 32 * N * blowup // world + (4 << 20))
    tt = fibonacci_air(ctx, N - 1)
    hs = HostStark(ctx, 20, blowup, N - 1, tt.constrain_number())
    d_trace = torch.from_numpy(tt.data.view(np.int64)).to(dev)
    ok = (0,) if world == 1 else (0, ms.ERR_LEAF_NOT_FOUND)
    def prove():
        rc = hs.prove_raw(tt, trace_device_ptr=d_trace.data_ptr(), read_fri_proof=False)
        assert rc in ok, (rc, ctx.last_error())
    prove()
    torch.cuda.synchronize()
    if world == 1 and not UNSHARDED:   # the reference the one-rank sharded legs below are checked against (the stub worlds compute on wrong values by construction)
        ctx.check(hs.prove_raw(tt, trace_device_ptr=d_trace.data_ptr(), read_fri_proof=True))
        UNSHARDED.update(fingerprint(hs))
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter(); prove(); ts.append((time.perf_counter() - t0) * 1e3)
    buf = C.create_string_buffer(1 << 15)
    ctx.check(ctx.L.ms_profile_begin(ctx.h)); prove(); ctx.check(ctx.L.ms_profile_end(ctx.h, buf, C.c_size_t(len(buf))))
    prof = json.loads(buf.value.decode())
    sh = prof.pop("shard", {})
    prof.pop("ntt_pass_variants", None)
    res = {"world": world, "shard_dist": dist, "log_rows": args.log_rows, "ms_wall_min": round(min(ts), 3), "ms_wall_all": [round(t, 3) for t in ts],
           "kernel_ms_total": round(sum(v["ms"] for v in prof.values()), 3), "partitioned_ms": sh.get("partitioned_ms"), "replicated_ms": sh.get("replicated_ms"),
           "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}, "replicated_by_kernel": sh.get("replicated_by_kernel"),
           "launches": sum(v["launches"] for v in prof.values()), "launches_by_kernel": {k: v["launches"] for k, v in prof.items() if v["launches"]}, "collective_calls": (stub.calls if args.python_stub else {i: c for i, c in enumerate(ctx.shard_stats()[:4])}) if stub else {}}
    if stub:
        ctx.set_shard(0, 1, 0, 0, 0, None)
    ctx.close()
    return res


base = one(1, 1)
print(json.dumps(base), flush=True)
if os.environ.get("MS_PROBE_W1_RCCL") == "1":
    # the sharded prover on a ONE-rank RCCL communicator (MS_SHARD_WORLD1): same work as the unsharded proof plus everything sharding adds on a rank - coset folds, digest
    # interleaves, the extra launches of the distributed scans - and ~80 RCCL calls per proof (to itself): what the sharded STRUCTURE costs before any link is involved
    from mini_stark_amd.dist import LocalShard
    os.environ["MS_SHARD_WORLD1"] = "1"
    for slices in ("1", "4"):
        os.environ["MS_SHARD_SLICES"] = slices
        ctx = ms.Context(0)
        sh = LocalShard(ctx, 32 * N * blowup + (4 << 20), rccl=True)
        ctx.shard_proof_on_root(True)
        tt = fibonacci_air(ctx, N - 1)
        hs = HostStark(ctx, 20, blowup, N - 1, tt.constrain_number())
        d_trace = torch.from_numpy(tt.data.view(np.int64)).to(dev)
        def prove():
            ctx.check(hs.prove_raw(tt, trace_device_ptr=d_trace.data_ptr(), read_fri_proof=False))
        prove(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); prove(); ts.append((time.perf_counter() - t0) * 1e3)
        st0 = ctx.shard_stats()
        prove()
        st1 = ctx.shard_stats()
        # this leg is no longer timing-only - the proof the sharded code paths produce must be the unsharded one, bit for bit
        ctx.check(hs.prove_raw(tt, trace_device_ptr=d_trace.data_ptr(), read_fri_proof=True))
        fp = fingerprint(hs)
        assert fp == UNSHARDED, {"sharded_one_rank": fp, "unsharded": UNSHARDED}
        print(json.dumps({"world": 1, "rccl_one_rank": True, "matches_unsharded": True, "fri_blob_bytes": fp["fri_blob_bytes"], "digest_exchange_slices": int(slices), "log_rows": args.log_rows, "ms_wall_min": round(min(ts), 3), "ms_wall_all": [round(t, 3) for t in ts],
                          "overhead_vs_unsharded_ms": round(min(ts) - base["ms_wall_min"], 3), "collective_calls_per_proof": [st1[i] - st0[i] for i in range(4)]}), flush=True)
        sh.close(); ctx.close()
    del os.environ["MS_SHARD_WORLD1"], os.environ["MS_SHARD_SLICES"]
for w in args.worlds:
    for d in args.dist:
        r = one(w, d)
        r["speedup_bound_no_link_time"] = round(base["ms_wall_min"] / r["ms_wall_min"], 3)
        print(json.dumps(r), flush=True)
