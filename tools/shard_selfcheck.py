#!/usr/bin/env python3
"""Diagnostic driver of tests/parity_cases.py:case_sharded_one_rank_matches_unsharded: the sharded code paths on a one-rank world against the unsharded proof of the
same library, for a list of sizes and exchange back ends; prints which stage output differs first.
  python3 tools/shard_selfcheck.py --log-rows 22 23 24 --backends rccl callback [--wide 64] [--slices 1 4]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mini_stark_amd as ms
import parity_cases as pc

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, nargs="+", default=[22])
ap.add_argument("--backends", nargs="+", default=["rccl", "callback"])
ap.add_argument("--slices", nargs="+", default=["1"])
ap.add_argument("--wide", type=int, default=0)
ap.add_argument("--root-only", type=int, default=0)
args = ap.parse_args()


def set_env(k, v):
    if v is None:
        os.environ.pop(k, None)
    else:
        os.environ[k] = v


for lr in args.log_rows:
    for be in args.backends:
        for sl in args.slices:
            t0 = time.time()
            try:
                res = pc.case_sharded_one_rank_matches_unsharded(lambda f: ms.Context(f), 0, lr, [(be == "rccl", {"MS_SHARD_SLICES": sl}, bool(args.root_only))], set_env,
                                                                 wide_w=args.wide, device=torch.device("cuda", 0), seed=5 if args.wide else 31)
                print(json.dumps({"log_rows": lr, "backend": be, "slices": sl, "wide": args.wide, "ok": True, "stats": [int(v) for v in res[0][0]], "dist_rounds": res[0][1], "s": round(time.time() - t0, 1)}), flush=True)
            except AssertionError as e:
                print(json.dumps({"log_rows": lr, "backend": be, "slices": sl, "wide": args.wide, "ok": False, "first_difference": str(e)[:400], "s": round(time.time() - t0, 1)}), flush=True)
                for k in ("MS_SHARD_WORLD1", "MS_SHARD_SLICES"):
                    os.environ.pop(k, None)
