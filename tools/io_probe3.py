#!/usr/bin/env python3
"""I/O-inclusive rate of the benchmark proof by read-back path, alternating the variants inside ONE process (8 lanes each):
resident (no I/O) / asynchronous read-back by the HIP runtime's copy / by an SDMA engine through the HSA runtime / blocking with either.
MS_READBACK is read at ms_create, so every variant owns its lanes; the timed runs alternate over `--rounds` passes in a fixed order.
  python3 tools/io_probe3.py [--steps 20] [--rounds 3] [--only NAME]      (one JSON line per pass + a summary line)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--only", default=None)
ap.add_argument("--inflight", type=int, default=8)
args = ap.parse_args()


class G:
    world, rank = 1, 0
    def barrier(self): torch.cuda.synchronize()
    def max_over_ranks(self, s): return s


# name, read-back on, read-back mode, MS_READBACK, upload on, MS_UPLOAD
VARIANTS = [("resident", False, None, "sdma", False, "hip"), ("async_hip", True, "async", "hip", True, "hip"), ("async_sdma", True, "async", "sdma-async", True, "hip"),
            ("blocking_hip", True, True, "hip", True, "hip"), ("blocking_sdma", True, True, "sdma", True, "hip"),
            ("upload_only_hip", False, None, "sdma", True, "hip"), ("upload_only_sdma", False, None, "sdma", True, "sdma"),
            ("readback_only_sdma", True, "async", "sdma", False, "hip"), ("async_sdma_upload_sdma", True, "async", "sdma", True, "sdma"),
            ("blocking_sdma_upload_sdma", True, True, "sdma", True, "sdma")]
if args.only:
    VARIANTS = [v for v in VARIANTS if v[0] in args.only.split(",")]
dev = torch.device("cuda", 0)
lanes = {}
for name, io, mode, rb, up, upm in VARIANTS:
    os.environ["MS_READBACK"] = rb
    os.environ["MS_UPLOAD"] = upm
    lanes[name] = bench.Lanes(0, 20, 8, args.inflight, 0, dev, io=io, io_mode=mode, upload=up)
allres = {n: [] for n, *_ in VARIANTS}
for r in range(args.rounds):
    res = {"pass": r}
    for name, io, mode, rb, up, upm in VARIANTS:
        ln = lanes[name]
        el = ln.timed(G(), args.steps, args.warmup)
        v = round(args.steps * args.inflight / el, 1)
        res[name] = v
        allres[name].append(v)
        if io:
            res[name + "_engine"] = [c.L.ms_io_engine(c.h) for c in ln.ctxs][:2]
            res[name + "_sampled"] = all(len(sm) == 1 and 0 not in sm for sm in ln.samples)
    print(json.dumps(res), flush=True)
summ = {n: {"median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v)} for n, v in allres.items() if v}
if "resident" in summ:
    for n in summ:
        summ[n]["vs_resident"] = round(summ[n]["median"] / summ["resident"]["median"], 3)
print(json.dumps({"summary": summ, "steps": args.steps, "inflight": args.inflight}), flush=True)
