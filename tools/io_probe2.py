#!/usr/bin/env python3
"""I/O-inclusive rate of the benchmark proof by read-back mode (8 lanes): resident / async read-back / blocking / into (kernels write the page-locked slot)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class G:
    world, rank = 1, 0
    def barrier(self): torch.cuda.synchronize()
    def max_over_ranks(self, s): return s
dev = torch.device("cuda", 0)
res = {"pinned_flags": os.environ.get("MS_PINNED_FLAGS", "default")}
for name, io, mode in (("resident", False, None), ("into", True, "into"), ("async", True, "async"), ("blocking", True, True)):
    ln = bench.Lanes(0, 20, 8, 8, 0, dev, io=io, io_mode=mode)
    el = ln.timed(G(), 10, 2)
    res[name] = round(10 * 8 / el, 1)
    ln.close()
print(json.dumps(res))
