"""Splits the I/O-inclusive rate: trace upload only / proof read-back only / both (8 lanes, 2^20 rows)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mini_stark_amd.dist import Group
grp = Group("nccl")
dev = grp.device
for name, io_in, io_out in (("none", 0, 0), ("upload only", 1, 0), ("read-back only (blocking)", 0, 1), ("read-back only (async)", 0, 2), ("both, blocking", 1, 1), ("both, async read-back", 1, 2)):
    ln = bench.Lanes(0, 20, 8, 8, 0, dev, io=True)
    ln.d_traces = [torch.from_numpy(t.data.view(np.int64)).to(dev) for t in ln.tts]
    torch.cuda.synchronize()
    def _prove_n(i, n, ln=ln, io_in=io_in, io_out=io_out):
        ptr = None if io_in else ln.d_traces[i].data_ptr()
        for _ in range(n):
            ln.ctxs[i].check(ln.starks[i].prove_raw(ln.tts[i], trace_device_ptr=ptr, read_fri_proof=("async" if io_out == 2 else bool(io_out))))
        if io_out == 2: ln.ctxs[i].check(ln.starks[i].wait_proof())
        ln.last[i] = ln.starks[i].last_proof(read_fri_proof=False)
    ln._prove_n = _prove_n
    el = ln.timed(grp, 10, 2)
    print(json.dumps({"io": name, "proofs_per_s": 10 * 8 / el}), flush=True)
    ln.close()
