#!/usr/bin/env python3
"""NTT micro-benchmark (GPU): times the device-resident coset-LDE stage (ms_bench_lde:
scale + NTT passes, no hashing, no host copies) with HIP events on the launching stream.
Algorithmic bytes = c*(N+L)*sizeof(T)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mini_stark_amd as ms
from mini_stark_amd.stark import fibonacci_air

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, nargs="+", default=[20])
ap.add_argument("--blowup", type=int, default=8)
ap.add_argument("--field", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="")
ap.add_argument("--passes", action="store_true", help="also print the per-kernel times of one LDE (the library's HIP-event profiler, ms_profile_begin / _end)")
ap.add_argument("--linear", type=int, default=0, help="1: let the LDE stage use the linear-provenance shortcut (3 NTTs + 3 lincombs); 0: six NTTs (kernel measurement)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
os.environ["MS_LDE_LINEAR"] = str(a.linear)
for lr in a.log_rows:
    ctx = ms.Context(a.field, lib_path=a.lib)
    ctx.set_stream(stream.cuda_stream)
    tt = fibonacci_air(ctx, (1 << lr) - 1)
    with torch.cuda.stream(stream):
        ctx.check(ctx.trace_commit(tt.data, 6)[0])
        ctx.check(ctx.interpolate())
        for sc, idx in tt.transitions:
            ctx.check(ctx.polys_lincomb(sc, idx))
        for _ in range(3):
            ctx.check(ctx.bench_lde(a.blowup, 12345))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.reps):
            ctx.check(ctx.bench_lde(a.blowup, 12345))
        e1.record(stream)
    torch.cuda.synchronize()
    ms_per = e0.elapsed_time(e1) / a.reps
    N, L, c = 1 << lr, (1 << lr) * a.blowup, 6
    s = 8 if a.field == 0 else 4
    alg = c * (N + L) * s
    print(json.dumps({"tag": a.tag, "linear_shortcut": a.linear, "log_rows": lr, "blowup": a.blowup, "field": a.field, "lde_ms": round(ms_per, 4), "alg_GBps": round(alg / ms_per / 1e6, 1),
                      "frac_of_8TBps": round(alg / ms_per / 1e6 / 8000, 4)}))
    if a.passes:
        import ctypes as C
        buf = C.create_string_buffer(1 << 14)
        with torch.cuda.stream(stream):
            ctx.check(ctx.L.ms_profile_begin(ctx.h))
            for _ in range(5):
                ctx.check(ctx.bench_lde(a.blowup, 12345))
            ctx.check(ctx.L.ms_profile_end(ctx.h, buf, C.c_size_t(len(buf))))
        prof = json.loads(buf.value.decode())
        print(json.dumps({"tag": a.tag, "log_rows": lr, "field": a.field,
                          "pass_us": {k.replace("msntt::PassKernel2", "P2"): round(v["ms"] / max(1, v["launches"]) * 1e3, 1) for k, v in prof["ntt_pass_variants"].items()}}))
    ctx.close()
