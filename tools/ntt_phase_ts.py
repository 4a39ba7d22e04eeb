#!/usr/bin/env python3
"""Where a tile's time goes inside the cooperative NTT pass kernels (GPU).

Builds a variant of the library with -DMS_NTT_TS (wave 0 of every workgroup stamps the shader clock at the phase boundaries of its first 16
work items; `MS_TS` of tools/ntt_instrumented.patch, applied to a copy of csrc/ for this build), runs the device-resident six-column coset LDE once and prints, per kernel mode, the mean clocks of every
phase of a work item as wave 0 sees them:

  load   item start -> tile in LDS (global loads issued, waited for and written to LDS; behind the virtual pass: hand-over + expansion)
  b      the barrier behind it (wave 0 waits for the slowest wave)
  sub0   first register sub-round (+ table builds behind the virtual pass)      b  barrier
  sub1   second sub-round                                                       b  barrier
  tail   last sub-round fused with the twiddle multiplication and the global stores (issue only: stores are not waited for)
  b      end-of-tile barrier

The instrumented build is a measurement aid only: it is written to tools/_build/ and never loaded by the product or the tests.
"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build", "libministark_ts.so")   # --out overrides


def instrumented_tree():
    """A copy of csrc/ with tools/ntt_instrumented.patch applied (the phase stamps MS_TS and the timing-only ablation branches MS_NTT_ABL_HOTLOAD / _NOSTORE live in the
    patch, not in the product source: VERDICT r3 #7).  Returns the directory that holds include/ and csrc/."""
    import shutil
    top = os.path.join(ROOT, "tools", "_build", "instr")
    shutil.rmtree(top, ignore_errors=True)
    os.makedirs(os.path.join(top, "pkg"))
    shutil.copytree(os.path.join(ROOT, "mini-stark_amd", "csrc"), os.path.join(top, "pkg", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(top, "include"))
    subprocess.check_call(["patch", "-s", "-p1", "-d", os.path.join(top, "pkg"), "-i", os.path.join(ROOT, "tools", "ntt_instrumented.patch")])
    return top


def build(extra, defines=("-DMS_NTT_TS",)):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(instrumented_tree(), "pkg", "csrc", "ministark.cpp")   # (includes ../../include/ministark.h relative to itself)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"] + list(defines) + extra + ["-x", "hip", src, "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--field", type=int, default=0)
    ap.add_argument("--define", action="append", default=[], help="extra -D for the variant build")
    ap.add_argument("--out", default=None, help="file name of the variant library under tools/_build/")
    ap.add_argument("--brief", action="store_true")
    a = ap.parse_args()
    global OUT
    if a.out:
        OUT = os.path.join(ROOT, "tools", "_build", a.out)
    if a.build_only or not os.path.exists(OUT):
        build(["-D" + d for d in a.define])
        if a.build_only:
            return
    import numpy as np
    import torch
    import mini_stark_amd as ms
    from mini_stark_amd.stark import fibonacci_air
    dev = torch.device("cuda", 0)
    os.environ["MS_LDE_LINEAR"] = "0"
    ctx = ms.Context(a.field, lib_path=OUT)
    TILES, SLOTS = 16, 12
    buf = torch.zeros(3 * 1024 * TILES * SLOTS, dtype=torch.int64, device=dev)
    ctx.L.ms_debug_ntt_ts.argtypes = [C.c_void_p]
    assert ctx.L.ms_debug_ntt_ts(C.c_void_p(buf.data_ptr())) == 0
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    tt = fibonacci_air(ctx, (1 << a.log_rows) - 1)
    with torch.cuda.stream(stream):
        ctx.check(ctx.trace_commit(tt.data, 6)[0])
        ctx.check(ctx.interpolate())
        for sc, idx in tt.transitions:
            ctx.check(ctx.polys_lincomb(sc, idx))
        for _ in range(3):
            ctx.check(ctx.bench_lde(8, 12345))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            ctx.check(ctx.bench_lde(8, 12345))
        e1.record(stream)
        torch.cuda.synchronize()
        lde_ms = e0.elapsed_time(e1) / 20
        buf.zero_()
        torch.cuda.synchronize()
        ctx.check(ctx.bench_lde(8, 12345))
        torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(3, 1024, TILES, SLOTS).astype(np.float64)
    names = ["load", "expand", "b", "sub0", "b", "sub1", "b", "tail", "b"]   # "expand": behind the virtual pass only (0 in the other modes)
    bounds = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9)]
    res = {"lde_ms_instrumented": round(lde_ms, 4), "log_rows": a.log_rows, "field": a.field}
    for pm, label in ((2, "first pass behind the virtual pass"), (1, "later pass")):
        x = t[pm]
        valid = (x[:, :, 0] > 0) & (x[:, :, 9] > 0)
        if not valid.any():
            continue
        wg = valid.any(axis=1)
        items = int(valid.sum())
        spans = []                                                # the shader clocks of the XCDs are not synchronised: one span per XCD (workgroup index & 7)
        for xcd in range(8):
            m = valid[xcd::8]
            if m.any():
                spans.append(float(x[xcd::8, :, 9][m].max() - x[xcd::8, :, 0][m].min()))
        d = {"workgroups": int(wg.sum()), "items_stamped": items, "kernel_span_ticks_mean_over_xcds": float(np.mean(spans))}
        tot = 0.0
        ph = []
        for (n, (lo, hi)) in zip(names, bounds):
            v = (x[:, :, hi] - x[:, :, lo])[valid]
            ph.append((n, float(v.mean()), float(np.percentile(v, 10)), float(np.percentile(v, 90))))
            tot += v.mean()
        item_ticks = (x[:, :, 9] - x[:, :, 0])[valid]
        d["item_ticks_mean"] = float(item_ticks.mean())
        # steady-state items only (not the first of a workgroup: its loads start cold)
        if x.shape[1] > 1:
            v2 = valid.copy(); v2[:, 0] = False
            if v2.any():
                d["item_ticks_mean_steady"] = float((x[:, :, 9] - x[:, :, 0])[v2].mean())
                d["gap_between_items_mean"] = float((x[:, 1:, 0] - x[:, :-1, 9])[valid[:, 1:] & valid[:, :-1]].mean())
        d["phases_mean_p10_p90_ticks"] = [{"phase": n, "mean": round(m, 1), "p10": round(p10, 1), "p90": round(p90, 1), "share": round(m / tot, 3)} for (n, m, p10, p90) in ph]
        res[label] = d
    if a.brief:
        print(os.path.basename(OUT), "| six-column LDE", res["lde_ms_instrumented"], "ms")
        for k, d in res.items():
            if isinstance(d, dict):
                print(os.path.basename(OUT), "|", k, "| item", round(d["item_ticks_mean"]), "span", round(d["kernel_span_ticks_mean_over_xcds"]), "|",
                      " ".join(f"{q['phase']}={q['mean']:.0f}" for q in d["phases_mean_p10_p90_ticks"]))
    else:
        print(json.dumps(res, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
