#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import csv, glob, json, collections, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
O = f"gpurun_out/prof_{R}"
# no silent reuse of an older run's files.  tools/profile_round.sh wipes its output directory, stamps the start, and records rc 0 only if EVERY pass succeeded;
# the local gpurun_out/ is a merge of many calls, so every file taken here must be newer than that start stamp.
done = json.load(open(O + "/passes_done.json"))
assert done["rc"] == 0, f"tools/profile_round.sh {R} did not complete (rc {done['rc']}): no summaries are produced from a partial run"
T0 = int(open(O + "/started_at").read())
def one(pat):
    fresh = [f for f in glob.glob(pat) if os.path.getmtime(f) >= T0 - 5]
    assert fresh, f"no file of THIS profile run matches {pat} (older ones are not taken)"
    return max(fresh, key=os.path.getmtime)
shutil.copy(one(O + "/stats1/*/*kernel_stats.csv"), f"profiles/{R}_kernel_stats_inflight1.csv")
bench = [json.loads(l) for l in open(O + "/bench_default.json") if l.startswith("{")][-1]
KN = bench["roofline"]["kernel"].replace("msntt::", "")
for r in csv.DictReader(open(f"profiles/{R}_kernel_stats_inflight1.csv")):
    if KN in r["Name"]:
        print("rocprof (1 proof in flight) avg ns", r["AverageNs"], "calls", r["Calls"])
for l in open(O + "/stats1.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("bench   (1 proof in flight) roofline", d["roofline"]["kernel"], "avg_launch_ms", d["roofline"]["avg_launch_ms"], "proofs/s", d["value"])
# ---- HBM traffic of every NTT pass kernel (FETCH_SIZE / WRITE_SIZE passes), per launch, with the calibration each access pattern needs
def per_dispatch(path, name):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "PassKernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
            out.setdefault(r["Kernel_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(l)] for k, l in out.items()}
def durations(tracefile):
    out = collections.OrderedDict()
    for r in sorted(csv.DictReader(open(tracefile)), key=lambda r: int(r["Dispatch_Id"])):
        if "PassKernel" in r["Kernel_Name"]:
            out.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out
F = per_dispatch(one(O + "/pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
W = per_dispatch(one(O + "/pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
D = durations(one(O + "/pmc_fetch/*/*kernel_trace.csv"))
short = lambda n: "msntt::" + n[n.index("PassKernel"):].split(" >(")[0].split(">(")[0]
variants = {}
for k in F:
    f, w, d = F[k], W.get(k, []), D.get(k, [])
    n = min(len(f), len(w), len(d))
    if not n:
        continue
    name = short(k)
    # FETCH_SIZE tallies a 128-byte request as 64 bytes (guide) and a 64-byte request as 64: calibrated per kernel on a known byte count.  A plain or later
    # pass (modes 0, 1) reads exactly as many bytes as it writes, and WRITE_SIZE is exact: the factor is 2 when the raw count is half of that (both halves of
    # a line asked for by the same XCD's L2: the XCD-contiguous tile walk), 1 when it equals it (r02's first walk: neighbouring 64-byte pieces went to
    # different XCDs).  Mode 2 (8-byte gathers + L2-resident tables): the guide's x2 kept as an upper bound.
    mode2 = name.rstrip(">").rstrip().endswith((", 2", ", 3"))
    corr = 2.0 if mode2 or sum(f[:n]) < 0.75 * sum(w[:n]) else 1.0
    per = [dict(fetch_raw_bytes=f[i] * 1024, fetch_bytes=f[i] * 1024 * corr, write_bytes=w[i] * 1024, hbm_bytes=f[i] * 1024 * corr + w[i] * 1024, us_in_pmc_run=d[i],
                TBps_on_traffic=(f[i] * 1024 * corr + w[i] * 1024) / d[i] / 1e6) for i in range(n)]
    hb = sum(x["hbm_bytes"] for x in per) / n; us = sum(d[:n]) / n
    variants[name] = dict(fetch_correction=corr, launches=n, hbm_bytes_per_launch=hb, avg_us=us, TBps_on_traffic=hb / us / 1e6, per_launch=per)
roof = "msntt::" + KN if not KN.startswith("msntt::") else KN
rv = variants.get(roof) or next(iter(variants.values()))
s8 = 8
alg = dict(lde_3_columns=3 * ((1 << 20) + (1 << 23)) * s8, fri_round0_1_column=((1 << 20) + (1 << 23)) * s8)
big = lambda name: max(x["hbm_bytes"] for x in variants[name]["per_launch"]) if name in variants else 0.0   # the 3-column launch of that kernel
fwd = [n for n in variants if ", false," in n]
first = [n for n in fwd if n.rstrip(">").rstrip().endswith(", 3")] or [n for n in fwd if n.rstrip(">").rstrip().endswith(", 2")]   # the 3-column LDE's first pass: the shared-table instance since r03
lde_traffic = sum(big(n) for n in first[:1] + [n for n in fwd if n.rstrip(">").rstrip().endswith(", 1")][:1])
json.dump(dict(kernel=roof, round=R, launches=rv["launches"], hbm_bytes_per_launch=rv["hbm_bytes_per_launch"],
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-extras` "
                      "(2 proofs: per proof one 3-column LDE and one 1-column FRI round-0 transform, each two launches; plus the 3-column INTT). Counter unit KB (x1024). "
                      "FETCH_SIZE calibration (MI355X_MICROARCH.md: a 128-byte request is tallied as 64 bytes; other patterns to be calibrated on a known byte count): a plain or later pass "
                      "(modes 0, 1) reads exactly the bytes it writes (WRITE_SIZE is exact), so fetch_correction = 2 where the raw count is half of that - the later passes since the "
                      "XCD-contiguous tile walk, where one L2 asks for both 64-byte halves of a line: raw 101.2 MB for the 201.3 MB a 3-column launch reads - and 1 where it equals it "
                      "(the first plain pass: raw 25.4 MB for 25.2 MB; before the new walk also the later passes: raw 201.5 MB). The pass behind the virtual zero-padding pass (mode 2) gathers "
                      "8-byte coefficients (25.2 MB of them, each 64-byte line shared by 8 tiles of one XCD) plus L2-resident twiddle tables; the guide's factor 2 is kept there as an upper bound "
                      "(about a tenth of that launch's bytes either way). WRITE_SIZE is exact. Written by tools/process_profiles.py.",
               algorithmic_bytes=alg, traffic_over_algorithmic=dict(lde_3_columns=lde_traffic / alg["lde_3_columns"]), variants=variants),
          open(f"profiles/{R}_pmc_ntt_pass.json", "w"), indent=1)
print("roofline kernel traffic/launch MB", rv["hbm_bytes_per_launch"] / 1e6, "launches", rv["launches"], "LDE traffic / algorithmic", lde_traffic / alg["lde_3_columns"])
for n, v in variants.items():
    print("  ", n, "avg_us %.1f  MB/launch %.1f  TB/s on traffic %.2f" % (v["avg_us"], v["hbm_bytes_per_launch"] / 1e6, v["TBps_on_traffic"]))
rows = list(csv.DictReader(open(one(O + "/pmc_sq/*/*counter_collection.csv"))))
tr = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(one(O + "/pmc_sq/*/*kernel_trace.csv")))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    name = r["Kernel_Name"].replace("void msrt::ms_kmain_coop<", "").replace("void msrt::ms_kmain<", "").split(">(")[0] + ">"
    key = (name, int(r["Grid_Size"]))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); agg[key]["n"] += 1 / 8; agg[key]["us"] += tr.get(r["Dispatch_Id"], 0) / 8
lines = ["kernel,grid_threads,dispatches,avg_us,valu_wave_instr_per_thread,lds_instr_per_thread,active_pct,wait_inst_pct,valu_issue_bound_us_at_3.4cyc_2.4GHz,proofs_profiled"]
for key, c in sorted(agg.items(), key=lambda kv: -kv[1]["us"])[:40]:
    n = c["n"]; g = key[1]
    lines.append("%s,%d,%d,%.1f,%.0f,%.1f,%.0f,%.0f,%.1f" % (key[0].replace(",", ";"), g, round(n), c["us"] / n, c["SQ_INSTS_VALU"] * 64 / n / g, c["SQ_INSTS_LDS"] * 64 / n / g,
                 100 * c["SQ_ACTIVE_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), 100 * c["SQ_WAIT_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_INSTS_VALU"] / n * 3.4 / 1024 / 2400) + ",2")
open(f"profiles/{R}_sq_counters_top_kernels.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:10]))
try:
    shutil.copy(O + "/valu_rate.txt", f"profiles/{R}_valu_issue_rate.txt")
except Exception:
    pass
for extra in ("ntt_lab.log", "latency_probe.txt"):
    try:
        shutil.copy(O + "/" + extra, f"profiles/{R}_" + extra)
    except Exception:
        pass
# ---- BabyBear NTT passes (six-column LDE 2^20 -> 2^23 of tools/ntt_bench.py --field 1): traffic, algorithmic bytes, instructions
try:
    Fb = per_dispatch(one(O + "/bb_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
    Wb = per_dispatch(one(O + "/bb_write/*/*counter_collection.csv"), "WRITE_SIZE")
    Db = durations(one(O + "/bb_fetch/*/*kernel_trace.csv"))
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(one(O + "/bb_sq/*/*counter_collection.csv"))):
        if "PassKernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
            sq[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]); sq[short(r["Kernel_Name"])]["grid"] = float(r["Grid_Size"])
    bbv, N_, L_ = {}, 1 << 20, 1 << 23
    alg_launch = 6 * (N_ + L_) * 4 / 2     # (N + L) * s per column with s = 4 bytes, six columns, split over the transform's two passes
    for k in Fb:
        if "false" not in k:
            continue        # the forward (LDE) transforms; the INTT of the three trace columns is not part of the figure
        f, w, d = Fb[k], Wb.get(k, []), Db.get(k, [])
        n = min(len(f), len(w), len(d))
        if not n:
            continue
        name = short(k)
        mode2 = name.rstrip(">").rstrip().endswith((", 2", ", 3"))
        corr = 2.0 if mode2 or sum(f[:n]) < 0.75 * sum(w[:n]) else 1.0
        hb = sum(f[i] * 1024 * corr + w[i] * 1024 for i in range(n)) / n; us = sum(d[:n]) / n
        c = sq.get(name, {})
        nd = max(1.0, c.get("SQ_WAVE_CYCLES", 0) and n)   # the SQ pass ran the same launches
        bbv[name] = dict(launches=n, fetch_correction=corr, hbm_bytes_per_launch=hb, avg_us_in_pmc_run=us, TBps_on_traffic=hb / us / 1e6, frac_of_hbm_peak_on_traffic=hb / us / 1e6 / 8.0,
                         algorithmic_bytes_per_launch=alg_launch, algorithmic_GBps=alg_launch / us / 1e3, frac_of_hbm_peak_algorithmic=alg_launch / us / 1e3 / 8000.0,
                         traffic_over_algorithmic=hb / alg_launch,
                         valu_instr_per_element=(c.get("SQ_INSTS_VALU", 0) * 64 / nd) / (6 * L_) if c else None,
                         salu_instr_per_element=(c.get("SQ_INSTS_SALU", 0) * 64 / nd) / (6 * L_) if c else None,
                         wave_cycles_waiting_pct=100 * c.get("SQ_WAIT_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 0)) if c else None)
    rates = [json.loads(l) for l in open(O + "/ntt_bb.log") if l.startswith("{")]
    json.dump(dict(round=R, what="BabyBear coset LDE 6 x 2^20 -> 2^23 (u32 storage) on msntt::PassKernel2<BB, BB, ., 10, ., ., 3, .> (three sub-rounds; 512 threads x 64 KiB tiles in the "
                                  "later pass, 256 threads x 32 KiB tiles behind the virtual pass); FETCH_SIZE calibrated as for Goldilocks (tools/process_profiles.py)",
                   rates_without_profiler=rates, variants=bbv), open(f"profiles/{R}_pmc_ntt_babybear.json", "w"), indent=1)
    for n_, v in bbv.items():
        print("  BB", n_, "avg_us %.1f MB/launch %.1f traffic/alg %.2f VALU/elem %s" % (v["avg_us_in_pmc_run"], v["hbm_bytes_per_launch"] / 1e6, v["traffic_over_algorithmic"], v["valu_instr_per_element"]))
except Exception as e:  # noqa: BLE001
    print("BabyBear section skipped:", type(e).__name__, e)
for extra in ("sq_counters_ntt_passes.txt", "single_proof_timeline.txt", "ntt_gl.log", "ntt_bb.log", "sha_lab.log", "stride_probe.log", "io_probe.log"):
    try:
        shutil.copy(O + "/" + extra, f"profiles/{R}_" + extra)
    except Exception:
        pass
shutil.copy(O + "/bench_default.json", f"profiles/{R}_bench_default.json")
# which kernel source these counters belong to: bench.py compares it with the source it runs (a kernel change without re-profiling shows as stale)
import hashlib
hsh = hashlib.sha256()
for fn in ("ntt.hpp", "field.hpp", "merkle.hpp", "poly.hpp", "fri_tail.hpp"):
    hsh.update(open(os.path.join("mini-stark_amd", "csrc", fn), "rb").read())
json.dump({"round": R, "kernel_source_sha256": hsh.hexdigest(), "files": ["ntt.hpp", "field.hpp", "merkle.hpp", "poly.hpp", "fri_tail.hpp"], "profile_run_started_at": T0, "profile_run_rc": done["rc"],
           "every_pass_succeeded": True}, open(f"profiles/{R}_profile_meta.json", "w"))
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step")}), bench["roofline"])
