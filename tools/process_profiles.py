#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import csv, glob, json, collections, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
O = f"gpurun_out/prof_{R}"
one = lambda pat: sorted(glob.glob(pat))[-1]
shutil.copy(one(O + "/stats/*/*kernel_stats.csv"), f"profiles/{R}_kernel_stats.csv")
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
def load(path, name):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if KN in r["Kernel_Name"] and r["Counter_Name"] == name]
f = load(one(O + "/pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
w = load(one(O + "/pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
n = len(f); fetch = sum(f) * 1024 * 2 / n; write = sum(w) * 1024 / n
json.dump(dict(kernel="msntt::" + KN, round=R, launches=n, fetch_bytes_per_launch_corrected=fetch, write_bytes_per_launch=write, hbm_bytes_per_launch=fetch + write,
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline` (2 proofs); "
                      "counter unit KB (x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), confirmed in r01a: an LDE pass writes 393216 KB = "
                      "6*2^23*8 B exactly while its raw FETCH_SIZE reads 197059 KB (half of the 402.65 MB it loads). Per-launch average over this kernel's launches only.",
               raw=dict(fetch_kb=f, write_kb=w)), open(f"profiles/{R}_pmc_ntt_pass.json", "w"), indent=1)
print("traffic/launch MB", (fetch + write) / 1e6, "launches", n)
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
for extra in ("ntt_lab.log",):
    try:
        shutil.copy(O + "/" + extra, f"profiles/{R}_" + extra)
    except Exception:
        pass
shutil.copy(O + "/bench_default.json", f"profiles/{R}_bench_default.json")
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step")}), bench["roofline"])
