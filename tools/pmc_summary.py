#!/usr/bin/env python3
"""Per (kernel, grid) averages of a rocprofv3 --pmc counter_collection.csv:  python tools/pmc_summary.py DIR [substring]"""
import csv, collections, glob, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("ms_kmain<")[-1][:60], int(r["Grid_Size"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in agg.items():
    n = max(1, cnt[k]); g = k[1]
    print(f"{k[0]} grid {g} vgpr {k[2]} lds {k[3]} scratch {k[4]}: {n} dispatches, avg {dur[k]/n:.1f} us")
    wc = v.get("SQ_WAVE_CYCLES", 0) / n
    for c, val in sorted(v.items()):
        val /= n
        extra = f"  per thread {val / g * 64:.1f}" if "INSTS" in c else (f"  {100 * val / wc:.0f}% of wave cycles" if wc and c != "SQ_WAVE_CYCLES" and "CYCLES" not in c[3:8] else "")
        print(f"    {c:24s} {val:14.0f}{extra}")
