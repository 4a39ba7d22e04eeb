#!/usr/bin/env python3
"""Per (kernel, grid) averages of rocprofv3 --pmc counter_collection.csv files:  python tools/pmc_summary.py DIR [DIR ...]"""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(f"== {d}: no counter_collection.csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]
        name = name.split("ms_kmain_coop<")[-1] if "ms_kmain_coop<" in name else name.split("ms_kmain<")[-1]
        k = (name[:70], int(r["Grid_Size"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"== {d}")
    for k, v in agg.items():
        n = max(1, max(cnt[k].values())); g = k[1]
        nd = sum(cnt[k].values())
        print(f"{k[0]} grid {g} vgpr {k[2]} lds {k[3]} scratch {k[4]}: {n} dispatches, avg {dur[k] / nd:.1f} us")
        wc = v.get("SQ_WAVE_CYCLES", 0) / n
        for c, val in sorted(v.items()):
            val /= n
            extra = f"  per thread {val / g * 64:.1f}" if "INSTS" in c else (f"  {100 * val / wc:.0f}% of wave cycles" if wc and c != "SQ_WAVE_CYCLES" else "")
            print(f"    {c:24s} {val:14.0f}{extra}")
