#!/usr/bin/env python3
"""Timeline of ONE proof alone from a rocprofv3 --kernel-trace csv: kernel time, gaps between consecutive dispatches (launch gaps vs host round trips).
   python tools/trace_gaps.py DIR [first_kernel_substring]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# proofs start with TransposeInKernel
starts = [i for i, r in enumerate(rows) if "TransposeInKernel" in r["Kernel_Name"]]
if len(starts) < 3:
    print("need >= 3 proofs in the trace"); sys.exit(1)
a, b = starts[-2], starts[-1]     # the last complete proof
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
ktime = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
gaps.append(t1 - int(seg[-1]["End_Timestamp"]))
small = [g for g in gaps if g < 12000]
big = [g for g in gaps if g >= 12000]
print(f"proof wall {(t1 - t0) / 1e3:.1f} us; {len(seg)} launches; kernel time {ktime / 1e3:.1f} us; gaps < 12 us: {len(small)} totalling {sum(small) / 1e3:.1f} us (avg {sum(small) / max(1, len(small)) / 1e3:.2f}); gaps >= 12 us (host round trips): {len(big)} totalling {sum(big) / 1e3:.1f} us (avg {sum(big) / max(1, len(big)) / 1e3:.1f})")
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"]
    n = n.split("ms_kmain_coop<")[-1] if "ms_kmain_coop<" in n else n.split("ms_kmain<")[-1]
    n = n.split(">(")[0][:60]
    per[n][0] += 1; per[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:9.1f} us  {c:4d} x  {n}")
# the kernel BEFORE each big gap
bg = collections.Counter()
for i, g in enumerate(gaps):
    if g >= 12000:
        n = seg[i]["Kernel_Name"]; n = n.split("ms_kmain_coop<")[-1] if "ms_kmain_coop<" in n else n.split("ms_kmain<")[-1]
        bg[n.split(">(")[0][:50]] += g
print("host round trips by preceding kernel (us):", {k: round(v / 1e3, 1) for k, v in bg.most_common(8)})
if "--dump" in sys.argv:   # every launch of the proof in order: start offset, duration, gap behind it (us), kernel - the per-round picture of the latency path
    print("\n#   t_us    dur_us  gap_us  kernel")
    for i, r in enumerate(seg):
        n = r["Kernel_Name"]; n = n.split("ms_kmain_coop<")[-1] if "ms_kmain_coop<" in n else n.split("ms_kmain<")[-1]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} {gaps[i] / 1e3:7.1f}  {n.split('>(')[0][:70]}  grid={r.get('Grid_Size_X', '?')}")
