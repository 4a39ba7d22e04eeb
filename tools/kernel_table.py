#!/usr/bin/env python3
"""Per (kernel, grid) dispatch count and average duration from a rocprofv3 --kernel-trace CSV directory."""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void msrt::ms_kmain<", "")[:58]
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    agg[(n, g)][0] += 1
    agg[(n, g)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pat = sys.argv[3] if len(sys.argv) > 3 else ""
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if pat in k[0] and top > 0:
        print(f"{k[0]:60s} grid {k[1]:9d}  x{v[0]:4d}  avg {v[1] / v[0]:8.1f} us  total {v[1] / 1e3:8.2f} ms")
        top -= 1
