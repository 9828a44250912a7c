#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats output directory: tools/kstats.py <dir> [n]"""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    print(f"{name:46s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
