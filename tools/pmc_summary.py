#!/usr/bin/env python3
"""Average PMC counter value per dispatch, per kernel, from rocprofv3 --pmc csv output."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if k.startswith("void at::") or k.startswith("__amd"): continue
    n = max(len(v) for v in cs.values())
    print(f"{k:42s} n={n:3d} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
