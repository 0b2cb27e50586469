#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel (solver / post kernels only) and print per-wave figures."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if "solver" in k or "post_k" in k:
        print(k, "dispatches", len(disp[k]))
        for c, x in sorted(v.items()):
            print(f"   {c:28s} {x:16.0f}")
