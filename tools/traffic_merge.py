#!/usr/bin/env python3
"""Fold per-workload traffic entries (tools/traffic_collect.py) into profiles/hbm_traffic.json:
{"_format": 2, "entries": [entry, ...]}, one entry per workload (a new entry replaces the one of the same workload).

  traffic_merge.py profiles/hbm_traffic.json entry1.json [entry2.json ...]"""
import json, os, sys
path = sys.argv[1]
cur = {"_format": 2, "entries": []}
if os.path.exists(path):
    old = json.load(open(path))
    if old.get("_format") == 2:
        cur = old
for p in sys.argv[2:]:
    e = json.load(open(p))
    cur["entries"] = [x for x in cur["entries"] if x["_workload"] != e["_workload"]] + [e]
json.dump(cur, open(path, "w"), indent=1)
print(path, [x["_workload"] for x in cur["entries"]])
