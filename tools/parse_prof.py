#!/usr/bin/env python3
"""Summarise a tools/prof_c2.sh output directory: kernel stats + per-launch counter averages."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
summary = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows[:4]]
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "k_id" in name or "k_fk" in name or "k_traj" in name:
            acc[(name.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        summary.setdefault("counters", {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
