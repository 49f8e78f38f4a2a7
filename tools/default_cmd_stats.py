#!/usr/bin/env python3
"""Per-kernel figures of a rocprofv3 kernel trace of the DEFAULT bench command (every configuration in one process): dispatches are
grouped by (kernel name, grid size) - c2, c4 and c4s launch the same symbol from different code objects - and each group's count,
average, minimum and maximum duration are printed beside the bench line's own figure for the configuration that launches it.

    python3 tools/default_cmd_stats.py <rocprofv3 output dir> <bench line json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, line = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (CONFIGS: the launch sizes)

d = json.loads([l for l in open(line) if l.startswith("{")][-1])
cfgs = {"c2": d, **{k: v for k, v in (d.get("configs") or {}).items()}}


def base_threads(name):
    c = bench.CONFIGS[name]
    if c["op"] == "fused":            # 256-lane workgroups, ceil(N / 2 / 256) of them per trajectory
        return c["B"] * (((c["N"] + 1) // 2 + 255) // 256) * 256
    if c["op"] == "fd_traj":          # one lane per trajectory
        return c["B"]
    return c["B"] * c["N"]            # one lane per row


# a launch may carry up to 512 leading workgroups (the previous launch's float64 pass): grid = base ... base + 512 x 256 lanes
buckets = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].strip()
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        owners = []
        for k, e in cfgs.items():
            rl = e.get("roofline") or {}
            if k in bench.CONFIGS and (rl.get("kernel") or e.get("kernel")) == name and base_threads(k) <= grid <= base_threads(k) + 512 * 256:
                owners.append(k)
        if owners:
            buckets[(name, " / ".join(owners))].append(dur)
rows = []
for (name, who), v in buckets.items():
    v.sort()
    km = [((cfgs[k].get("roofline") or {}).get("kernel_ms") or cfgs[k].get("kernel_ms")) for k in who.split(" / ")]
    rows.append({"kernel": name, "config": who, "dispatches": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "median_us": round(v[len(v) // 2] / 1e3, 2),
                 "min_us": round(v[0] / 1e3, 2), "max_us": round(v[-1] / 1e3, 2), "bench_kernel_ms": [round(x, 5) for x in km]})
rows.sort(key=lambda r: r["config"])
print(json.dumps({"command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 (under rocprofv3 --kernel-trace --stats)",
                  "what": "every dispatch of a configuration's kernel in the whole run (cold windows, ramps, warm-up, timed steps, clock pass, single-set loop); "
                          "c4 and c4s launch the same symbol on the same number of rows and cannot be told apart in a trace",
                  "groups": rows}, indent=1))
