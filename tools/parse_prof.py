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
# The timed region alone: --stats averages ramp + cold + warm-up + timed launches together.  bench.py launches the
# dominant kernel for the last time in its timed region (tools/prof_cfg.sh passes --no-cpu-baseline, the probes that follow
# launch other kernels), so the LAST K dispatches of that kernel in the trace are the K timed steps; K and the kernel's
# name come from the JSON line bench.py printed into trace.log.
try:
    line = [ln for ln in open(os.path.join(out, "trace.log")).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    K, kname = int(bench["steps"]), bench["roofline"]["kernel"]
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"])
        if len(d) >= K:
            dur = [e - b for b, e in d[-K:]]
            period = (d[-1][1] - d[-K][0]) / K
            summary["kernel_stats_timed_region"] = {
                "Name": kname, "Calls": K, "AverageNs": sum(dur) / K, "MinNs": min(dur), "MaxNs": max(dur), "LaunchPeriodNs": period,
                "of_dispatches_in_trace": len(d), "bench_kernel_ms": bench["roofline"]["kernel_ms"], "bench_ms_per_step": bench["ms_per_step"],
                "how": "the last K dispatches of the kernel in the kernel trace = the K timed steps of bench.py"}
except Exception as exc:  # older runs without a JSON line: the whole-run stats stay
    summary["kernel_stats_timed_region"] = {"error": str(exc)[:200]}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if any(t in r.get("Kernel_Name", "") for t in ("k_id", "k_fk", "k_traj", "k_fd", "mp_spec_"))]
    # per kernel name only the launches of its most frequent grid: the benchmark's own (mp_model_specialize's self-check
    # launches the same kernels on 128 rows once per model)
    grids = defaultdict(lambda: defaultdict(int))
    for r in rows:
        grids[r["Kernel_Name"]][r.get("Grid_Size", "")] += 1
    main_grid = {k: max(v, key=v.get) for k, v in grids.items()}
    for r in rows:
        name = r["Kernel_Name"]
        if r.get("Grid_Size", "") == main_grid[name]:
            acc[(name.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        summary.setdefault("counters", {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
# HBM traffic per launch of the dominant kernel: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
# half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section) -> doubled.
for k, c in summary.get("counters", {}).items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"]["mean"] * 1024 * 2, c["WRITE_SIZE"]["mean"] * 1024
        summary.setdefault("traffic", {})[k] = {"read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
print(json.dumps(summary, indent=1))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
