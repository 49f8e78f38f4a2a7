#!/usr/bin/env python3
"""What the first launches after an idle phase are made of.

    python3 tools/cold_trace.py <rocprofv3 output dir> [kernel substring]

Reads the kernel trace (and, when collected, the HIP API trace) of a `bench.py --config c2` run and prints, for every idle gap of
more than 0.25 s between two dispatches, the dispatches that follow it: start relative to the first, duration, gap to the previous
dispatch's end, name - and the HIP API calls of that window that took more than 20 us on the host.  The bench's cold figure is the
HIP-event time around the first `ncold` launches of the step after such a gap, / ncold."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "mp_spec_id_co"
NEXT = 9


def load(pattern):
    rows = []
    for f in glob.glob(os.path.join(out, "**", pattern), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in load("*kernel_trace.csv")))
api = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in load("*hip_api_trace.csv")))
report = {"dispatches": len(disp), "windows": []}
sustained = [e - b for b, e, k in disp if want in k]
if sustained:
    tail = sorted(sustained[-20:])
    report["sustained_last20_median_us"] = tail[len(tail) // 2] / 1e3
for i in range(1, len(disp)):
    if disp[i][0] - disp[i - 1][1] < 250_000_000:
        continue
    win = disp[i:i + NEXT]
    if not any(want in k for _, _, k in win):
        continue
    t0 = win[0][0]
    rows = []
    prev_end = None
    for b, e, k in win:
        rows.append({"start_us": round((b - t0) / 1e3, 2), "dur_us": round((e - b) / 1e3, 2),
                     "gap_us": None if prev_end is None else round((b - prev_end) / 1e3, 2), "kernel": k})
        prev_end = e
    mine = [r for r in rows if want in r["kernel"]]
    w = {"idle_before_ms": round((disp[i][0] - disp[i - 1][1]) / 1e6, 1), "dispatches": rows,
         "kernel_dur_us": [r["dur_us"] for r in mine],
         "span_of_first5_us": round((sorted(x for x in win if want in x[2])[:5][-1][1] - t0) / 1e3, 2) if len(mine) >= 5 else None}
    t1 = win[-1][1]
    calls = [{"at_us": round((b - t0) / 1e3, 1), "host_us": round((e - b) / 1e3, 1), "call": f} for b, e, f in api
             if b >= t0 - 300_000 and b <= t1 and e - b > 20_000]
    if api:
        w["hip_calls_over_20us"] = calls
    report["windows"].append(w)
print(json.dumps(report, indent=1))
json.dump(report, open(os.path.join(out, "cold_windows.json"), "w"), indent=1)
