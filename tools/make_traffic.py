#!/usr/bin/env python3
"""profiles/traffic_<cfg>.json from a tools/prof_cfg.sh summary: tools/make_traffic.py <cfg> <summary.json> [kernel-substring] [issue-cycles]

issue-cycles: average SIMD-32 cycles one wave64 VALU instruction of this kernel occupies (default 2.0: float32 / integer;
float64 arithmetic runs at half rate, 4 cycles - pass the kernel's ISA-weighted mean, tools/spec_asm.py gives the mix)."""
import json
import os
import sys

cfg, path = sys.argv[1], sys.argv[2]
s = json.load(open(path))
# the kernel the run's bench line names (the dominant kernel of the step), unless given explicitly
want = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else (s.get("kernel_stats_timed_region") or {}).get("Name", "")
issue = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
name, t = next((k, v) for k, v in s["traffic"].items() if want in k)
n = s["counters"][name]["FETCH_SIZE"]["n"]
out = {
    "kernel": name, "config": cfg, "hbm_bytes_per_launch": t["hbm_bytes_per_launch"],
    "read_bytes_corrected": t["read_bytes_corrected"], "write_bytes": t["write_bytes"],
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof_cfg.sh), KiB -> bytes, FETCH_SIZE "
              f"doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section); mean over {n} launches",
    "source": os.environ.get("MP_TRAFFIC_SOURCE", os.path.relpath(path)),
}
# VALU issue figures for bench.py's `roofline_valu` (same PMC run): wave-level VALU instructions per launch, the shader clock
# the launch held (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the issue cost of a wave64 VALU instruction on a
# SIMD-32 (2 cycles, MI355X_MICROARCH.md 'Wave scheduling'; packed float32 forms take 4, tools/ubench_issue2.hip).
c = s["counters"][name]
stats = next((k for k in s.get("kernel_stats", []) if k["Name"].split("(")[0].endswith(name) or name in k["Name"]), None)
timed = s.get("kernel_stats_timed_region") or {}
if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and stats:
    # (the clock is GRBM cycles over the duration of the SAME launches the counter pass averaged: all of the pass's launches)
    dur_s = float(stats["AverageNs"]) * 1e-9
    out["valu"] = {"valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"], "issue_cycles_per_inst": issue,
                   "clock_hz": c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / dur_s, "simds": 1024,
                   "waves_per_launch": c.get("SQ_WAVES", {}).get("mean"),
                   "wave_cycles_split": {k: c[k]["mean"] / c["SQ_WAVE_CYCLES"]["mean"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")
                                         if k in c and "SQ_WAVE_CYCLES" in c},
                   "source": os.environ.get("MP_TRAFFIC_SOURCE", os.path.relpath(path))}
if timed.get("AverageNs"):
    out["kernel_ns_timed_region"] = {k: timed[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "bench_kernel_ms") if k in timed}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"traffic_{cfg}.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
