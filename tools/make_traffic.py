#!/usr/bin/env python3
"""profiles/traffic_<cfg>.json from a tools/prof_cfg.sh summary: tools/make_traffic.py <cfg> <summary.json> [kernel-substring]"""
import json
import os
import sys

cfg, path = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else ""
s = json.load(open(path))
name, t = next((k, v) for k, v in s["traffic"].items() if want in k)
n = s["counters"][name]["FETCH_SIZE"]["n"]
out = {
    "kernel": name, "config": cfg, "hbm_bytes_per_launch": t["hbm_bytes_per_launch"],
    "read_bytes_corrected": t["read_bytes_corrected"], "write_bytes": t["write_bytes"],
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof_cfg.sh), KiB -> bytes, FETCH_SIZE "
              f"doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section); mean over {n} launches",
    "source": os.path.relpath(path),
}
# VALU issue figures for bench.py's `roofline_valu` (same PMC run): wave-level VALU instructions per launch, the shader clock
# the launch held (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the issue cost of a wave64 VALU instruction on a
# SIMD-32 (2 cycles, MI355X_MICROARCH.md 'Wave scheduling'; packed float32 forms take 4, tools/ubench_issue2.hip).
c = s["counters"][name]
stats = next((k for k in s.get("kernel_stats", []) if k["Name"].split("(")[0].endswith(name) or name in k["Name"]), None)
if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and stats:
    dur_s = float(stats["AverageNs"]) * 1e-9
    out["valu"] = {"valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"], "issue_cycles_per_inst": 2.0,
                   "clock_hz": c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / dur_s, "simds": 1024,
                   "waves_per_launch": c.get("SQ_WAVES", {}).get("mean"),
                   "wave_cycles_split": {k: c[k]["mean"] / c["SQ_WAVE_CYCLES"]["mean"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")
                                         if k in c and "SQ_WAVE_CYCLES" in c},
                   "source": os.path.relpath(path)}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"traffic_{cfg}.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
