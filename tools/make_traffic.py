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
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"traffic_{cfg}.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
