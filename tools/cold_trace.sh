#!/bin/bash
# Trace of the cold launches of c2: kernel trace + HIP API trace of one bench run, the windows after each idle
# gap written by tools/cold_trace.py.  Run on the MI355X box from the repo root:  bash tools/cold_trace.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-cold}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/untraced.json 2> $OUT/untraced.err
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/trace.log 2>&1
cd $R
python3 tools/cold_trace.py $OUT/trace > $OUT/cold_windows.txt 2>&1
python3 - <<PY
import json
d = json.loads([l for l in open("$OUT/untraced.json") if l.startswith("{")][-1])
r = d["roofline"]
print("untraced: kernel_ms", r["kernel_ms"], "cold", r["kernel_ms_cold"], "frac", r["frac"], "frac_cold", r["frac_cold"])
PY
