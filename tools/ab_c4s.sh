#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do for f in "packed|MANIPULAPY_X=0" "scalar_co|MANIPULAPY_HIP_F32=scalar" "scalar_per_lane|MANIPULAPY_HIP_F32=scalar MANIPULAPY_X_NOCO=1"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config c4s --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'c4s', d['roofline']['kernel'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done
