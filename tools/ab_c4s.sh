#!/bin/bash
# c4s (n = 7): packed two-rows-per-lane kernel (the r02 choice for odd n) against the one-row kernels, whole-line and per-lane
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do for f in "packed|MANIPULAPY_HIP_F32=packed" "scalar_co|MANIPULAPY_X=0" "scalar_per_lane|MANIPULAPY_HIP_ID_CO=0"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config c4s --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'c4s', d['roofline']['kernel'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done
