#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do for f in "wo_nt|MANIPULAPY_X=0" "plain|MANIPULAPY_HIP_JIT_DEFINES=MP_WO_NT=0"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config c2f --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'c2f', d['roofline']['kernel'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done
python $R/tools/time_ops.py 2>/dev/null | grep "batch_trajectory\|cartesian" | cut -c1-160
