#!/bin/bash
# round 4 A/B within one box: the fused launch's float64 pass parked (up to four launches' lists in one kernel) against run at once
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4; do
  for f in "parked|MANIPULAPY_X=0" "at_once|MANIPULAPY_HIP_FUSED_PARK=0"; do
    name=${f%%|*}; kv=${f##*|}
    env $kv python $R/bench.py --config c2f --steps 400 --warmup 10 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f', '$name', d['ms_per_step'])"
  done
done
