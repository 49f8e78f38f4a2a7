#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $3 python $R/bench.py --config $2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-4s %.4f ms  %.3e jt/s  %s' % ('$1', '$2', d['roofline']['kernel_ms'], d['value'], d['roofline']['kernel']))"; }
for round in 1 2; do
  run "packed" c2f "MANIPULAPY_HIP_F32=packed"
  run "scalar" c2f "MANIPULAPY_HIP_F32=scalar"
  run "default" c2 "X=1"
  run "default" c4s "X=1"
done
