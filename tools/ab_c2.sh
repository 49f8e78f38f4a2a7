#!/bin/bash
# A/B of the float32 inverse-dynamics kernel forms on ONE box (config c2 / c4 / c4s): packed two-rows-per-lane vs scalar one-row-per-lane
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, env
  env $3 python $R/bench.py --config $2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-4s %.4f ms  frac %.3f  %s' % ('$1', '$2', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['kernel_variant']))"
}
for round in 1 2; do
  for cfg in c2 c4 c4s; do
    run "packed (2 rows/lane)" $cfg "MANIPULAPY_HIP_F32=packed"
    run "scalar (1 row/lane)"  $cfg "MANIPULAPY_HIP_F32=scalar"
  done
done
