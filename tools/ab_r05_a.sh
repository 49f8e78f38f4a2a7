#!/bin/bash
# round 5, batch A (one box, interleaved rounds): what the conditioning test's joint set and the kernel's wave budget cost c2 / c2f,
# and what a caller that reuses ONE set of arrays gets (--input-sets 1: one float64 pass per launch)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra args
  MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline $4 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-22s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  run "default(j1)" c2 "MP_X=0"
  run "all-joints" c2 "MP_HARD_JOINTS=0x3e"
  run "plain" c2 "MP_ADAPTIVE_F32=0"
  run "waves4" c2 "MP_ID_CO_WAVES=4"
  run "waves3" c2 "MP_ID_CO_WAVES=3"
  run "default(j1) 1set" c2 "MP_X=0" "--input-sets 1"
  run "plain 1set" c2 "MP_ADAPTIVE_F32=0" "--input-sets 1"
  run "default(j1)" c2f "MP_X=0"
  run "all-joints" c2f "MP_HARD_JOINTS=0x3e"
  run "plain" c2f "MP_ADAPTIVE_F32=0"
done
