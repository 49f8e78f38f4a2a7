#!/bin/bash
# round 5, batch K (one box, interleaved): what is left between the c2 headline and plain float32 rows, piece by piece:
#   plain | test + pushes only (no pass anywhere: MANIPULAPY_HIP_LEAD=0 + SKIP_PASS, wrong results) | default (pass carried) |
#   default with the float32 rows at a higher issue priority than the carried float64 waves (MP_ID_CO_PRIO) | K = 10 (fewer carried rows)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra env
  env $4 MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline --no-single-set 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-30s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  run "plain" c2 "MP_ADAPTIVE_F32=0,MP_ID_LEAD=0" "A=0"
  run "test + pushes, no pass" c2 "MP_ID_LEAD=0" "MANIPULAPY_HIP_LEAD=0 MANIPULAPY_HIP_SKIP_PASS=1"
  run "default (carried)" c2 "MP_X=0" "A=0"
  run "carried, f32 rows prio 2" c2 "MP_ID_CO_PRIO=2" "A=0"
  run "carried, f32 rows prio 3" c2 "MP_ID_CO_PRIO=3" "A=0"
  run "carried, K = 10" c2 "MP_HARD_ROW_K=10.0f" "A=0"
done
