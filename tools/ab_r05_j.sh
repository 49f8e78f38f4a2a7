#!/bin/bash
# round 5, batch J (one box, interleaved): mp_spec_id_co with every kernel argument its float32 rows need requested in the entry block
# (default: one scalar-load round trip) against one request per branch (MP_ID_CO_HOIST_ARGS=0: three dependent round trips)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines
  MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline --no-single-set 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-26s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3 4; do
  for cfg in c2 c4 c4s; do
    run "hoisted (default)" $cfg "MP_X=0"
    run "per branch (old)" $cfg "MP_ID_CO_HOIST_ARGS=0"
  done
  run "plain, hoisted" c2 "MP_ADAPTIVE_F32=0,MP_ID_LEAD=0"
done
