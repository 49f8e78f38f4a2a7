#!/bin/bash
# round 5, batch D (one box, interleaved): the previous launch's float64 pass carried by the next launch's first workgroups
# (mp_body_id_lead) against a pass kernel of its own (MANIPULAPY_HIP_LEAD=0); stream and hipGraph
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra args, extra env
  env $5 MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline $4 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-26s ms_per_step %.5f kernel_ms %.5f frac %.3f 1set %s' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_single_set')), flush=True)"
}
for round in 1 2 3; do
  for cfg in c2 c4 c4s; do
    run "lead (default)" $cfg "MP_X=0" "" "A=0"
    run "pass kernel" $cfg "MP_X=0" "" "MANIPULAPY_HIP_LEAD=0"
    run "pass kernel, 5 waves" $cfg "MP_ID_CO_WAVES=5" "" "MANIPULAPY_HIP_LEAD=0"
    run "plain" $cfg "MP_ADAPTIVE_F32=0" "" "A=0"
  done
  run "lead graph" c2 "MP_X=0" "--launch graph" "A=0"
  run "plain graph" c2 "MP_ADAPTIVE_F32=0" "--launch graph" "A=0"
done
