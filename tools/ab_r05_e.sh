#!/bin/bash
# round 5, batch E (one box, interleaved): the fused launch's float64 pass carried by the next fused launch's first workgroups
# (default) against a pass kernel of its own (MANIPULAPY_HIP_LEAD_FUSED=0) and against plain float32 rows
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra args, extra env
  env $5 MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline $4 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-26s ms_per_step %.5f kernel_ms %.5f frac %.3f 1set %s' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_single_set')), flush=True)"
}
for round in 1 2 3 4; do
  run "carried (default)" c2f "MP_X=0" "" "A=0"
  run "pass kernel" c2f "MP_X=0" "" "MANIPULAPY_HIP_LEAD_FUSED=0"
  run "plain" c2f "MP_ADAPTIVE_F32=0" "" "A=0"
  run "default" c2 "MP_X=0" "" "A=0"
done
