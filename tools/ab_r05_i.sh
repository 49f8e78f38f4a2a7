#!/bin/bash
# round 5, batch I (one box, interleaved): lanes past a row span's last 16-byte chunk load and stage that chunk again (default) instead of
# being masked off (MP_STAGE_CLAMP=0): no exec-mask branches / zero fills around the partial instruction of every array
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, steps
  MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps $4 --warmup 10 --no-cpu-baseline --no-single-set 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-26s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  for cfg in c2 c4s; do
    run "clamped (default)" $cfg "MP_X=0" 300
    run "masked (old)" $cfg "MP_STAGE_CLAMP=0" 300
  done
  run "clamped (default)" c3 "MP_X=0" 20
  run "masked (old)" c3 "MP_STAGE_CLAMP=0" 20
done
