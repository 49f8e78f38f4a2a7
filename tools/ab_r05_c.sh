#!/bin/bash
# round 5, batch C (one box, interleaved): (1) is three waves per SIMD (168 VGPRs: room for an unrolled float64 path inside the float32 kernel)
# as fast as five on c4 / c4s too?  (2) what a step costs with the test and the pushes but WITHOUT the pass kernel (the bound for hiding it)
# (3) is a hipGraph of plain launches slower than the stream too (i.e. the graph, not the passes)?
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra args, extra env
  env $5 MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline $4 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-22s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  for cfg in c2 c4 c4s; do
    run "default (5 waves)" $cfg "MP_X=0" "" "A=0"
    run "3 waves" $cfg "MP_ID_CO_WAVES=3" "" "A=0"
    run "no pass kernel" $cfg "MP_X=0" "" "MANIPULAPY_HIP_SKIP_PASS=1"
    run "plain" $cfg "MP_ADAPTIVE_F32=0" "" "A=0"
  done
  run "plain graph" c2 "MP_ADAPTIVE_F32=0" "--launch graph" "A=0"
  run "default graph" c2 "MP_X=0" "--launch graph" "A=0"
  run "no pass kernel" c2f "MP_X=0" "" "MANIPULAPY_HIP_SKIP_PASS=1"
  run "default" c2f "MP_X=0" "" "A=0"
done
