#!/bin/bash
# round 5, batch B (one box, interleaved rounds): the conditioning test built from live values (MpRowScale, K = 8) against plain float32
# rows; the K timed launches as ONE hipGraph (captured launches now carry their float64 passes); one input set (a pass per launch)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra args
  MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline $4 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-22s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  run "default" c2 "MP_X=0"
  run "plain" c2 "MP_ADAPTIVE_F32=0"
  run "default graph" c2 "MP_X=0" "--launch graph"
  run "default 1set" c2 "MP_X=0" "--input-sets 1"
  run "default 1set graph" c2 "MP_X=0" "--input-sets 1 --launch graph"
  run "default" c2f "MP_X=0"
  run "plain" c2f "MP_ADAPTIVE_F32=0"
  run "default" c4 "MP_X=0"
  run "plain" c4 "MP_ADAPTIVE_F32=0"
done
