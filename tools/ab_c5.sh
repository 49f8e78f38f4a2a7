#!/bin/bash
# A/B of the c5 roll-out kernel variants on ONE box: tools/ab_c5.sh  (prints kernel ms per variant, two rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, defines, flags
  D="$2" F="$3"
  MANIPULAPY_HIP_JIT_DEFINES="$D" MANIPULAPY_HIP_JIT_FLAGS="$F" python $R/bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.4f ms  frac %.3f' % ('$1', d['roofline']['kernel_ms'], d['roofline']['frac']))"
}
for round in 1 2; do
  run "scalar"              "MP_FD_PAIR=0" ""
  run "scalar max-ilp"      "MP_FD_PAIR=0" "-mllvm,-amdgpu-sched-strategy=max-ilp"
  run "scalar bias0"        "MP_FD_PAIR=0" "-mllvm,-amdgpu-schedule-metric-bias=0"
  run "pair"                "MP_FD_PAIR=1" ""
  run "pair max-ilp"        "MP_FD_PAIR=1" "-mllvm,-amdgpu-sched-strategy=max-ilp"
done
