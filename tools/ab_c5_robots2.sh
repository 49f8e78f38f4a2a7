#!/bin/bash
# roll-out kernel by robot (row size): default build vs switches given as JIT defines.  usage: tools/ab_c5_robots2.sh "name|defines" ...
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
for robot in ${ROBOTS:-xarm6 panda7 panda ur5 iiwa14}; do
  for round in 1 2 3; do
    for spec in "$@"; do
      IFS='|' read -r name D <<< "$spec"
      MANIPULAPY_HIP_JIT_DEFINES="$D" python $R/bench.py --config c5 --robot $robot --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$robot', d['config']['dof'], '$name', round(d['roofline']['kernel_ms'],4))"
    done
  done
done
