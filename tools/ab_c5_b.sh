#!/bin/bash
# roll-out kernel: whole kernel vs the build without tile I/O, by trajectories per GPU (1 wave per SIMD = 65536)
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
for B in ${BS:-32768 65536 98304 131072 196608 262144}; do
  for spec in "all|" "noio|MP_FD_EXP_NOIN,MP_FD_EXP_NOOUT" "noout|MP_FD_EXP_NOOUT" "noin|MP_FD_EXP_NOIN"; do
    IFS='|' read -r name D <<< "$spec"
    MANIPULAPY_HIP_JIT_DEFINES="$D" python $R/bench.py --config c5 --B $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print('B', $B, '$name', 'kernel_ms %.4f' % k)"
  done
done
