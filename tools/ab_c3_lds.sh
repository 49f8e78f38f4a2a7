#!/bin/bash
# c3: the Jacobian's staging pitch without padding (default) against round 1's padded pitch (MP_WSF_PAD_R01), interleaved
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4 5; do for f in "unpadded|MANIPULAPY_X=0" "r01_padding|MANIPULAPY_HIP_JIT_DEFINES=MP_WSF_PAD_R01"; do
  name=${f%%|*}; kv=${f##*|}
  env $kv python $R/bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', '$name', round(d['roofline']['kernel_ms'],4), round(d['roofline']['kernel_ms_cold'],4))"
done; done
