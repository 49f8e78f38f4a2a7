#!/bin/bash
# A/B of the generic (no hiprtc) float32 inverse-dynamics kernels on one box: packed two-rows-per-lane (kernel-argument model),
# scalar one-row-per-lane (kernel-argument model), scalar with the model read from device memory joint by joint.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do for cfg in c2 c4 c4s; do for f in "packed|MANIPULAPY_HIP_F32=packed" "scalar|MANIPULAPY_HIP_F32=scalar" "dm|MANIPULAPY_X=0"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-specialize 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generic $name', '$cfg', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done; done
