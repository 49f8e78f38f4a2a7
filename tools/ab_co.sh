#!/bin/bash
# whole-line non-temporal row movement (mp_spec_id_co: the default, six waves per SIMD asked for) against the per-lane kernel
# (MANIPULAPY_HIP_ID_CO=0) and against itself without the occupancy hint
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4; do for cfg in c2 c4; do for f in "per_lane|MANIPULAPY_HIP_ID_CO=0" "co_w1|MANIPULAPY_HIP_JIT_DEFINES=MP_ID_CO_WAVES=1" "co|MANIPULAPY_X=0"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$cfg', d['roofline']['kernel'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done; done
