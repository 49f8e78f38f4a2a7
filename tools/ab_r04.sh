#!/bin/bash
# round 4 A/B, interleaved: the float32 inverse dynamics with and without the adaptive-precision rows (MP_ADAPTIVE_F32), the fused
# kernel as flat scalar rows with whole-line tau (default) against the packed two-timesteps-per-lane form (MANIPULAPY_HIP_TRAJ_CO=0)
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-3}
for round in $(seq $ROUNDS); do
  for cfg in c2 c4 c4s c2f; do
    for f in "adaptive|MANIPULAPY_X=0" "plain_f32_w6|MANIPULAPY_HIP_JIT_DEFINES=MP_ADAPTIVE_F32=0,MP_ID_CO_WAVES=6,MP_TRAJ_CO_WAVES=6" "plain_f32_w5|MANIPULAPY_HIP_JIT_DEFINES=MP_ADAPTIVE_F32=0" "K24|MANIPULAPY_HIP_JIT_DEFINES=MP_HARD_ROW_K=24.0f" "in_place|MANIPULAPY_HIP_HARD_PASS=0" "flat_fused|MANIPULAPY_HIP_TRAJ_CO=1" "flat_fused_plain|MANIPULAPY_HIP_TRAJ_CO=1 MANIPULAPY_HIP_JIT_DEFINES=MP_ADAPTIVE_F32=0"; do
      name=${f%%|*}; kv=${f##*|}
      if [ "${name#flat_fused}" != "$name" ] && [ "$cfg" != c2f ]; then continue; fi
      env $kv python $R/bench.py --config $cfg --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$name', d['roofline']['kernel_ms'], d['roofline'].get('kernel'))"
    done
  done
done | tee /dev/stderr | python -c "
import sys, collections, statistics
d = collections.defaultdict(list)
for l in sys.stdin:
    c, n, v, k = l.split(); d[(c, n, k)].append(float(v))
for k, v in sorted(d.items()): print(k, 'min %.4f median %.4f' % (min(v), statistics.median(v)), v)
"
