#!/bin/bash
# round 4 A/B: the packed fused kernel (mp_spec_traj_id_pk) at 2 / 3 / 4 blocks of 256 per CU asked of the compiler
# (152 VGPRs = 3 waves per SIMD as shipped; 4 blocks = 128 VGPRs with 38 scratch instructions in the re-evaluation block)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-3}
for round in $(seq $ROUNDS); do
  for f in "blocks2|MANIPULAPY_X=0" "blocks3|MANIPULAPY_HIP_JIT_DEFINES=MP_TRAJ_PK_BLOCKS=3" "blocks4|MANIPULAPY_HIP_JIT_DEFINES=MP_TRAJ_PK_BLOCKS=4" "blocks4_plain|MANIPULAPY_HIP_JIT_DEFINES=MP_TRAJ_PK_BLOCKS=4,MP_ADAPTIVE_F32=0" "blocks2_plain|MANIPULAPY_HIP_JIT_DEFINES=MP_ADAPTIVE_F32=0"; do
    name=${f%%|*}; kv=${f##*|}
    env $kv python $R/bench.py --config c2f --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f', '$name', d['roofline']['kernel_ms'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'))"
  done
done | tee /dev/stderr | python -c "
import sys, collections, statistics
d = collections.defaultdict(list)
for l in sys.stdin:
    c, n, v, p, ok = l.split(); d[(c, n)].append((float(v), float(p), ok))
for k, v in sorted(d.items()): print(k, 'kernel min %.4f  period min %.4f' % (min(x[0] for x in v), min(x[1] for x in v)), v)
"
