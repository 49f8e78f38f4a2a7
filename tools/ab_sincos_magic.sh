#!/bin/bash
# round 4 A/B: sin / cos argument reduction rounded by the 1.5 * 2^23 trick inside an FMA (MP_SINCOS_MAGIC) against v_rndne + v_cvt
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4 5; do
  for cfg in c2 c2f c5; do
    for f in "rndne|MANIPULAPY_X=0" "magic|MANIPULAPY_HIP_JIT_DEFINES=MP_SINCOS_MAGIC"; do
      name=${f%%|*}; kv=${f##*|}
      env $kv python $R/bench.py --config $cfg --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$name', d['ms_per_step'])"
    done
  done
done | python -c "
import sys, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    c, n, v = l.split(); d[(c, n)].append(float(v))
for k, v in sorted(d.items()): print(k, 'min %.5f mean %.5f' % (min(v), sum(v) / len(v)))
"
