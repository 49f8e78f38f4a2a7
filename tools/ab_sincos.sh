#!/bin/bash
# round 4 A/B: sin / cos signs from the quadrant's bits (v_bitop3_b32: shift, and-xor) against compare + select (MP_SINCOS_SELECT_SIGNS)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for cfg in c2 c4 c4s c2f c5 c3; do
    for f in "bits|MANIPULAPY_X=0" "select|MANIPULAPY_HIP_JIT_DEFINES=MP_SINCOS_SELECT_SIGNS"; do
      name=${f%%|*}; kv=${f##*|}
      steps=300; [ $cfg = c3 ] && steps=20
      env $kv python $R/bench.py --config $cfg --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$name', d['ms_per_step'])"
    done
  done
done | tee /dev/stderr | python -c "
import sys, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    c, n, v = l.split(); d[(c, n)].append(float(v))
for k, v in sorted(d.items()): print(k, 'min %.5f mean %.5f' % (min(v), sum(v) / len(v)))
"
