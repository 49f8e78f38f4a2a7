#!/bin/bash
# round 4 A/B: the c3 kernel (mp_spec_fk_jac_id_d) at 3 waves per SIMD asked (168 VGPRs, 12 scratch instructions; shipped) against 2 (176 VGPRs, none)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4 5 6; do
  for f in "waves3|MANIPULAPY_X=0" "waves2|MANIPULAPY_HIP_JIT_DEFINES=MP_FK_WAVES=2"; do
    name=${f%%|*}; kv=${f##*|}
    env $kv python $R/bench.py --config c3 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', '$name', d['ms_per_step'])"
  done
done
