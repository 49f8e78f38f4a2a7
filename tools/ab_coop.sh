#!/bin/bash
# c3 (fp64 FK + Jacobian + ID): non-temporal cooperative stores, whole-line non-temporal inputs / tau, both
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4 5 6; do for f in "plain|MANIPULAPY_HIP_JIT_DEFINES=MP_COOP_NT=0,MP_FK_CO=0" "coop_nt|MANIPULAPY_HIP_JIT_DEFINES=MP_FK_CO=0" "fk_co|MANIPULAPY_HIP_JIT_DEFINES=MP_COOP_NT=0" "both|MANIPULAPY_X=0"; do
IFS='|' read -r name kv <<< "$f"
env $kv python $R/bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'c3', d['roofline']['kernel'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done
python $R/bench.py --config c3 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parity c3 both', d['parity_sample'])"
