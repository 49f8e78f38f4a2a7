export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
for robot in xarm6 panda panda7; do for D in "" "MP_FD_EXP_NOOUT=1" "MP_FD_EXP_NOIN=1,MP_FD_EXP_NOOUT=1"; do
for i in 1 2 3; do MANIPULAPY_HIP_JIT_DEFINES="$D" python $R/bench.py --config c5 --robot $robot --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$robot', '[$D]', round(d['roofline']['kernel_ms'],4), d['config']['dof'])"; done; done; done
