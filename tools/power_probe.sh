#!/bin/bash
# sample the card's power / clocks while a config runs its sustained loop: tools/power_probe.sh <config> [steps] [JIT defines]
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c5}; STEPS=${2:-4000}; export MANIPULAPY_HIP_JIT_DEFINES="${3:-}"
python $R/bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline > /tmp/pp_bench.txt 2>/dev/null &
BP=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|junction\|edge" | tr '\n' ';'
  echo
  sleep 0.4
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/pp_bench.txt').read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'], 'cold', d['roofline'].get('kernel_ms_cold'))"
