#!/bin/bash
# Interleaved A/B of two builds of the library on bench configurations:  bash tools/ab_lib.sh <variant> cfg1 [cfg2 ...]
# (python -m manipulapy_amd.build --variant <variant> -DX=... builds manipulapy_amd/libmanipula_hip_<variant>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; shift
for round in 1 2 3 4; do
  for v in base $V; do
    for cfg in "$@"; do
      if [ $v = base ]; then unset MANIPULAPY_HIP_LIB; else export MANIPULAPY_HIP_LIB=$R/manipulapy_amd/libmanipula_hip_$v.so; fi
      python3 $R/bench.py --config $cfg --no-cpu-baseline --no-single-set 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$cfg', '$v', 'kernel_ms', round(r['kernel_ms'],5), 'GHz', round(r['clock']['hz']/1e9,3), 'cold', round(r['kernel_ms_cold'],5))"
    done
  done
done
