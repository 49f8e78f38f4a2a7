#!/bin/bash
# round 5, batch G (one box, interleaved): the carried float64 pass in mp_spec_id_co with the kernel held to FIVE waves per SIMD (96 VGPRs):
# the unrolled float64 path then spills (UR5: 16 scratch stores + 16 loads per row, in the leading workgroups only; the float32 rows' code
# stays free of scratch) - slower per carried row, but at the front of the kernel and without costing the float32 rows their fifth wave
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines, extra env
  env $4 MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline --no-single-set 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-30s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  for cfg in c2 c4 c4s; do
    run "pass kernel (default)" $cfg "MP_X=0" "A=0"
    run "carried, 5 waves + scratch" $cfg "MP_ID_LEAD=1,MP_ID_CO_WAVES=5,MP_ID_CO_WAVES_F=4" "MANIPULAPY_HIP_LEAD=1"
    run "plain" $cfg "MP_ADAPTIVE_F32=0" "A=0"
  done
done
