#!/bin/bash
# round 5, batch F (one box, interleaved): what a resident wave per SIMD is worth to mp_spec_id_co with the SAME code - unused LDS per
# block lowers the number of resident one-wave blocks per CU: 4608 B (c2) -> 20 blocks = 5 waves per SIMD; + 5632 -> 16 = 4; + 8960 -> 12 = 3
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name, config, defines
  MANIPULAPY_HIP_JIT_DEFINES="$3" python $R/bench.py --config $2 --steps 300 --warmup 10 --no-cpu-baseline --no-single-set 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-4s %-26s ms_per_step %.5f kernel_ms %.5f frac %.3f' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), flush=True)"
}
for round in 1 2 3; do
  run "5 waves (default)" c2 "MP_X=0"
  run "4 waves (LDS pad)" c2 "MP_ID_CO_LDS_PAD=5632"
  run "3 waves (LDS pad)" c2 "MP_ID_CO_LDS_PAD=8960"
  run "plain 5 waves" c2 "MP_ADAPTIVE_F32=0"
  run "plain 4 waves" c2 "MP_ADAPTIVE_F32=0,MP_ID_CO_LDS_PAD=5632"
done
