#!/bin/bash
# PMC passes for an arbitrary bench config: tools/prof_cfg.sh <config> <outdir-name> [extra bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}; OUT=$R/gpurun_out/${2:-prof_$CFG}; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# timed launches must dominate the --stats average (the ramp alone is ~800 launches of the short kernels): 3000 steps, 20 for the 4 ms c3
STEPS=3000; [ "$CFG" = "c3" ] && STEPS=20
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline --no-single-set --no-clock-sample "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample "$@" > $OUT/pmc_grbm.log 2>&1
cd $R
python3 tools/parse_prof.py $OUT
