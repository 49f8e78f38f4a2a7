#!/bin/bash
# Profiling recipe for config c2 (run on the MI355X box from the repo root):
#   kernel trace + stats, then HBM traffic counters in separate --pmc passes (MI355X_MICROARCH.md §HBM).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_grbm.log 2>&1
cd $R
python3 tools/parse_prof.py $OUT
