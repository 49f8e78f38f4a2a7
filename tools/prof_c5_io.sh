#!/bin/bash
# PMC passes that split the roll-out kernel's memory-side stalls: tools/prof_c5_io.sh <outdir-name> [JIT defines]
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_c5_io}; D=${2:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MANIPULAPY_HIP_JIT_DEFINES="$D"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_a -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_b -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/pmc_c -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c.log 2>&1
cd $R
python3 tools/parse_prof.py $OUT | python3 -c "
import sys,json
s=json.load(sys.stdin)
for k,v in s.get('counters',{}).items():
    if 'fd_traj' in k: print(k, {a: round(b['mean']) for a,b in sorted(v.items())})
"
