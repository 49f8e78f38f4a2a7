#!/bin/bash
# PMC passes on the cache side of the roll-out kernel (L1 = TCP, L2 = TCC): tools/prof_c5_tcp.sh <outdir-name> [JIT defines]
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_c5_tcp}; D=${2:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MANIPULAPY_HIP_JIT_DEFINES="$D"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_avr" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum TCC_WRITEBACK_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$i.log 2>&1 || echo "set $i failed: $set"
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fd_traj" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:48s} {statistics.mean(acc[k]):16.0f}  ({len(acc[k])} launches)")
PY
