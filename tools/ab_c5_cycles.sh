#!/bin/bash
# A/B of c5 roll-out kernel variants in CYCLES (GRBM_GUI_ACTIVE / 8 XCDs per launch, immune to the clock the box happens to
# hold) plus the wave-cycle split.  usage: tools/ab_c5_cycles.sh "name|defines" ...
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r name D <<< "$spec"
  OUT=$R/gpurun_out/abcyc_$name; rm -rf $OUT; mkdir -p $OUT
  MANIPULAPY_HIP_JIT_DEFINES="$D" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 $R/bench.py --config c5 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
  python3 - "$OUT" "$name" <<'PY'
import csv, glob, sys, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fd_traj" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
g = [x / 8 for x in acc["GRBM_GUI_ACTIVE"]]
w = statistics.mean(acc["SQ_WAVE_CYCLES"])
print(f"{sys.argv[2]:24s} kernel cycles: min {min(g):9.0f} median {statistics.median(g):9.0f}  | wave cycles: active {statistics.mean(acc['SQ_ACTIVE_INST_ANY'])/w:.2f} issue-stall {statistics.mean(acc['SQ_WAIT_INST_ANY'])/w:.2f} waitcnt {statistics.mean(acc['SQ_WAIT_ANY'])/w:.2f}  ({len(g)} launches)")
PY
done
