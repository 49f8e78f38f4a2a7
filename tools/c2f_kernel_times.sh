#!/bin/bash
# round 4: per-kernel durations of the fused configuration (c2f), shipped build and plain float32 build, from rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c2f_kt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adaptive -- python3 $R/bench.py --config c2f --steps 600 --warmup 3 --no-cpu-baseline > $OUT/adaptive.log 2>&1
export MANIPULAPY_HIP_EXPERIMENT=1 MANIPULAPY_HIP_JIT_DEFINES=MP_ADAPTIVE_F32=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plain -- python3 $R/bench.py --config c2f --steps 600 --warmup 3 --no-cpu-baseline > $OUT/plain.log 2>&1
cd $R
for v in adaptive plain; do
  f=$(find $OUT/$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in d.items():
    if "traj_id" in k:
        last = v[-600:] if len(v) >= 600 else v
        print(sys.argv[2], k[:40], "launches", len(v), "avg of last %d: %.2f us" % (len(last), sum(e - s for s, e in last) / len(last) / 1e3))
PY
done
