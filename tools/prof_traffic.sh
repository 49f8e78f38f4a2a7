#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of a config with given JIT defines: tools/prof_traffic.sh <config> <outdir-name> [defines]
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c5}; OUT=$R/gpurun_out/${2:-traffic_$CFG}; export MANIPULAPY_HIP_JIT_DEFINES="${3:-}"
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-single-set --no-clock-sample > $OUT/pmc_write.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fd_traj" in r["Kernel_Name"] or "mp_spec_id" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    m = statistics.mean(v) * 1024
    print(k, f"{m/1e6:.1f} MB raw" + (f"  (x2 = {2*m/1e6:.1f} MB for 128-byte requests tallied at 64)" if k[1] == "FETCH_SIZE" else ""), len(v), "launches")
PY
