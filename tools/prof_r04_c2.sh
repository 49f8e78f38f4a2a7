#!/bin/bash
# round 4: what the adaptive float32 rows cost c2, kernel by kernel (rocprofv3 kernel trace, last K dispatches = the timed steps)
export MANIPULAPY_HIP_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_r04_c2}
CFG=${2:-c2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "default|" "predictor_only|MP_HARD_ROW_K=1e30f" "K24|MP_HARD_ROW_K=24.0f" "plain|MP_ADAPTIVE_F32=0"; do
  name=${v%%|*}; export MANIPULAPY_HIP_JIT_DEFINES=${v##*|}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --config $CFG --steps 300 --warmup 10 --no-cpu-baseline --no-single-set > $OUT/$name.log 2>&1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
for name in ("default", "predictor_only", "K24", "plain"):
    line = [l for l in open(f"{out}/{name}.log").read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    for f in glob.glob(f"{out}/{name}/**/*kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        main = [r for r in rows if r["Kernel_Name"].startswith("mp_spec_id_co") or r["Kernel_Name"].startswith("mp_spec_traj_id_co")]
        K = 300
        last = main[-K:]
        t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
        inwin = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
        by = collections.defaultdict(list)
        for r in inwin: by[r["Kernel_Name"][:24]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print(name, "bench kernel_ms", round(b["roofline"]["kernel_ms"], 5), "period_us", round((t1 - t0) / K / 1e3, 2),
              {k: (len(v), round(sum(v) / len(v) / 1e3, 2)) for k, v in by.items()})
PY
