export MANIPULAPY_HIP_EXPERIMENT=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for nb in 1024 512 256 128; do
  export MANIPULAPY_HIP_HARD_BLOCKS=$nb
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pb_$nb -- python3 $R/bench.py --config c2 --steps 300 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pb_$nb.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
for nb in (1024, 512, 256, 128):
    line = [l for l in open(f"gpurun_out/pb_{nb}.log").read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    for f in glob.glob(f"gpurun_out/pb_{nb}/**/*kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        main = [r for r in rows if r["Kernel_Name"].startswith("mp_spec_id_co")][-300:]
        t0, t1 = int(main[0]["Start_Timestamp"]), int(main[-1]["End_Timestamp"])
        by = collections.defaultdict(list)
        for r in rows:
            if t0 <= int(r["Start_Timestamp"]) <= t1: by[r["Kernel_Name"][:20]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("blocks", nb, "period_us", round((t1 - t0) / 300 / 1e3, 2), {k: (len(v), round(sum(v) / len(v) / 1e3, 2)) for k, v in by.items()})
PY
