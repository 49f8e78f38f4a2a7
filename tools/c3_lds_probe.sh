export MANIPULAPY_HIP_EXPERIMENT=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "default|" "no_row_stage|MP_FK_CO=0"; do
  name=${v%%|*}; export MANIPULAPY_HIP_JIT_DEFINES=${v##*|}
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/c3lds_$name -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/c3lds_$name.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in ("default", "no_row_stage"):
    for f in glob.glob(f"gpurun_out/c3lds_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fk_jac" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(name, {k: sum(v)/len(v) for k, v in acc.items()})
PY
