R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do for cfg in c2 c4 c4s; do for f in packed scalar; do
MANIPULAPY_HIP_F32=$f python $R/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-specialize 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generic $f', '$cfg', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done; done
