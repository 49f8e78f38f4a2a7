#!/bin/bash
# A/B of c5 roll-out kernel variants on ONE box, interleaved over several rounds (the kernel is power-limited: the same
# build measures 0.54 - 0.65 ms from run to run, so variants are alternated and min / median reported).
# usage: tools/ab_c5.sh "name1|defines1|flags1" "name2|defines2|flags2" ...
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-5}
OUT=$(mktemp)
for round in $(seq $ROUNDS); do
  for spec in "$@"; do
    IFS='|' read -r name D F <<< "$spec"
    MANIPULAPY_HIP_JIT_DEFINES="$D" MANIPULAPY_HIP_JIT_FLAGS="$F" python $R/bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['roofline']['kernel_ms'])" >> $OUT
  done
done
python - "$OUT" <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    k, v = line.rsplit(None, 1); d[k].append(float(v))
for k, v in d.items():
    print(f"{k:28s} min {min(v):.4f}  median {statistics.median(v):.4f}  max {max(v):.4f} ms  ({len(v)} runs)")
PY
