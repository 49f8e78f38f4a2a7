#!/bin/bash
# roll-out kernel time against the horizon N (B fixed): per-step cost vs the address footprint one tile boundary sweeps
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in ${NS:-12 24 48 100 200 400}; do
  for B in ${BS:-131072}; do
    python $R/bench.py --config c5 --N $N --B $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print('B', $B, 'N', $N, 'kernel_ms %.4f' % k, 'us/step %.3f' % (k*1e3/$N), 'GB %.2f' % ($B*$N*6*4*5/1e9))"
  done
done
