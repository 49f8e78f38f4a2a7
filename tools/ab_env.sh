#!/bin/bash
# A/B of an environment switch on several configs, interleaved: tools/ab_env.sh VAR=value cfg1 cfg2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
KV="$1"; shift
for cfg in "$@"; do
  for round in 1 2 3 4; do
    for name in base "$KV"; do
      if [ "$name" = base ]; then E=""; else E="$KV"; fi
      env $E python $R/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$name', d['roofline']['kernel_ms'])"
    done
  done
done | python -c "
import sys, collections, statistics
d = collections.defaultdict(list)
for l in sys.stdin:
    c, n, v = l.split(); d[(c, n)].append(float(v))
for k, v in d.items(): print(k, 'min %.4f median %.4f' % (min(v), statistics.median(v)))
"
