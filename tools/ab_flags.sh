#!/bin/bash
# A/B of extra hiprtc flags on several configs, interleaved: tools/ab_flags.sh "<flags>" cfg1 cfg2 ...   (flags comma-separated)
export MANIPULAPY_HIP_EXPERIMENT=1  # JIT_DEFINES / JIT_FLAGS are honoured only with this
R=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS="$1"; shift
for cfg in "$@"; do
  for round in 1 2 3 4; do
    for name in base flags; do
      F=""; [ $name = flags ] && F="$FLAGS"
      MANIPULAPY_HIP_JIT_FLAGS="$F" python $R/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null \
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
