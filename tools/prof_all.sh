#!/bin/bash
# tools/prof_cfg.sh for every configuration of the default bench line, into gpurun_out/<tag>_<cfg>/ (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
for cfg in ${2:-c2 c2f c4 c4s c5 c5b c3}; do
  bash $R/tools/prof_cfg.sh $cfg ${TAG}_$cfg > $R/gpurun_out/${TAG}_$cfg.log 2>&1
  echo "$cfg done: $(python3 -c "import json;s=json.load(open('$R/gpurun_out/${TAG}_$cfg/summary.json'));print(s.get('kernel_stats_timed_region'))" 2>&1 | cut -c1-300)"
done
