#!/bin/bash
# gpurun_out/<tag>_<cfg>/ (tools/prof_all.sh) -> profiles/<round>_<cfg>_{summary.json,kernel_stats.csv,bench.json} + profiles/traffic_<cfg>.json
#   bash tools/collect_profiles.sh prof_r05 r05
TAG=${1:-prof_r05}; RND=${2:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for cfg in c2 c2f c3 c4 c4s c5 c5b; do
  D=gpurun_out/${TAG}_$cfg
  [ -f $D/summary.json ] || { echo "$cfg: no summary"; continue; }
  cp $D/summary.json profiles/${RND}_${cfg}_summary.json
  f=$(find $D/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f" > profiles/${RND}_${cfg}_kernel_stats.csv
  grep '^{' $D/trace.log | tail -1 > profiles/${RND}_${cfg}_bench.json
  # issue cycles per VALU instruction: 2 (float32); the packed / float64-heavy kernels carry their ISA-weighted mean (tools/spec_asm.py)
  case $cfg in c2f) ISSUE=3.32;; c3) ISSUE=3.18;; *) ISSUE=2.0;; esac
  MP_TRAFFIC_SOURCE=profiles/${RND}_${cfg}_summary.json python3 tools/make_traffic.py $cfg profiles/${RND}_${cfg}_summary.json "" $ISSUE > /dev/null && echo "$cfg: traffic_$cfg.json written"
done
