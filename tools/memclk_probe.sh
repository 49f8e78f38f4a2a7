#!/bin/bash
# What the memory side of a box is doing while c3 (22.5 GB per launch, 76 % writes) runs: rocm-smi clock levels and package power
# sampled twice a second beside `bench.py --config c3`.  c3's time differs by 15 % between boxes at a shader clock that is HIGHER on
# the slow ones; this looks for the clock that does differ.   bash tools/memclk_probe.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-memclk}
mkdir -p $OUT
python3 $R/bench.py --config c3 --steps 4000 --no-cpu-baseline --no-single-set > $OUT/c3.json 2>/dev/null &
BPID=$!
sleep 5
for i in $(seq 1 40); do
  kill -0 $BPID 2>/dev/null || break
  rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' >> $OUT/smi.jsonl; echo >> $OUT/smi.jsonl
  sleep 0.5
done
wait $BPID
python3 - <<PY
import json, collections
d = json.loads([l for l in open("$OUT/c3.json") if l.startswith("{")][-1])
print("c3 kernel_ms", round(d["roofline"]["kernel_ms"], 4), "frac", round(d["roofline"]["frac"], 3), "of_probe", round(d["roofline"].get("frac_of_probe", 0), 3),
      "shader clock sampled", round(d["roofline"]["clock"]["hz"] / 1e9, 3), "GHz")
vals = collections.defaultdict(list)
for l in open("$OUT/smi.jsonl"):
    l = l.strip()
    if not l.startswith("{"): continue
    try: j = json.loads(l)
    except Exception: continue
    for card, kv in j.items():
        for k, v in kv.items():
            vals[k].append(str(v))
for k, v in vals.items():
    c = collections.Counter(v)
    print(k, dict(c.most_common(4)))
PY
