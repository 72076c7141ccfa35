#!/bin/bash
# one bench line per non-headline config -> gpurun_out/<tag>_configs.jsonl
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_configs.jsonl
: > $OUT
for c in cfg1 cfg3 cfg4 cfg2scene many manyflat; do
  timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline --no-rank-share >> $OUT || exit 1
done
python3 - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    o = d.get("one_frame_at_a_time", {})
    print("%-60s %8.0f Mrays/s %8.3f ms  (one at a time %8.0f, %8.3f ms)" % (d["config"]["workload"][:60], d["value"], d["ms_per_step"], o.get("value", 0), o.get("ms_per_step", 0)))
PY
