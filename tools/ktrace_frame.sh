#!/bin/bash
# per-launch start/duration of the last frame of a one-frame-at-a-time bench run:  bash tools/ktrace_frame.sh [bench args]
set -e
TAG=${TAG:-ktf}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rank-share --once --in-flight 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % os.environ["TAG"], recursive=True), key=os.path.getmtime)[-1]
rows = [x for x in csv.DictReader(open(f)) if any(k in x["Kernel_Name"] for k in ("k_wave", "k_accum", "k_cover", "k_bins", "k_lists", "k_sky", "fillBuffer"))]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
# the last frame = from the last-but-one 'cover' on ... simply the last 14 launches
last = rows[-28:-14]
t0 = int(last[0]["Start_Timestamp"])
for x in last:
    n = x["Kernel_Name"]
    short = next((k for k in ("primary", "bounce", "k_cover", "k_bins", "k_lists", "k_sky", "accumulate") if k in n), "memset" if "fillBuffer" in n else n[:20])
    print("   %-12s start %8.1f us  end %8.1f us  dur %7.1f us  grid %s" % (short, (int(x["Start_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3, x.get("Grid_Size", "?")))
PY
