#!/bin/bash
# per-launch start/duration of the last frames of tools/rank_frame.py:  bash tools/ktrace_rank.sh "<rank> <frames> [world]" [ENV=...]
set -e
ARGS=${1:-"7 6"}
TAG=${TAG:-ktr}
export TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/rank_frame.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % os.environ["TAG"], recursive=True), key=os.path.getmtime)[-1]
rows = [x for x in csv.DictReader(open(f)) if any(k in x["Kernel_Name"] for k in ("k_wave", "k_accum", "k_cover", "k_bins", "k_lists", "k_sky", "fillBuffer"))]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
last = rows[-24:]
t0 = int(last[0]["Start_Timestamp"])
for x in last:
    name = x["Kernel_Name"]
    short = "primary" if "primary" in name else "bounce" if "bounce" in name else "cover" if "k_cover" in name else "bins" if "k_bins" in name else "lists" if "k_lists" in name else "sky" if "k_sky" in name else "memset" if "fillBuffer" in name else "accumulate"
    print("   %-10s queue %-3s start %8.1f us  end %8.1f us  dur %7.1f us" % (short, x.get("Queue_Id", "?"), (int(x["Start_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3))
PY
