#!/bin/bash
# average duration of every launch of a frame, by its place in the frame (one frame at a time): bash tools/ktrace_levels.sh <world> <spp>
set -e
TAG=${TAG:-ktl}
export NSTREAMS=1 DEAL=0
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/two_stream.py ${1:-1} ${2:-16} 20 > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % os.environ["TAG"], recursive=True), key=os.path.getmtime)[-1]
rows = [x for x in csv.DictReader(open(f))]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
def short(n):
    return next((k for k in ("primary", "bounce", "k_cover", "k_bins", "k_lists", "k_sky", "accumulate") if k in n), "memset" if "fillBuffer" in n else None)
rows = [x for x in rows if short(x["Kernel_Name"])]
frames, cur = [], []
for x in rows:
    cur.append(x)
    if "accumulate" in x["Kernel_Name"]:
        frames.append(cur); cur = []
frames = frames[-15:]
n = len(frames[0])
assert all(len(fr) == n for fr in frames), [len(fr) for fr in frames]
t_prev_end = None
for i in range(n):
    d = sum(int(fr[i]["End_Timestamp"]) - int(fr[i]["Start_Timestamp"]) for fr in frames) / len(frames) / 1e3
    s = sum(int(fr[i]["Start_Timestamp"]) - int(fr[0]["Start_Timestamp"]) for fr in frames) / len(frames) / 1e3
    print("   %-10s start %8.1f us  dur %7.1f us" % (short(frames[0][i]["Kernel_Name"]), s, d))
span = sum(int(fr[-1]["End_Timestamp"]) - int(fr[0]["Start_Timestamp"]) for fr in frames) / len(frames) / 1e3
print("   frame span %.1f us" % span)
PY
