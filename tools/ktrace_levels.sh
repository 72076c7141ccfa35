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
for x in rows:      # a frame starts at the first memset behind an accumulate pass (two of those a frame: sky tiles, heavy tiles)
    if short(x["Kernel_Name"]) == "memset" and any("accumulate" in y["Kernel_Name"] for y in cur):
        frames.append(cur); cur = []
    cur.append(x)
frames = frames[-15:]
# by (kernel, occurrence in the frame): the side stream's kernels do not keep their place in start order
keys, acc = [], {}
for fr in frames:
    seen = {}
    for x in fr:
        k = short(x["Kernel_Name"]); seen[k] = seen.get(k, 0) + 1
        key = (k, seen[k])
        if key not in acc: acc[key] = [0.0, 0.0, 0]; keys.append(key)
        a = acc[key]
        a[0] += int(x["Start_Timestamp"]) - int(fr[0]["Start_Timestamp"]); a[1] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"]); a[2] += 1
for key in sorted(keys, key=lambda k: acc[k][0] / acc[k][2]):
    a = acc[key]
    print("   %-10s start %8.1f us  dur %7.1f us" % (key[0], a[0] / a[2] / 1e3, a[1] / a[2] / 1e3))
span = sum(max(int(x["End_Timestamp"]) for x in fr) - int(fr[0]["Start_Timestamp"]) for fr in frames) / len(frames) / 1e3
print("   frame span %.1f us" % span)
PY
