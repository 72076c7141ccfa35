#!/bin/bash
# Timeline of rank frames in flight: bash tools/ktrace_inflight.sh [world] [spp]   (env DEAL=0/1)
set -e
TAG=${TAG:-kti}
WORLD=${1:-8}; SPP=${2:-64}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/two_stream.py $WORLD $SPP 30 > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % os.environ["TAG"], recursive=True), key=os.path.getmtime)[-1]
rows = [x for x in csv.DictReader(open(f))]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
def short(n):
    return next((k for k in ("primary", "bounce", "k_cover", "k_bins", "k_lists", "k_sky", "accumulate") if k in n), "memset" if "fillBuffer" in n else "other")
# two_stream.py runs 1, 2, 3, 4 streams twice; take the second "3 streams" run = find by counting primary launches: each run = 4 + 30 frames
prim = [i for i, x in enumerate(rows) if "primary" in x["Kernel_Name"]]
runs = [prim[k * 34: (k + 1) * 34] for k in range(8)]
for label, k in (("1 stream", 4), ("3 streams", 6)):
    idx = runs[k]
    lo, hi = idx[8], idx[28]          # 20 frames in the steady state
    win = rows[lo:hi]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
    ev = sorted([(int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in win])
    busy, cur_s, cur_e = 0, None, None
    for s, e in ev:
        e = min(e, t1)
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    by = {}
    for x in win:
        by.setdefault(short(x["Kernel_Name"]), []).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    print("%s: %.1f us per frame, some kernel running %.1f %% of the time" % (label, (t1 - t0) / 20e3, 100.0 * busy / (t1 - t0)))
    for k2, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("    %-10s n %3d  avg %7.1f us  sum/frame %7.1f us" % (k2, len(v), sum(v) / len(v) / 1e3, sum(v) / 20e3))
PY
