#!/bin/bash
# One rank's frame next to the whole frame / 8, kernel by kernel (rocprofv3 --kernel-trace, one frame at a time), and the
# same two frames with 4 in flight: bash tools/rank_trace.sh <tag>  -> gpurun_out/<tag>_rank_trace.csv
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
export DEAL=0 TIMING=0
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_rt_rank -- python3 $R/tools/rank_frame.py 2 10 8 > $R/gpurun_out/${TAG}_rt_rank.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_rt_full -- python3 $R/tools/rank_frame.py 0 10 1 > $R/gpurun_out/${TAG}_rt_full.log 2>&1
cd $R
NSTREAMS=4 timeout -k 10 120 python3 tools/two_stream.py 8 64 60 > gpurun_out/${TAG}_rt_inflight.log
NSTREAMS=4 timeout -k 10 120 python3 tools/two_stream.py 1 64 16 >> gpurun_out/${TAG}_rt_inflight.log
NSTREAMS=4 timeout -k 10 120 python3 tools/two_stream.py 8 128 40 >> gpurun_out/${TAG}_rt_inflight.log
NSTREAMS=4 timeout -k 10 120 python3 tools/two_stream.py 8 256 24 >> gpurun_out/${TAG}_rt_inflight.log
python3 tools/rank_trace_csv.py $TAG > gpurun_out/${TAG}_rank_trace.csv
cat gpurun_out/${TAG}_rank_trace.csv
