#!/bin/bash
# Timeline of frames in flight (4 streams): bash tools/ktrace_inflight.sh [world] [spp]   (env DEAL=0/1)
set -e
TAG=${TAG:-kti}
WORLD=${1:-8}; SPP=${2:-64}
export NSTREAMS=4
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 $GRAFT_REPO_ROOT/tools/two_stream.py $WORLD $SPP 30 > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 tools/ktrace_inflight_parse.py
