#!/bin/bash
# kernel-trace summary of one bench configuration: bash tools/ktrace.sh <tag> [bench args]
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$TAG -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rank-share --once --in-flight 1 "$@" > $OUT/kt_$TAG.log 2>&1 || exit 1
find $OUT/kt_$TAG -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-60,100-400 | head -8
