#!/bin/bash
# Work a rank's frame costs against 1/8 of the whole frame's, kernel by kernel, in SQ counters (kernels serialised by
# the counter collection, so this is work, not overlap): bash tools/sq_rank.sh <tag>
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/sqrank_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DEAL=0 TIMING=0
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/rank -- python3 $REPO/tools/rank_frame.py 2 5 8 > $OUT/rank.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/full -- python3 $REPO/tools/rank_frame.py 0 5 1 > $OUT/full.log 2>&1
cd $REPO
python3 tools/sq_rank_parse.py $OUT 8
