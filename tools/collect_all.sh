#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call (run from the repo root on the GPU box):
#   bash tools/collect_all.sh r01
# -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, pmc_<tag>/, sq_<tag>/ ; copy into profiles/ afterwards
# (tools/finish_profiles.sh does that here).
set -e
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench.json
echo bench-done
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1 )
echo trace-done
bash tools/collect_pmc.sh $TAG
bash tools/collect_sq.sh $TAG
python3 tools/parse_pmc.py $OUT/pmc_$TAG > $OUT/${TAG}_pmc_traffic.json
python3 tools/parse_sq.py $OUT/sq_$TAG > $OUT/${TAG}_sq_summary.txt
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo all-done
