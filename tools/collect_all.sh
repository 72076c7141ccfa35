#!/bin/bash
# Everything profiles/ holds for a round and a config, in one GPU call (run from the repo root on the GPU box):
#   bash tools/collect_all.sh r02            (cfg2, the bench workload)
#   bash tools/collect_all.sh r02 cfg5       (any other --config: files get the suffix _cfg5)
# -> gpurun_out/<tag><sfx>_bench.json, _kernel_stats.csv, _pmc_traffic.json, _sq_summary.txt, _tcp.txt + the raw trees;
# tools/finish_profiles.sh copies them into profiles/ here.  Every GPU step has its own time limit and the
# script stops at the first one that fails.
set -e
TAG=${1:-r02}
CFG=${2:-cfg2}
SFX=""; ARGS=""
if [ "$CFG" != "cfg2" ]; then SFX="_$CFG"; ARGS="--config $CFG"; fi
REPO=$PWD
OUT=$REPO/gpurun_out
T=$TAG$SFX
mkdir -p $OUT
timeout -k 10 500 python3 bench.py $ARGS > $OUT/${T}_bench.json
echo bench-done
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T}_trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --once --in-flight 1 $ARGS > $OUT/${T}_trace.log 2>&1 )
echo trace-done
bash tools/collect_pmc.sh $T $ARGS
bash tools/collect_sq.sh $T $ARGS
bash tools/tcp_cfg2.sh $T $ARGS
python3 tools/parse_pmc.py $OUT/pmc_$T > $OUT/${T}_pmc_traffic.json
python3 tools/parse_sq.py $OUT/sq_$T > $OUT/${T}_sq_summary.txt
python3 tools/parse_counters.py $OUT/tcp_$T > $OUT/${T}_tcp.txt
find $OUT/${T}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${T}_kernel_stats.csv \;
echo all-done
