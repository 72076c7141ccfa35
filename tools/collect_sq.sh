#!/bin/bash
# SQ issue/stall/lane-utilisation counters for the hot kernels (one pass, 8 SQ slots).
# Run on the GPU box from the repo root: bash tools/collect_sq.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/sq_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --once --in-flight 1 "$@" > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --once --in-flight 1 "$@" > $OUT/p2.log 2>&1 || true
echo sq-done
