#!/bin/bash
# L1 (TCP) / texture-address (TA) load of the trace kernels: bash tools/tcp_cfg2.sh <tag> [bench args]
set -e
TAG=${1:-t}; shift || true
OUT=$PWD/gpurun_out/tcp_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --once --in-flight 1 "$@" > $OUT/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --once --in-flight 1 "$@" > $OUT/p2.log 2>&1 || true
timeout -s KILL 150 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --once --in-flight 1 "$@" > $OUT/p3.log 2>&1 || true
echo tcp-done
