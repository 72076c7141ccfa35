#!/bin/bash
# HBM traffic of the hot kernels from the TCC counters, one rocprofv3 pass per
# counter (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Run on the GPU
# box from the repo root:  bash tools/collect_pmc.sh <tag>
# then:  python tools/parse_pmc.py gpurun_out/pmc_<tag> > profiles/pmc_traffic.json
set -e
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $REPO/bench.py --steps 2 --warmup 1 --once --in-flight 1 --no-cpu-baseline "$@" > $OUT/$C.log 2>&1
done
echo pmc-done
