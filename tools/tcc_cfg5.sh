set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcc_cfg5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 1 --once --no-cpu-baseline > $OUT/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 1 --once --no-cpu-baseline > $OUT/p2.log 2>&1 || true
echo done
