cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ic -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --once --in-flight 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/ic.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="primary" if "k_wave_primary" in n else "bounce" if "k_wave_bounce" in n else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
tail -3 gpurun_out/ic.log
