set -e
cd /tmp && export TMPDIR=/tmp
for r in 3 7; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktr_$r -- python3 $GRAFT_REPO_ROOT/tools/rank_frame.py $r 6 > $GRAFT_REPO_ROOT/gpurun_out/ktr_$r.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for r in (3, 7):
    f = glob.glob("gpurun_out/ktr_%d/**/*kernel_trace.csv" % r, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows = [x for x in rows if "k_wave" in x["Kernel_Name"] or "k_accum" in x["Kernel_Name"]]
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    last = rows[-10:]
    t0 = int(last[0]["Start_Timestamp"])
    print("rank", r)
    for x in last:
        print("   %-16s start %8.1f us  dur %7.1f us  grid %s" % (x["Kernel_Name"].split("(")[0].split("::")[-1][:16], (int(x["Start_Timestamp"]) - t0) / 1e3, (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3, x.get("Grid_Size", "")))
PY
