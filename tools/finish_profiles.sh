#!/bin/bash
# Copies what tools/collect_all.sh produced (merged back under gpurun_out/) into profiles/.
set -e
TAG=${1:-r01}
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/${TAG}_sq_summary.txt profiles/${TAG}_sq_summary.txt
python3 tools/valu_busy.py gpurun_out/sq_$TAG profiles/${TAG}_kernel_stats.csv > profiles/valu_busy.json
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find gpurun_out/pmc_$TAG/$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" profiles/${TAG}_pmc_$C.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_" in r["Kernel_Name"]]
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
done
# bench.py attaches profiles/pmc_traffic.json and profiles/valu_busy.json as it finds them at run time, i.e. the
# PREVIOUS collection's; re-attach the ones this collection produced so the committed line is self-consistent.
python3 - profiles/${TAG}_bench.json <<'PY'
import json, sys
f = sys.argv[1]
d = json.load(open(f))
t = json.load(open("profiles/pmc_traffic.json")).get("k_wave_primary")
v = json.load(open("profiles/valu_busy.json")).get("k_wave_primary")
if t:
    d["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
    d["roofline"]["traffic_source"] = t.get("source")
if v:
    d["roofline"]["valu_issue"] = v
json.dump(d, open(f, "w"))
PY
ls -la profiles
