#!/bin/bash
# Copies what tools/collect_all.sh produced (merged back under gpurun_out/) into profiles/:
#   bash tools/finish_profiles.sh r02 [cfg5]
set -e
TAG=${1:-r02}
CFG=${2:-cfg2}
SFX=""
if [ "$CFG" != "cfg2" ]; then SFX="_$CFG"; fi
T=$TAG$SFX
mkdir -p profiles/counters
# gpurun merges every collection's raw files into the same local directories: keep only the newest file of each pass,
# or the parsers below average this collection with the ones before it
python3 - gpurun_out/sq_$T gpurun_out/pmc_$T gpurun_out/tcp_$T <<'PY'
import glob, os, sys
for top in sys.argv[1:]:
    by_dir = {}
    for f in glob.glob(top + "/**/*_counter_collection.csv", recursive=True):
        by_dir.setdefault(os.path.dirname(f), []).append(f)
    for d, fs in by_dir.items():
        fs.sort(key=os.path.getmtime)
        for f in fs[:-1]:
            os.remove(f)
            a = f.replace("_counter_collection.csv", "_agent_info.csv")
            if os.path.exists(a): os.remove(a)
PY
cp gpurun_out/${T}_bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/${T}_pmc_traffic.json profiles/pmc_traffic${SFX}.json
cp gpurun_out/${T}_sq_summary.txt profiles/${T}_sq_summary.txt
cp gpurun_out/${T}_tcp.txt profiles/counters/${T}_tcp.txt
test -s profiles/counters/${T}_tcp.txt || { echo "empty TCP summary"; exit 1; }
python3 tools/valu_busy.py gpurun_out/sq_$T profiles/${T}_kernel_stats.csv > profiles/valu_busy${SFX}.json
python3 tools/ta_busy.py gpurun_out/tcp_$T > profiles/ta_busy${SFX}.json
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find gpurun_out/pmc_$T/$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" profiles/${T}_pmc_$C.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_" in r["Kernel_Name"]]
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep); w.writeheader()
for r in rows: w.writerow({k: r[k] for k in keep})
PY
done
# bench.py attaches profiles/pmc_traffic*.json and profiles/valu_busy*.json as it finds them at run time, i.e. the
# PREVIOUS collection's; re-attach the ones this collection produced so the committed line is self-consistent.
python3 - profiles/${T}_bench.json "$SFX" <<'PY'
import json, sys
f, sfx = sys.argv[1], sys.argv[2]
d = json.load(open(f))
t = json.load(open("profiles/pmc_traffic%s.json" % sfx)).get("k_wave_primary")
v = json.load(open("profiles/valu_busy%s.json" % sfx)).get("k_wave_primary")
r = d["roofline"]
ta = json.load(open("profiles/ta_busy%s.json" % sfx)).get("k_wave_primary")
if ta and r.get("bound") == "gather_issue":
    r.update({"achieved": ta["ta_busy_cycles"], "peak": ta["ta_units"] * ta["launch_cycles"], "frac": ta["frac"],
              "gather_wave_instructions": ta["gather_wave_instructions"], "gather_source": "profiles/ta_busy%s.json: %s" % (sfx, ta.get("source"))})
if t:
    r["traffic"] = t["hbm_bytes_per_launch"]
    if r.get("bound") == "gather_issue" and "hbm" in r:
        gbps = t["hbm_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
        r["hbm"].update({"achieved": gbps, "frac": gbps / r["hbm"]["peak"]})
    if "hbm" in r: r["hbm"]["traffic"] = t["hbm_bytes_per_launch"]
    r["traffic_source"] = "profiles/pmc_traffic%s.json: %s" % (sfx, t.get("source"))
if v:
    r["executed"] = {"valu_busy": v["valu_busy"], "lane_utilisation": v["lane_utilisation"],
                     "issue_slot_frac": v["valu_busy"] * (v["lane_utilisation"] or 0.0), "waves_per_simd": v["waves_per_simd"],
                     "avg_launch_ns": v["avg_launch_ns"], "source": "profiles/valu_busy%s.json: %s" % (sfx, v.get("source"))}
    if v.get("valu_wave_insts"):
        r["issue_floor_ms"] = v["valu_wave_insts"] * 4.0 / (1024 * 2.4e9) * 1e3
        r["launch_ms"] = v["avg_launch_ns"] * 1e-6
        r["issue_floor_frac"] = r["issue_floor_ms"] / r["launch_ms"]
json.dump(d, open(f, "w"))
PY
ls -la profiles
