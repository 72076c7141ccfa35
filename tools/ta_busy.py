#!/usr/bin/env python3
"""Texture-address (gather) unit load of the primary kernel from a tools/tcp_cfg2.sh tree:
   python tools/ta_busy.py gpurun_out/tcp_<tag>  -> JSON {k_wave_primary: {ta_busy_cycles, launch_cycles, ta_units, frac, ...}}
TA_TA_BUSY_sum adds the busy cycles of the 256 TA units (one per CU); GRBM_GUI_ACTIVE adds the launch's cycles over
the 8 XCDs: frac = TA_BUSY / (256 x GUI_ACTIVE / 8)."""
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/p2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k:
            continue
        k = k[k.find("k_"):].split("(")[0].split("<")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, v in acc.items():
    if "TA_TA_BUSY_sum" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    busy = v["TA_TA_BUSY_sum"][0] / v["TA_TA_BUSY_sum"][1]
    cyc = v["GRBM_GUI_ACTIVE"][0] / v["GRBM_GUI_ACTIVE"][1] / 8.0
    waves = v.get("TA_FLAT_READ_WAVEFRONTS_sum", [0.0, 1])
    out[k] = {"ta_busy_cycles": busy, "launch_cycles": cyc, "ta_units": 256, "frac": busy / (256.0 * cyc) if cyc else None,
              "gather_wave_instructions": waves[0] / waves[1], "launches": v["TA_TA_BUSY_sum"][1],
              "source": "rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE (tools/tcp_cfg2.sh, pass p2), bench.py --steps 2"}
print(json.dumps(out, indent=1))
