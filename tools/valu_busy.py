#!/usr/bin/env python3
"""VALU issue utilisation of the trace kernels from the SQ counters and the kernel trace:
   busy = SQ_ACTIVE_INST_VALU [quad-cycles, summed over SIMDs] / (SIMDs x launch duration x clock / 4)
   usage: valu_busy.py <sq dir> <kernel_stats.csv>  > profiles/valu_busy.json
MI355X: 256 CUs x 4 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)."""
import csv, glob, json, re, sys
from collections import defaultdict

SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
        if m:
            a = acc[m.group(1)][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    m = re.search(r"(k_[a-z_]+)", r["Name"])
    if m:
        dur[m.group(1)] = float(r["AverageNs"])
out = {}
for k, cs in acc.items():
    if k not in dur or "SQ_ACTIVE_INST_VALU" not in cs:
        continue
    v = {c: s / n for c, (s, n) in cs.items()}
    quad = SIMDS * dur[k] * 1e-9 * CLOCK_HZ / 4.0
    out[k] = {"avg_launch_ns": dur[k], "simd_quad_cycles": quad,
              "valu_busy": v["SQ_ACTIVE_INST_VALU"] / quad,
              "any_inst_busy": v.get("SQ_ACTIVE_INST_ANY", 0.0) / quad,
              "waves_per_simd": v.get("SQ_WAVE_CYCLES", 0.0) / quad,
              "valu_wave_insts": v.get("SQ_INSTS_VALU"), "salu_wave_insts": v.get("SQ_INSTS_SALU"),
              "lane_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64) if "SQ_THREAD_CYCLES_VALU" in v else None,
              "source": "rocprofv3 --pmc SQ_* (tools/collect_sq.sh) over rocprofv3 --kernel-trace durations; 1024 SIMDs, 2.4 GHz"}
json.dump(out, sys.stdout, indent=1)
print()
