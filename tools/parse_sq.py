#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters written by tools/collect_sq.sh."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        import re
        m = re.search(r"(k_[a-z_]+)", k)
        k = m.group(1) if m else k[:30]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if not k.startswith("k_"): continue
    print(k)
    v = {c: s / n for c, (s, n) in cs.items()}
    for c in sorted(v): print("   %-26s %16.0f" % (c, v[c]))
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        print("   lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
    if "SQ_WAVE_CYCLES" in v:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
            if c in v: print("   %s / WAVE_CYCLES = %.3f" % (c, v[c] / v["SQ_WAVE_CYCLES"]))
