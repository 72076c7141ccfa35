#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a rocprofv3 --pmc output tree: python tools/parse_counters.py <dir>"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("k_"):].split("(")[0] if "k_" in k else k[:30]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in acc.items():
    if not k.startswith("k_"): continue
    print(k)
    for c, (s, n) in sorted(v.items()):
        print("   %-40s %18.0f per launch (%d launches)" % (c, s / n, n))
