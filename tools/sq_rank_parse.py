"""Sums the counters of tools/sq_rank.sh per kernel and frame: rank frame against the whole frame / world."""
import csv
import glob
import re
import sys
from collections import defaultdict

out, world = sys.argv[1], int(sys.argv[2])
FR = 8          # rank_frame.py: 3 warm-up + 5 frames


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.split(r"[<(]", r["Kernel_Name"].replace("void ", "").replace("rrt::", "").replace("(anonymous namespace)::", ""))[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                n[k] += 1
    return acc, n


ra, rn = load(out + "/rank")
fa, fn = load(out + "/full")
cs = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS"]
print("%-28s %6s | %s" % ("kernel (per frame)", "n", " | ".join("%22s" % c for c in cs)))
tot = defaultdict(lambda: [0.0, 0.0])
for k in sorted(set(ra) | set(fa)):
    cells = []
    for c in cs:
        r, f = ra[k].get(c, 0.0) / FR, fa[k].get(c, 0.0) / FR / world
        tot[c][0] += r; tot[c][1] += f
        cells.append("%9.3g/%9.3g %4.2f" % (r, f, r / f if f else 0.0))
    print("%-28s %3d/%2d | %s" % (k[:28], rn[k] // FR, fn[k] // FR, " | ".join(cells)))
print("%-28s        | %s" % ("total", " | ".join("%9.3g/%9.3g %4.2f" % (tot[c][0], tot[c][1], tot[c][0] / tot[c][1]) for c in cs)))
