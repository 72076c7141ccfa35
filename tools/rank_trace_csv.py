"""profiles/<tag>_rank_trace.csv from the traces of tools/rank_trace.sh."""
import csv, glob, os, re, sys
tag = sys.argv[1]
FR = 10          # timed frames of rank_frame.py (after 3 warm-up frames)


def short(n):
    return next((k for k in ("k_wave_primary", "k_wave_bounce", "k_cover", "k_bins", "k_lists", "k_sky", "k_accumulate") if k in n), "memset" if "fillBuffer" in n else None)


def load(d):
    f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True), key=os.path.getmtime)[-1]
    rows = [x for x in csv.DictReader(open(f)) if short(x["Kernel_Name"])]
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    prim = [i for i, x in enumerate(rows) if "primary" in x["Kernel_Name"]]
    # frame k = from the first launch after the accumulate of frame k-1; take the last FR frames
    acc = [i for i, x in enumerate(rows) if "accumulate" in x["Kernel_Name"]]
    start = acc[-FR - 1] + 1
    win = rows[start:]
    by, n = {}, {}
    for x in win:
        k = short(x["Kernel_Name"])
        by[k] = by.get(k, 0) + int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
        n[k] = n.get(k, 0) + 1
    span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / FR / 1e3
    # the last frame's chain: critical path = everything but k_sky (side stream)
    return {k: v / FR / 1e3 for k, v in by.items()}, {k: v / FR for k, v in n.items()}, span


rank, rn, rspan = load(tag + "_rt_rank")
full, fn, fspan = load(tag + "_rt_full")
print("section,kernel,launches_per_frame,rank_frame_us,whole_frame_us,whole_frame_over_8_us,rank_minus_eighth_us")
order = ["memset", "k_cover", "k_bins", "k_lists", "k_wave_primary", "k_sky", "k_wave_bounce", "k_accumulate"]
tr = tf = 0.0
for k in order:
    r, f = rank.get(k, 0.0), full.get(k, 0.0)
    print("one_frame_at_a_time,%s,%g,%.1f,%.1f,%.1f,%.1f" % (k, rn.get(k, 0), r, f, f / 8, r - f / 8))
    if k != "k_sky":
        tr += r; tf += f
print("one_frame_at_a_time,sum_without_k_sky (k_sky runs beside the bounce levels),,%.1f,%.1f,%.1f,%.1f" % (tr, tf, tf / 8, tr - tf / 8))
print("one_frame_at_a_time,frame_to_frame (with launch gaps),,%.1f,%.1f,%.1f,%.1f" % (rspan, fspan, fspan / 8, rspan - fspan / 8))
for line in open("gpurun_out/%s_rt_inflight.log" % tag):
    m = re.match(r"world (\d+) spp (\d+) \S+ ms/frame with \[4\] frames in flight: ([\d.]+)", line)
    if m:
        print("four_frames_in_flight,world %s spp %s,,%.1f,,," % (m.group(1), m.group(2), float(m.group(3)) * 1e3))
