import csv, glob, os
f = sorted(glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % os.environ.get("TAG", "kti"), recursive=True), key=os.path.getmtime)[-1]
rows = [x for x in csv.DictReader(open(f))]
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
BIG = ("primary", "bounce", "k_sky", "accumulate")
def short(n):
    return next((k for k in ("primary", "bounce", "k_cover", "k_bins", "k_lists", "k_sky", "accumulate") if k in n), "memset" if "fillBuffer" in n else "other")
# two_stream.py with NSTREAMS=4: two runs of 8 warm-up + 30 frames; the steady state of the second
prim = [i for i, x in enumerate(rows) if "primary" in x["Kernel_Name"]]
idx = prim[38:76]
lo, hi = idx[12], idx[32]          # 20 frames
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
win = [x for x in rows if int(x["End_Timestamp"]) > t0 and int(x["Start_Timestamp"]) < t1]
ev = []
for x in win:
    s, e = max(int(x["Start_Timestamp"]), t0), min(int(x["End_Timestamp"]), t1)
    k = short(x["Kernel_Name"])
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
n_any = n_big = 0
hist_any, hist_big = {}, {}
last = t0
for t, d, k in ev:
    hist_any[n_any] = hist_any.get(n_any, 0) + t - last
    hist_big[n_big] = hist_big.get(n_big, 0) + t - last
    last = t
    n_any += d
    if k in BIG: n_big += d
T = t1 - t0
print("%.1f us per frame over 20 frames" % (T / 20e3))
print("kernels running at once  : " + "  ".join("%d: %4.1f %%" % (k, 100.0 * v / T) for k, v in sorted(hist_any.items()) if v))
print("big kernels running at once: " + "  ".join("%d: %4.1f %%" % (k, 100.0 * v / T) for k, v in sorted(hist_big.items()) if v))
by = {}
for x in win:
    by.setdefault(short(x["Kernel_Name"]), []).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
for k2, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("    %-10s n %3d  avg %7.1f us  sum/frame %7.1f us" % (k2, len(v), sum(v) / len(v) / 1e3, sum(v) / 20e3))
# queue ids
qs = {}
for x in win:
    qs.setdefault(x["Queue_Id"], set()).add(short(x["Kernel_Name"]))
print("queues:", {q: sorted(v) for q, v in qs.items()})
