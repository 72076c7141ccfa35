#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs written by tools/collect_pmc.sh into
profiles/pmc_traffic.json: HBM bytes per launch for each hot kernel.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
coalesced streaming read, WRITE_SIZE reads exactly.  The guide calibrates that
on 16 B/lane accesses and calls other widths uncalibrated; this path moves
8 B/lane, so it was calibrated on the round's first collections on k_accumulate,
whose traffic was then known exactly (it read 24 B per path sample and wrote
24 B per pixel, nothing else; profiles/r01_v1_pmc_*.csv): counter/known =
0.50 for reads, 1.00 for writes, i.e. the guide's factors hold at 8 B/lane
(the `calibration` entry carries these over).  Reads are
therefore doubled: `hbm_bytes_per_launch` = 2 x FETCH_SIZE + WRITE_SIZE."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (dirname, counter), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def main():
    d = sys.argv[1]
    fetch, write = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    out = {}
    for k in fetch:
        short = "k_wave_primary" if "k_wave_primary" in k else "k_wave_bounce" if "k_wave_bounce" in k else \
                "k_accumulate" if "k_accumulate" in k else None
        if not short:
            continue
        fs, n = fetch[k]
        ws, nw = write.get(k, (0.0, 1))
        rd = fs / n * 1024.0
        wr = ws / max(nw, 1) * 1024.0
        out[short] = {"launches": n, "fetch_bytes_per_launch_raw": rd, "fetch_bytes_per_launch_x2": 2 * rd,
                      "write_bytes_per_launch": wr, "hbm_bytes_per_launch": 2 * rd + wr,
                      "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, "
                                "bench.py --steps 2 (tools/collect_pmc.sh); gfx950 correction: reads x2 "
                                "(see `calibration`)"}
    # Calibration of the x2 read correction at this path's 8 B/lane access width: done on the round's first
    # collections, when k_accumulate streamed exactly 24 B per path sample in and 24 B per pixel out
    # (profiles/r01_v1_pmc_{FETCH,WRITE}_SIZE.csv: FETCH_SIZE x 1 KiB = 0.5000 x known reads, WRITE_SIZE x 1 KiB
    # = 1.000 x known writes).  Since the compact sky samples its traffic is data-dependent, so the ratios are
    # carried over, not re-measured.
    out["calibration"] = {"kernel": "k_accumulate (streaming version, profiles/r01_v1_pmc_*.csv)",
                          "fetch_raw_over_known": 0.5000, "write_over_known": 1.000, "carried_over": True}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
