#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs written by tools/collect_pmc.sh into
profiles/pmc_traffic.json: HBM bytes per launch for each hot kernel.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced streaming reads.  This path's reads are narrow, divergent (8-24 B
per lane) -- an uncalibrated width -- so both the raw and the doubled read
figure are kept and `hbm_bytes_per_launch` uses the raw one (lower bound)."""
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
                "k_accumulate" if "k_accumulate" in k else "k_resolve" if "k_resolve" in k else None
        if not short:
            continue
        fs, n = fetch[k]
        ws, nw = write.get(k, (0.0, 1))
        rd = fs / n * 1024.0
        wr = ws / max(nw, 1) * 1024.0
        out[short] = {"launches": n, "fetch_bytes_per_launch_raw": rd, "fetch_bytes_per_launch_x2": 2 * rd,
                      "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                      "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, "
                                "bench.py --steps 2 (tools/collect_pmc.sh); read side raw (narrow divergent loads, "
                                "x2 correction for wide streams not applied)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
