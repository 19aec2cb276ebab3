#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/<name>_traffic.json: HBM bytes per launch for every kernel, with the gfx950 correction (FETCH_SIZE counts half
the bytes of wide coalesced reads: doubled; WRITE_SIZE is exact for 16-byte streaming stores).

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python3 bench.py ...   (same command)
"""
import collections
import csv
import glob
import json
import sys


def load(d, name):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                a = agg[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return agg


def main():
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, (v, n) in fe.items():
        w, wn = wr.get(k, [0.0, 1])
        out[k] = {"launches": n, "fetch_size_kb_per_launch_raw": v / n, "write_size_kb_per_launch_raw": w / max(1, wn),
                  "hbm_bytes_per_launch": (2.0 * v / n + w / max(1, wn)) * 1024.0}
    json.dump({"note": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
               "kernels": out}, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
