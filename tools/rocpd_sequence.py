#!/usr/bin/env python3
"""The kernels of ONE step in launch order, from a rocprofv3 rocpd database of a one-stream run (bench.py --serial): what runs around a kernel
of interest (e.g. which launches the runtime's __amd_rocclr_copyBuffer sits between).

usage: python tools/rocpd_sequence.py <results.db> [pattern]     -> lines "index  start offset us  duration us  name"; a pattern marks matching
rows with '>>' and prints, per match, its two neighbours on either side only.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = c.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % name_col).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam" in r[0]]
    if len(adam) < 3:
        raise SystemExit("fewer than three steps in the trace")
    lo, hi = adam[-2] + 1, adam[-1] + 1                       # the last complete step
    step = rows[lo:hi]
    pat = sys.argv[2] if len(sys.argv) > 2 else None
    short = lambda n: n.split("(")[0].replace("void ", "").replace("mcav::", "")[-70:]
    t0 = step[0][1]
    keep = set(range(len(step)))
    if pat:
        hits = [i for i, r in enumerate(step) if pat in r[0]]
        keep = set(j for i in hits for j in range(max(0, i - 2), min(len(step), i + 3)))
    last = -2
    for i, (n, s, e) in enumerate(step):
        if i not in keep:
            continue
        if i != last + 1:
            print("   ...")
        last = i
        print("%s %4d %9.1f %8.1f  %s" % (">>" if pat and pat in n else "  ", i, (s - t0) / 1e3, (e - s) / 1e3, short(n)))
    print("%d launches in the step" % len(step))


if __name__ == "__main__":
    main()
