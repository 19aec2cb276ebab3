#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd SQLite database.

usage: python tools/rocpd_stats.py <results.db> [out.csv]
Same columns as rocprofv3's kernel_stats.csv (this ROCm writes the database by default; the CSV needs --output-format csv).
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = c.execute("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                     "group by s.%s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    span = c.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, mn, mx in rows:
        w.writerow([name, calls, tot, "%.1f" % (tot / calls), "%.2f" % (100.0 * tot / total), mn, mx])
    w.writerow(["# sum of kernel durations (ns)", "", total, "", "", "", ""])
    w.writerow(["# first start .. last end (ns)", "", span[1] - span[0], "", "", "", ""])


if __name__ == "__main__":
    main()
