#!/usr/bin/env python3
"""Per-kernel MFMA-pipe occupancy and effective clock from one rocprofv3 PMC pass (tools/pmc_mfma.sh).

For every kernel: launches, mean duration, effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (the guide's DVFS recipe; reads high
on dispatches shorter than ~0.3 ms), and MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles per XCD).
usage: python tools/pmc_mfma_summary.py <rocprof_dir> <out.csv>
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    ctr = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> dispatch -> counter -> value
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            c = ctr[r["Kernel_Name"]][r["Dispatch_Id"]]
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur = collections.defaultdict(dict)
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for k, disp in ctr.items():
        n = len(disp)
        ga = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in disp.values())
        mf = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in disp.values())
        ns = sum(dur[k].get(i, 0.0) for i in disp)
        if ns <= 0 or ga <= 0:
            continue
        clock_ghz = ga / 8.0 / ns
        rows.append((ns, k, n, ns / n / 1e3, clock_ghz, mf / (1024.0 * ga / 8.0)))
    rows.sort(reverse=True)
    w = csv.writer(open(sys.argv[2], "w", newline=""))
    w.writerow(["kernel", "launches", "mean_us", "effective_clock_GHz", "mfma_busy_fraction_of_active_cycles", "total_ms"])
    for ns, k, n, us, clk, busy in rows:
        w.writerow([k, n, "%.1f" % us, "%.3f" % clk, "%.3f" % busy, "%.3f" % (ns / 1e6)])


if __name__ == "__main__":
    main()
