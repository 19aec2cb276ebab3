"""Per-step time of the kernels that are not convolutions, from a rocpd kernel-statistics CSV (tools/rocpd_stats.py) of a 13-step run."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
conv = other = 0.0
for r in rows:
    if not r["Calls"]:
        continue
    n = r["Name"]
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    if any(k in n for k in ("igemm", "wgrad_tab", "wgrad_kernel", "wgrad_bf16", "halo", "stem7x7", "conv3x3r", "splitk_finish")):
        conv += ms
        continue
    other += ms
    if ms > 0.008:
        print("%-60s calls/step %6.1f  ms/step %.3f  avg us %.1f" % (n.split("(")[0][-60:], int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
print("conv kernels %.3f ms/step, everything else %.3f ms/step" % (conv, other))
