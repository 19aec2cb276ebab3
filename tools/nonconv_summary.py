"""Per-step time of the kernels that are not convolutions, from a rocpd kernel-statistics CSV (tools/rocpd_stats.py) of a 13-step run."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
conv = other = slab = 0.0
launches = 0.0
for r in rows:
    if not r["Calls"]:
        continue
    n = r["Name"]
    if n.startswith("#"):
        continue
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    launches += int(r["Calls"]) / steps
    if any(k in n for k in ("wgrad_reduce", "wgrad_presum")):      # part of the weight-gradient stage since round 3 (bench.py counts it there)
        slab += ms
        conv += ms
        continue
    if any(k in n for k in ("igemm", "wgrad_tab", "wgrad_kernel", "wgrad_bf16", "wgrad3x3_patch", "halo", "stem7x7", "conv3x3r", "conv3x3_patch", "splitk_finish")):
        conv += ms
        continue
    other += ms
    if ms > 0.008:
        print("%-60s calls/step %6.1f  ms/step %.3f  avg us %.1f" % (n.split("(")[0][-60:], int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
print("conv stage %.3f ms/step (of it the weight gradients' slab reduction %.3f), everything else %.3f ms/step, %.0f launches per step" % (conv, slab, other, launches))
