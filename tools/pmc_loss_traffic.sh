#!/bin/bash
# Fabric traffic (L2 misses: FETCH_SIZE, WRITE_SIZE; separate passes) of the fused loss kernel alone, tools/loss_bench.py as the workload.
# usage (repo root, on the GPU box): bash tools/pmc_loss_traffic.sh <tag> [loss_bench flags]   -> gpurun_out/pmc_loss_traffic_<tag>.txt
set -eo pipefail
TAG=${1:-r}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $ROOT/tools/loss_bench.py --iters 10 "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $ROOT/tools/loss_bench.py --iters 10 "$@" > $OUT/w.log 2>&1
python3 - $OUT/f $OUT/w $ROOT/gpurun_out/pmc_loss_traffic_$TAG.txt <<'PY'
import collections, csv, glob, sys
res = {}
for d, name in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    per = collections.defaultdict(float)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "warp_loss" in r["Kernel_Name"] and r["Counter_Name"] == name:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(per.values())
    res[name] = (len(v), v[len(v) // 2] if v else 0.0, v[-1] if v else 0.0)
with open(sys.argv[3], "w") as o:
    for k, (n, med, mx) in res.items():
        o.write("%s: %d launches, median %.1f KB, max %.1f KB per launch (raw counter; the loss kernel's 4-byte loads are NOT doubled: DESIGN.md section 6)\n" % (k, n, med, mx))
    o.write("fabric bytes per launch (median): %.1f MB\n" % ((res["FETCH_SIZE"][1] + res["WRITE_SIZE"][1]) * 1024 / 1e6))
PY
rm -rf $OUT
cat $ROOT/gpurun_out/pmc_loss_traffic_$TAG.txt
