#!/bin/bash
# Where the wavefronts of the fused loss kernels spend their cycles (SQ counters only + kernel trace; tools/loss_bench.py as the workload).
# usage (repo root, on the GPU box): bash tools/pmc_loss.sh <tag> [loss_bench flags]   -> gpurun_out/pmc_loss_<tag>.txt
set -eo pipefail
TAG=${1:-r}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_loss_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/tools/loss_bench.py --iters 10 "$@" > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw2 -- python3 $ROOT/tools/loss_bench.py --iters 10 "$@" > $OUT/run2.log 2>&1
python3 - $OUT/raw $OUT/raw2 $OUT/../pmc_loss_$TAG.txt <<'PY'
import collections, csv, glob, sys
out = open(sys.argv[3], "w")
for d in sys.argv[1:3]:
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "warp_loss" not in r["Kernel_Name"]:
                continue
            ctr[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    for k, c in ctr.items():
        out.write("%s  (%d launches)\n" % (k[:80], len(n[k])))
        for name, v in sorted(c.items()):
            out.write("   %-24s %.4g per launch\n" % (name, v / len(n[k])))
PY
rm -rf $OUT/raw $OUT/raw2
cat $OUT/../pmc_loss_$TAG.txt
