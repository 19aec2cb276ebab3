#!/bin/bash
# SQ / LDS / cache counters of one conv kernel on one layer shape (tools/conv_bench.py as the workload; counters + kernel trace only, three passes).
# usage (repo root, on the GPU box): CONV_BENCH_MMA=2 CONV_BENCH_SHAPES=1 bash tools/pmc_conv.sh <tag> <kernel name substring> [fwd|dgrad|wgrad]
#   -> gpurun_out/pmc_conv_<tag>.txt
set -eo pipefail
TAG=${1:-r}; KERN=${2:-patch}; WHAT=${3:-fwd}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_conv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CONV_BENCH_BATCH=${CONV_BENCH_BATCH:-24}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw1 -- python3 $ROOT/tools/conv_bench.py $WHAT > $OUT/run1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/raw2 -- python3 $ROOT/tools/conv_bench.py $WHAT > $OUT/run2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/raw3 -- python3 $ROOT/tools/conv_bench.py $WHAT > $OUT/run3.log 2>&1 || true
python3 - $OUT/raw1 $OUT/raw2 $OUT/raw3 $OUT/../pmc_conv_$TAG.txt "$KERN" <<'PY'
import collections, csv, glob, sys
out = open(sys.argv[4], "w")
for d in sys.argv[1:4]:
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[5] not in r["Kernel_Name"]:
                continue
            ctr[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    for k, c in ctr.items():
        out.write("%s  (%d launches)\n" % (k[:100], len(n[k])))
        for name, v in sorted(c.items()):
            out.write("   %-28s %.5g per launch\n" % (name, v / len(n[k])))
PY
rm -rf $OUT/raw1 $OUT/raw2 $OUT/raw3
cat $OUT/../pmc_conv_$TAG.txt
