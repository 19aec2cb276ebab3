#!/bin/bash
# Where the wavefronts of each kernel spend their cycles (one PMC pass, SQ counters only + kernel trace), one-stream run:
# SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; LDS bank conflicts.
# usage (repo root, on the GPU box): bash tools/pmc_stalls.sh <tag>   -> gpurun_out/pmc_stalls_<tag>.csv
set -eo pipefail
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_stalls_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --serial --steps 3 --warmup 1 > $OUT/run.log 2>&1
python3 - $OUT/raw $OUT/../pmc_stalls_$TAG.csv <<'PY'
import collections, csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        ctr[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]].add(r["Dispatch_Id"])
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"]
rows = sorted(ctr.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))
w = csv.writer(open(out, "w", newline=""))
w.writerow(["kernel", "launches", "wave_cycles", "parked_frac(WAIT_ANY)", "issue_stall_frac(WAIT_INST_ANY)", "issuing_frac(ACTIVE_INST_ANY)", "lds_issue_stall_frac",
            "lds_bank_conflict_frac_of_lds_cycles"])
for k, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    w.writerow([k, len(n[k]), "%.3g" % wc, "%.3f" % (c.get("SQ_WAIT_ANY", 0) / wc), "%.3f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc), "%.3f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc),
                "%.3f" % (c.get("SQ_WAIT_INST_LDS", 0) / wc), "%.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds > 0 else 0.0)])
PY
rm -rf $OUT/raw
