#!/bin/bash
# MFMA-pipe occupancy and effective clock per kernel (one PMC pass, counters only + kernel trace), one-stream run.
# usage (repo root, on the GPU box): bash tools/pmc_mfma.sh <tag>   -> gpurun_out/pmc_mfma_<tag>.csv
set -eo pipefail
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --serial --steps 3 --warmup 1 > $OUT/run.log 2>&1
python3 $ROOT/tools/pmc_mfma_summary.py $OUT/raw $OUT/../pmc_mfma_$TAG.csv
rm -rf $OUT/raw
