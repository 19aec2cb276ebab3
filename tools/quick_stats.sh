#!/bin/bash
# One-stream kernel statistics of the default workload (or "$@" extra bench flags) -> gpurun_out/<tag>_kernel_stats_serial.csv
set -eo pipefail
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qs_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/qs_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --serial "$@" > $OUT/${TAG}_stats.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(ls /tmp/qs_$TAG/*/*_results.db | head -1) $OUT/${TAG}_kernel_stats_serial.csv
