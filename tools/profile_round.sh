#!/bin/bash
# Round-end evidence on the GPU box: kernel statistics (default three-stream run and a one-stream run), PMC FETCH_SIZE / WRITE_SIZE
# passes (separate runs, counters never combined with API traces), for the default workload and for the SSIM loss variant.
# usage (from the repo root on the box):  bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
set -eo pipefail
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $B --steps 10 --warmup 3 > $OUT/stats.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/stats/*/*_results.db | head -1) $OUT/kernel_stats.csv
echo stats done
rocprofv3 --kernel-trace --stats -d $OUT/stats_serial -- python3 $B --steps 10 --warmup 3 --serial > $OUT/stats_serial.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/stats_serial/*/*_results.db | head -1) $OUT/kernel_stats_serial.csv
python3 $ROOT/tools/rocpd_sequence.py $(ls $OUT/stats_serial/*/*_results.db | head -1) > $OUT/step_sequence.txt
python3 $ROOT/tools/rocpd_sequence.py $(ls $OUT/stats_serial/*/*_results.db | head -1) copyBuffer > $OUT/step_copies.txt
echo serial stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/traffic.json
echo pmc done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_ssim -- python3 $B --steps 2 --warmup 1 --ssim > $OUT/fetch_ssim.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_ssim -- python3 $B --steps 2 --warmup 1 --ssim > $OUT/write_ssim.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/fetch_ssim $OUT/write_ssim $OUT/traffic_ssim.json
echo ssim pmc done
rocprofv3 --kernel-trace --stats -d $OUT/stats_bf16 -- python3 $B --steps 10 --warmup 3 --serial --dtype bf16 > $OUT/stats_bf16.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/stats_bf16/*/*_results.db | head -1) $OUT/kernel_stats_serial_bf16.csv
echo bf16 stats done
rocprofv3 --kernel-trace --stats -d $OUT/stats_split -- python3 $B --steps 10 --warmup 3 --serial --dtype fp32-mfma > $OUT/stats_split.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/stats_split/*/*_results.db | head -1) $OUT/kernel_stats_serial_fp32_mfma.csv
echo fp32-mfma stats done
rm -rf $OUT/stats $OUT/stats_serial $OUT/fetch $OUT/write $OUT/fetch_ssim $OUT/write_ssim $OUT/stats_bf16 $OUT/stats_split
