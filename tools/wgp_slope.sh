#!/bin/bash
# kernel time against blocks per workgroup (batch 12 / 24 / 48 = 6 / 12 / 23 blocks): slope = cost of a block, intercept = set-up + epilogue
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $ROOT/gpurun_out/wgp_slope.txt
for v in "" wgp2 wgp8 wgp16 wgp24; do
  lib=$ROOT/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth${v:+_$v}.so
  [ -f $lib ] || continue
  for b in 24 48; do
    export MCAV_LIB_PATH=$lib CONV_BENCH_BATCH=$b CONV_BENCH_MMA=2 CONV_BENCH_SHAPES=0
    d=$ROOT/gpurun_out/wgp_slope_${v:-ship}_$b
    timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $ROOT/tools/conv_bench.py wgrad > /dev/null 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then echo "${v:-shipped} batch $b: $(grep wgrad3x3_patch "$f" | cut -d, -f2-4)" >> $ROOT/gpurun_out/wgp_slope.txt; fi
  done
done
cat $ROOT/gpurun_out/wgp_slope.txt
