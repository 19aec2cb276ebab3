#!/bin/bash
# per-kernel durations of one weight-gradient shape (conv_bench wgrad), shipped library and the no-MFMA timing build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "" wgp1 wgp2; do
  lib=$ROOT/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth${v:+_$v}.so
  [ -f $lib ] || continue
  for sh in 0 3; do
    export MCAV_LIB_PATH=$lib CONV_BENCH_BATCH=24 CONV_BENCH_MMA=2 CONV_BENCH_SHAPES=$sh
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/wgp_prof_${v:-ship}_$sh -o p -- python3 $ROOT/tools/conv_bench.py wgrad > /dev/null 2>&1
    f=$(find $ROOT/gpurun_out/wgp_prof_${v:-ship}_$sh -name "*kernel_stats.csv" | head -1)
    echo "== ${v:-shipped} shape $sh" >> $ROOT/gpurun_out/wgp_prof.txt
    if [ -n "$f" ]; then head -6 "$f" | cut -c1-160 >> $ROOT/gpurun_out/wgp_prof.txt; else echo "(no kernel_stats.csv)" >> $ROOT/gpurun_out/wgp_prof.txt; fi
  done
done
cat $ROOT/gpurun_out/wgp_prof.txt
