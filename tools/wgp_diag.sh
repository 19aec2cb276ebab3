#!/bin/bash
# What the weight-gradient patch kernel's time is made of: timing-only builds (-DMCAV_WGP_DIAG=n: 1 no MFMAs, 2 no staging in the loop, 4 no slab
# stores, 7 all three) beside the shipped library, tools/conv_bench.py wgrad (GEMM + slab reduction) on the four trunk shapes at the step's batch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wgp_diag.txt
: > $OUT
for v in "" wgp1 wgp2 wgp4 wgp7; do
  lib=$ROOT/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth${v:+_$v}.so
  [ -f $lib ] || continue
  echo "== ${v:-shipped}" >> $OUT
  MCAV_LIB_PATH=$lib CONV_BENCH_BATCH=24 CONV_BENCH_MMA=2 CONV_BENCH_SHAPES=0,1,2,3 python3 $ROOT/tools/conv_bench.py wgrad 2>/dev/null >> $OUT
done
echo "== fp32 MFMA kernels" >> $OUT
CONV_BENCH_BATCH=24 CONV_BENCH_MMA=0 CONV_BENCH_SHAPES=0,1,2,3 python3 $ROOT/tools/conv_bench.py wgrad 2>/dev/null >> $OUT
cat $OUT
