#!/bin/bash
# The round's bench lines beyond the default one: every BASELINE.json config shape that fits one GPU, each mode.  usage: bash tools/bench_matrix.sh <tag>
# -> gpurun_out/<tag>_bench_*.json (one JSON line each; stderr beside it)
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
run() { name=$1; shift; python3 $ROOT/bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/${TAG}_bench_$name.err; echo "$name: $(cut -c1-160 $OUT/${TAG}_bench_$name.json)"; }
run default --layer-report $OUT/${TAG}_layers.txt
run fp32_mfma --dtype fp32-mfma --no-cpu-baseline --steps 50 --warmup 10 --layer-report $OUT/${TAG}_layers_fp32_mfma.txt
run default_graph --graph --no-cpu-baseline --steps 50 --warmup 10
run batch4 --batch 4 --no-cpu-baseline --steps 50 --warmup 10
run batch4_graph --batch 4 --graph --no-cpu-baseline --steps 50 --warmup 10
run bf16 --dtype bf16 --no-cpu-baseline --steps 50 --warmup 10 --layer-report $OUT/${TAG}_layers_bf16.txt
run bf16_256x832 --dtype bf16 --height 256 --width 832 --no-cpu-baseline --steps 30 --warmup 10
run config3_r50_320x1024_ssim --depth-layers 50 --height 320 --width 1024 --ssim --no-cpu-baseline --steps 20 --warmup 5
run config0_eager --batch 2 --height 64 --width 128 --no-cpu-baseline --steps 100 --warmup 20
run config0_graph --batch 2 --height 64 --width 128 --graph --no-cpu-baseline --steps 100 --warmup 20
