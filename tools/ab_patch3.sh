#!/bin/bash
# Same box, alternating: the step with and without the twelve-wavefront forward / data-gradient patch kernel (tune build, MCAV_PATCH3=0|1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export MCAV_LIB_PATH=$ROOT/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth_tune.so
: > $OUT/ab_patch3.txt
for rep in 1 2; do
  for v in 0 1; do
    MCAV_PATCH3=$v python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 $@ 2> /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('patch3=$v rep=$rep: %.3f ms  %.1f /s' % (d['ms_per_step'], d['value']))" >> $OUT/ab_patch3.txt
  done
done
MCAV_PATCH3=1 python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --layer-report $OUT/p3_layers.txt > /dev/null 2>&1
cat $OUT/ab_patch3.txt
