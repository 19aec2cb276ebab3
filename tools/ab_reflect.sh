#!/bin/bash
# Same box, alternating: the step with and without the reflection forms of the patch kernel (tune build: make variant NAME=tune FLAGS=-DMCAV_TUNE_ENV).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export MCAV_LIB_PATH=$ROOT/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth_tune.so
: > $OUT/ab_reflect.txt
for rep in 1 2; do
  for v in 0 1; do
    for minpix in 1000 400 0; do
      if [ $v = 0 ] && [ $minpix != 1000 ]; then continue; fi
      MCAV_PATCH_REFLECT=$v MCAV_PATCH_REFLECT_MINPIX=$minpix python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 $@ 2> /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('reflect=$v minpix=$minpix rep=$rep: %.3f ms  %.1f /s' % (d['ms_per_step'], d['value']))" >> $OUT/ab_reflect.txt
    done
  done
done
cat $OUT/ab_reflect.txt
