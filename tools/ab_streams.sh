#!/bin/bash
# Same box: the default three-stream issue, weight gradients in line (MCAV_WGRAD_SIDE=0), everything on one stream (--serial)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_streams.txt
: > $OUT
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 "$@" 2> /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f ms  %.1f /s' % (d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
  echo "default rep $rep: $(run)" >> $OUT
  echo "wgrad in line rep $rep: $(MCAV_WGRAD_SIDE=0 run)" >> $OUT
  echo "one stream rep $rep: $(run --serial)" >> $OUT
done
cat $OUT
