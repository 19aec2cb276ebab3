#!/bin/bash
# How the bench line depends on --steps / --warmup on one box (the first timed step after the fence runs with the host not yet ahead: +2 ms)
for cfg in "20 5" "20 20" "50 5" "100 20" "20 5"; do set -- $cfg; python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('steps $1 warmup $2: %.3f ms %.1f/s'%(d['ms_per_step'],d['value']))"; done
