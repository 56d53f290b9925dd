#!/bin/bash
# same-box A/B of one environment switch on the scan kernel alone + the steps: bash tools/exp/ab_env_scan.sh VAR v1 v2 ...
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $var=$v timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 30 2>/dev/null | sed "s/^/$var=$v /"
  env $var=$v timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | sed "s/^/$var=$v /"
  env $var=$v timeout -k 5 200 python tools/bench_kernels.py --stages music,pipe,mpipe --streams 4 --reps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'music', round(d['music_us'][0],2), 'pipe', round(d['pipe_us'][0],2), 'mpipe', round(d['mpipe_us'][0],2), round(d['mpipe_us'][1],2))
"
  env $var=$v timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v bench300 us/step', round(d['ms_per_step']*1e3,2))
"
done
done
