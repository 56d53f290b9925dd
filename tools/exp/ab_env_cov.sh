#!/bin/bash
# K1 alone + 4-stream step under several values of one switch (same box, 2 rounds)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v timeout -k 5 200 python tools/bench_kernels.py --stages cov,mpipe --streams 4 --reps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'cov', round(d['cov_us'][0],2), round(d['cov_GBs']), 'GB/s  mpipe', round(d['mpipe_us'][0],2), round(d['mpipe_us'][1],2))
"
  env $var=$v timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   $var=$v bench300 us/step', round(d['ms_per_step']*1e3,2))
"
done; done
