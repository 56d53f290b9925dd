#!/bin/bash
run() { # label env...
  local label=$1; shift
  env "$@" timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 30 2>/dev/null | sed "s/^/$label /"
  env "$@" timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | sed "s/^/$label /"
  env "$@" timeout -k 5 200 python tools/bench_kernels.py --stages music,pipe,mpipe --streams 4 --reps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'music', round(d['music_us'][0],2), 'pipe', round(d['pipe_us'][0],2), 'mpipe', round(d['mpipe_us'][0],2), round(d['mpipe_us'][1],2))
"
  env "$@" timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label bench300 us/step', round(d['ms_per_step']*1e3,2))
"
}
for rep in 1 2; do
run base DOA_SCAN_ZLDS=0
run zlds12 DOA_SCAN_ZLDS=1
run zlds16 DOA_SCAN_ZLDS=1 DOA_SCAN_LEAN_WAVES_PER_CU=16
run zlds24 DOA_SCAN_ZLDS=1 DOA_SCAN_LEAN_WAVES_PER_CU=24
run zlds32 DOA_SCAN_ZLDS=1 DOA_SCAN_LEAN_WAVES_PER_CU=32
done
