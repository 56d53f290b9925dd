#!/bin/bash
# same-box A/B of two builds of the library: bash tools/exp/ab_lib.sh <base.so> [rounds]
# (the tree's own gr-doa_amd/lib/libdoa_hip.so is "new"; scan kernel alone at two batches, serial and 4-stream step)
base=$1; rounds=${2:-2}
one() {  # $1 label, $2 lib ("" = tree)
  local pre=""; [ -n "$2" ] && pre="DOA_HIP_LIB=$2"
  env $pre timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 30 2>/dev/null | sed "s/^/$1 /"
  env $pre timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | sed "s/^/$1 /"
  env $pre timeout -k 5 200 python tools/bench_kernels.py --stages music,pipe,mpipe --streams 4 --reps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'music', round(d['music_us'][0],2), 'pipe', round(d['pipe_us'][0],2), 'mpipe', round(d['mpipe_us'][0],2), round(d['mpipe_us'][1],2))
"
  env $pre timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 bench300 us/step', round(d['ms_per_step']*1e3,2))
"
  env $pre timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 bench20 us/step', round(d['ms_per_step']*1e3,2))
"
}
for r in $(seq $rounds); do one base $PWD/$base; one new ""; done
