#!/bin/bash
# N=8 (and cfg4) 4-stream step: this tree against the round-1 tree (_ab/r01tree, built from the round-1 commit) and a few switches
run() { # label, script, extra env...
  local label=$1 script=$2; shift 2
  for cfg in "--N 8 --M 2" "--N 16 --M 3 --P 4096 --reps 20"; do
    env "$@" timeout -k 5 300 python $script $cfg --stages pipe,mpipe --streams 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label | $cfg |', 'pipe', round(d['pipe_us'][0],1), 'mpipe', round(d['mpipe_us'][0],1), round(d['mpipe_us'][1],1))
"
  done
}
for rep in 1 2; do
run r01 _ab/r01tree/tools/bench_kernels.py X=1
run new tools/bench_kernels.py X=1
run new_lean8 tools/bench_kernels.py DOA_SCAN_LEAN_WAVES_PER_CU=8
run new_cov16 tools/bench_kernels.py DOA_COV_WAVES_PER_CU=16
done
