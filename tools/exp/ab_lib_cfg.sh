#!/bin/bash
# same-box A/B of two library builds on one bench_kernels configuration: bash tools/exp/ab_lib_cfg.sh <base.so> <args...>
base=$1; shift
for rep in 1 2; do for which in base new; do
  pre=""; [ $which = base ] && pre="DOA_HIP_LIB=$PWD/$base"
  env $pre timeout -k 5 300 python tools/bench_kernels.py "$@" --stages music,pipe,mpipe --streams 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', 'music', round(d['music_us'][0],1), 'pipe', round(d['pipe_us'][0],1), 'mpipe', round(d['mpipe_us'][0],1), round(d['mpipe_us'][1],1))
"
done; done
