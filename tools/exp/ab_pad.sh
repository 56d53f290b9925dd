#!/bin/bash
# K1 alone and the 4-stream step against the byte distance between the N input streams (BENCH_CH_PAD added to the
# natural batch*K*8 bytes): do the streams alias in the memory system?
for rep in 1 2; do for v in ${PADS:--1 0 256 1024 4096 4352 65792 1052928}; do
  env BENCH_CH_PAD=$v timeout -k 5 200 python tools/bench_kernels.py ${BKARGS:-} --stages cov,mpipe --streams 4 --reps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pad $v', 'cov', round(d['cov_us'][0],2), round(d['cov_GBs']), 'GB/s  mpipe', round(d['mpipe_us'][0],2), round(d['mpipe_us'][1],2))
"
done; done
