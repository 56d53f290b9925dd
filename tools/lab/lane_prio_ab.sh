#!/bin/bash
# lab (round 4): lanes on streams of different priorities (does the dispatcher then run the four first covariance kernels one
# after the other instead of interleaved, i.e. stagger the lanes for free?)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
B="--no-cpu-baseline --no-scan-roofline --no-other-configs"
for rep in 1 2 3; do
  for p in 0 1 2; do
    export DOA_LANE_PRIO=$p
    a=$(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1e3,2))')
    b=$(python bench.py --steps 200 --warmup 20 $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1e3,2))')
    echo "lane priorities $p: 20 steps $a us/step, 200 steps $b us/step"
  done
done
