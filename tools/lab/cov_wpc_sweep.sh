#!/bin/bash
# lab: the benchmark step by the covariance kernel's waves per CU (its 120-VGPR waves fill the register file at 16 per CU:
# does leaving room for an EVD / scan wave beside them pay?)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
B="--no-cpu-baseline --no-scan-roofline --no-other-configs"
for rep in 1 2; do
for w in 16 14 12 10 8; do
    export DOA_COV_WAVES_PER_CU=$w
    a=$(python3 bench.py --steps 20 --warmup 5 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step (K1 alone %.2f)' % (d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))")
    b=$(python3 bench.py --steps 300 --warmup 30 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step' % (d['ms_per_step']*1e3))")
    echo "cov waves/CU $w: 20 steps $a | 300 steps $b"
done
done
