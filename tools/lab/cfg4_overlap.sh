#!/bin/bash
# lab: cfg4 step (N=16, M=3, P=4096) serial and over 4 lanes, by the covariance kernel's waves per CU
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for w in 16 12 8 6 4; do
    export DOA_COV_MFMA_WAVES_PER_CU=$w
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['other_configs']['cfg4_n16']
print('cov waves/CU $w: serial %.1f us  lanes %.1f us' % (c['us_per_step_serial'], c['us_per_step_overlapped']))"
done
