#!/bin/bash
# lab: stand-alone find_local_max, LDS-staged blocked kernels against the streaming mask kernel (DOA_K5_STREAM=1) and, for short
# vectors with more than one peak wanted, against the register-resident kernel (DOA_K5_STREAM=2)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for old in 0 1 2; do
  export DOA_K5_STREAM=$old
  for cfg in "16 1 4096" "16 3 4096" "4 2 2048" "4 2 1024" "4 3 1024" "4 2 512"; do
    set -- $cfg
    a=$(python tools/bench_kernels.py --N $1 --M $2 --P $3 --stages music,peak --reps 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('peak %.2f us' % d['peak_us'][0])")
    echo "variant=$old N=$1 M=$2 P=$3 | $a"
  done
done
