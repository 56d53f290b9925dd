#!/bin/bash
# lab: stand-alone find_local_max on long vectors, LDS-staged blocked kernel against the streaming mask kernel
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for old in 0 1; do
  export DOA_K5_STREAM=$old
  for cfg in "16 1 4096" "16 3 4096" "4 2 2048" "4 2 1536"; do
    set -- $cfg
    a=$(python tools/bench_kernels.py --N $1 --M $2 --P $3 --stages music,peak --reps 40 2>/dev/null | tail -1)
    echo "stream=$old N=$1 M=$2 P=$3 | $a"
  done
done
