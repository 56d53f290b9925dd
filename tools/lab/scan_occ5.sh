#!/bin/bash
# lab: lean scan kernel compiled for five waves per SIMD (96 VGPRs, 6 spilled) against the shipped 104-VGPR build
for lib in libdoa_hip_lab.so libdoa_hip_lab_occ5.so; do
  export DOA_HIP_LIB=$PWD/_ab/$lib
  for cfg in "4 16 0" "4 20 1" "1 20 1" "1 24 1" "4 12 1"; do
    set -- $cfg
    export DOA_SCAN_WPB=$1 DOA_SCAN_LEAN_WAVES_PER_CU=$2 DOA_SCAN_NOTRIM=$3
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 batch 262144: //')
    echo "$lib wpb $1 wpc $2 | $a"
  done
done
DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab_occ5.so python tools/lab/scan_check.py 2>&1 | tail -1
DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so python tools/lab/scan_check.py 2>&1 | tail -1
