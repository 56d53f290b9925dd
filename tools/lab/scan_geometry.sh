#!/bin/bash
# lab: lean scan kernel at batch 262144 by workgroup size (waves), waves per CU and ablation (lab build of the library)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for cfg in "4 16 0" "1 16 1" "4 12 1" "1 12 1" "1 8 1"; do
  for abl in 0 1 2; do
    set -- $cfg
    export DOA_SCAN_WPB=$1 DOA_SCAN_LEAN_WAVES_PER_CU=$2 DOA_SCAN_NOTRIM=$3 DOA_SCAN_ABLATE=$abl
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 batch 262144: //')
    echo "wpb $1 wpc $2 notrim $3 ablate $abl | $a"
  done
done
