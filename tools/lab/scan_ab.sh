#!/bin/bash
# lab: same-box A/B of the lean scan kernel's launch knobs (lab build of the library)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for rep in 1 2 3; do
for cfg in "0 12" "0 16" "1 16" "1 20" "1 24"; do
    set -- $cfg
    export DOA_SCAN_ZLDS=$1 DOA_SCAN_LEAN_WAVES_PER_CU=$2
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 batch 262144: //')
    echo "zlds $1 wpc $2 | $a"
done
done
