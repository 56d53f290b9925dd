#!/bin/bash
# lab: producer/consumer scan kernel (DOA_SCAN_PC = computing waves per workgroup), ablations
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for cfg in "0 0 0" "7 2 0" "7 2 1" "7 2 2" "3 4 0" "3 4 1" "3 4 2"; do
    set -- $cfg
    export DOA_SCAN_PC=$1 DOA_SCAN_PC_WG_PER_CU=$2 DOA_SCAN_PC_ABLATE=$3
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 batch 262144: //')
    echo "pc $1 wg/CU $2 ablate $3 | $a"
done
