#!/bin/bash
# lab (round 4): is it the record loads? pair kernel with every wave reusing its first record (results invalid)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2; do
    for c in "0 0" "1 0" "1 1"; do
        set -- $c; export DOA_SCAN_PAIR=$1 DOA_SCAN_CONST_RECORD=$2
        echo "pair $1 const-record $2 | $(run)"
    done
done
