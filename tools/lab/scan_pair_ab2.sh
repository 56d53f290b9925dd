#!/bin/bash
# lab (round 4): proxy and real kernel on the same box
tools/lab/bin/scan_proxy_x4 262144 r12 | grep -v "linear nt\|set up"
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2; do
    for c in "0 8 16" "1 8 16" "1 8 12" "0 8 12" "1 8 8"; do
        set -- $c; export DOA_SCAN_PAIR=$1 DOA_SCAN_PAIR_STRIDE=$2 DOA_SCAN_LEAN_WAVES_PER_CU=$3
        echo "pair $1 stride $2 waves/CU $3 | $(run)"
    done
done
