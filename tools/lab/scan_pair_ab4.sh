#!/bin/bash
# lab (round 4): does waiting for the previous burst (s_waitcnt vmcnt(0) before the next row / pair) help?
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2 3; do
    for c in "0 0 16" "0 1 16" "1 0 16" "1 1 16" "1 1 12" "0 1 12"; do
        set -- $c; export DOA_SCAN_PAIR=$1 DOA_SCAN_DRAIN=$2 DOA_SCAN_LEAN_WAVES_PER_CU=$3
        echo "pair $1 drain $2 waves/CU $3 | $(run)"
    done
done
