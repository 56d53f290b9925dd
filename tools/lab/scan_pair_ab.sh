#!/bin/bash
# lab (round 4): same-box A/B of the lean scan kernel's row-pair form (lab build of the library), with a bit comparison
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for c in "0 8" "1 8" "1 16"; do set -- $c; export DOA_SCAN_PAIR=$1 DOA_SCAN_PAIR_STRIDE=$2; echo "pair $1 stride $2 | $(timeout -k 5 100 python tools/lab/scan_check.py 2>&1 | tail -1)"; done
for rep in 1 2 3; do
    for c in "0 8" "1 8" "1 16" "1 4" "1 32"; do
        set -- $c; export DOA_SCAN_PAIR=$1 DOA_SCAN_PAIR_STRIDE=$2
        echo "pair $1 stride $2 | $(run)"
    done
done
export DOA_SCAN_PAIR_STRIDE=8
echo "-- ablations (1 = no row stores, 2 = row stores only)"
for p in 0 1; do for abl in 1 2; do export DOA_SCAN_PAIR=$p DOA_SCAN_ABLATE=$abl; echo "pair $p ablate $abl | $(run)"; done; done
unset DOA_SCAN_ABLATE
echo "-- other batches"
for B in 16384 32768 65536 131072 524288; do
    for c in 0 1; do export B DOA_SCAN_PAIR=$c; echo "batch $B pair $c | $(run)"; done
done
