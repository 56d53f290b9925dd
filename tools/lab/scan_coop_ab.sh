#!/bin/bash
# lab (round 4): same-box A/B of the lean scan kernel's workgroup-cooperative form (lab build of the library)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2 3; do
    for cfg in "0 8 16" "1 8 16" "1 16 16" "1 4 16" "1 8 8" "1 16 8" "1 32 16"; do
        set -- $cfg
        export DOA_SCAN_COOP=$1 DOA_SCAN_COOP_STRIDE=$2 DOA_SCAN_COOP_WPB=$3
        echo "coop $1 stride $2 waves/wg $3 | $(run)"
    done
done
export DOA_SCAN_COOP_STRIDE=8 DOA_SCAN_COOP_WPB=16
echo "-- ablations, coop on (1 = no row stores, 2 = row stores only)"
for abl in 1 2; do DOA_SCAN_COOP=1 DOA_SCAN_ABLATE=$abl; export DOA_SCAN_COOP DOA_SCAN_ABLATE; echo "ablate $abl coop 1 | $(run)"; done
unset DOA_SCAN_ABLATE
echo "-- other batches and num_max_vals = 2"
for B in 16384 32768 65536 131072 524288; do
    for c in 0 1; do export B DOA_SCAN_COOP=$c; echo "batch $B coop $c | $(run)"; done
done
B=262144
for c in 0 1; do export B M=2 DOA_SCAN_COOP=$c; echo "M=2 coop $c | $(run)"; done
