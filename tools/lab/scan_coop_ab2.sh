#!/bin/bash
# lab (round 4): the cooperative form by workgroup size and row stride, real kernel (lab build)
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
run() { timeout -k 5 200 python tools/profile_scan.py --batch ${B:-262144} --reps 20 --M ${M:-1} 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2; do
    export DOA_SCAN_COOP=0; echo "shipped (coop 0) | $(run)"
    for wpb in 4 8 16; do for st in 8 16 32; do
        export DOA_SCAN_COOP=1 DOA_SCAN_COOP_STRIDE=$st DOA_SCAN_COOP_WPB=$wpb
        echo "coop stride $st waves/wg $wpb | $(run)"
    done; done
done
