#!/bin/bash
# lab (round 4): variant 8 = the tie rule out of the conversion loop (masks first, branch-free conversions, selects last);
# 24 = 8 + no IEEE division; with bit comparison against the shipped form
run() { timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for v in lab b8; do export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_$v.so; for p in 0 1; do export DOA_SCAN_PAIR=$p; echo "variant $v pair $p | $(timeout -k 5 100 python tools/lab/scan_check.py 2>&1 | tail -1)"; done; done
for rep in 1 2 3; do
  for v in lab b1 b8 b24; do
    export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_$v.so
    for p in 1 0; do export DOA_SCAN_PAIR=$p; echo "variant $v pair $p | $(run)"; done
  done
done
