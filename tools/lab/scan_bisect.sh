#!/bin/bash
# lab (round 4): bisecting the real pair kernel toward the stand-alone model (variants built with -DDOA_SCAN_BISECT=bits:
# 1 no tie logic, 2 peak records before the burst, 4 no peak records, 16 no IEEE division in LeanNorm)
run() { timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 //'; }
for rep in 1 2; do
  for v in lab b1 b2 b4 b7 b16 b23; do
    export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_$v.so
    for p in 1 0; do export DOA_SCAN_PAIR=$p; echo "variant $v pair $p | $(run)"; done
  done
done
