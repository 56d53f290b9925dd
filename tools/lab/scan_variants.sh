#!/bin/bash
# lab: same-box A/B of the scan-kernel variants of csrc/music_scan_lab.hpp (library built with `make LAB=1`)
out=gpurun_out/r03; mkdir -p $out
for v in - 0 1 2 4 7 15 23; do
  if [ "$v" = "-" ]; then unset DOA_SCAN_VARIANT; else export DOA_SCAN_VARIANT=$v; fi
  timeout -k 5 120 python tools/lab/scan_check.py 2>/dev/null | tail -1
done
unset DOA_SCAN_VARIANT
for rep in 1 2; do
for v in - 0 1 2 3 4 6 7 8 9 11 15 16 23 32 33 41 71 79; do
  for wpc in 12 ${EXTRA_WPC}; do
    if [ "$v" = "-" ]; then unset DOA_SCAN_VARIANT; else export DOA_SCAN_VARIANT=$v; fi
    export DOA_SCAN_LEAN_WAVES_PER_CU=$wpc
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>gpurun_out/r03/err_$v.txt | tail -1)
    b=$(timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | tail -1)
    echo "variant $v wpc $wpc | $a | $b"
    grep "\[lab\]" gpurun_out/r03/err_$v.txt | head -1
  done
done
done
