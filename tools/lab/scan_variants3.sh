#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
for v in - 1031 1287 3335 2311; do
  if [ "$v" = "-" ]; then unset DOA_SCAN_VARIANT; else export DOA_SCAN_VARIANT=$v; fi
  timeout -k 5 120 python tools/lab/scan_check.py 2>/dev/null | tail -1
done
unset DOA_SCAN_VARIANT
for rep in 1 2; do
for v in - 7 519 775 1031 1287 2311 3335 3351 1047; do
  for wpc in 12 16; do
    if [ "$v" = "-" ]; then unset DOA_SCAN_VARIANT; else export DOA_SCAN_VARIANT=$v; fi
    export DOA_SCAN_LEAN_WAVES_PER_CU=$wpc
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1)
    echo "variant $v wpc $wpc | $a"
  done
done
done
