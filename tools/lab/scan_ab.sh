#!/bin/bash
# lab: same-box A/B of the lean scan kernel's launch knobs + the two ablations (library built with make LAB=1)
for rep in 1 2; do
for pf in 1 2; do
for shift in 0 4; do
  for wpc in 12 16; do
    export DOA_SCAN_ITEM_SHIFT=$shift DOA_SCAN_LEAN_WAVES_PER_CU=$wpc DOA_SCAN_PREFETCH=$pf
    unset DOA_SCAN_ABLATE
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 20 2>/dev/null | tail -1 | sed 's/scan-only launches: 20 batch 262144: //')
    echo "prefetch $pf shift $shift wpc $wpc | full: $a"
  done
done
done
done
unset DOA_SCAN_ABLATE DOA_SCAN_ITEM_SHIFT DOA_SCAN_LEAN_WAVES_PER_CU DOA_SCAN_PREFETCH
timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | tail -1
timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 30 2>/dev/null | tail -1
