#!/bin/bash
# A/B of bench configurations inside one box: each variant twice, interleaved
for rep in 1 2; do
for v in "3 12" "4 12" "3 8" "4 8" "2 12"; do
  set -- $v
  r=$(DOA_SCAN_LEAN_WAVES_PER_CU=$2 timeout -k 5 120 python bench.py --streams $1 --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2))")
  r20=$(DOA_SCAN_LEAN_WAVES_PER_CU=$2 timeout -k 5 120 python bench.py --streams $1 --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2))")
  echo "rep=$rep streams=$1 scan_wpc=$2: 300-step $r us, 20-step $r20 us"
done
done
