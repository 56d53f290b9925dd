#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step']*1e3,2))
"; }
for rep in 1 2; do
for w in 16 12 8; do for s in 3 4; do
  a=$(DOA_COV_WAVES_PER_CU=$w timeout -k 5 200 python bench.py --steps 20 --warmup 5 --streams $s --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "")
  b=$(DOA_COV_WAVES_PER_CU=$w timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams $s --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "")
  echo "rep=$rep cov_wpc=$w streams=$s: 20-step $a | 300-step $b"
done; done; done
