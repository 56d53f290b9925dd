#!/bin/bash
# timed-region overhead of bench.py: T(K) for several K and warm-up lengths (same box)
for rep in 1 2; do
for w in 5 500; do for k in 20 40 80 160 500; do
  timeout -k 5 200 python bench.py --steps $k --warmup $w --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warmup $w steps $k: us/step', round(d['ms_per_step']*1e3,2), 'total us', round(d['ms_per_step']*1e3*$k,1))
"
done; done; done
