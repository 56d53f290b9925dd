#!/bin/bash
var=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  env $var=$v timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v bench300 us/step', round(d['ms_per_step']*1e3,2))
"
done; done
