#!/bin/bash
# driver-style run (20 steps) under several values of one environment variable, 3 rounds; "unset" = variable absent
var=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = unset ]; then pre=""; else pre="$var=$v"; fi
  env $pre timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v steps 20: us/step', round(d['ms_per_step']*1e3,2), ' value %.3e' % d['value'])
"
done; done
