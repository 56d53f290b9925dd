#!/bin/bash
# same-box A/B: round-1 tree (_r01cmp/) against the current one
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sr=d.get('scan_kernel_roofline') or d.get('roofline_scan',{}).get('at_large_batch',{})
sb=d.get('roofline_scan',{}).get('at_benchmark_batch',{})
print('$1', 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v['us'],1) for k,v in d['kernels'].items()}, 'scan262144', round(sr.get('avg_launch_us',0),1), 'scan4096', round(sb.get('avg_launch_us',0),2))
"; }
for rep in 1 2; do
  (cd _r01cmp && timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null) | show "r01(4 streams)"
  timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | show "r02(3 streams)"
  timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams 4 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "r02(4 streams)"
  (cd _r01cmp && timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null) | show "r01 20-step"
  timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "r02 20-step"
done
