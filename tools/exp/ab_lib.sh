#!/bin/bash
# same-box A/B of library variants through the current bench.py (DOA_HIP_LIB selects the .so)
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v['us'],1) for k,v in d['kernels'].items()})
"; }
for rep in 1 2; do
  for s in 3 4; do
  timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams $s --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "current      streams=$s"
  DOA_HIP_LIB=$PWD/_abscan/gr-doa_amd/lib/libdoa_hip.so timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams $s --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "old-scan lib streams=$s"
  done
  (cd _r01cmp && timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null) | show "r01 tree (4 streams)  "
done
