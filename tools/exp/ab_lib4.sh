#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v['us'],1) for k,v in d['kernels'].items()})
"; }
for rep in 1 2 3 4; do
  timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "EVD 128 VGPR (spills)   "
  DOA_HIP_LIB=$PWD/_abB/gr-doa_amd/lib/libdoa_hip.so timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "EVD 164 VGPR            "
  DOA_HIP_LIB=$PWD/_abevd/gr-doa_amd/lib/libdoa_hip.so timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "old scan + old EVD (136)"
done
