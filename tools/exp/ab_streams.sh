#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'us/step', round(d['ms_per_step']*1e3,2))
"; }
for rep in 1 2; do
for s in 2 3 4 5 6 8; do
  timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams $s --nbuf 10 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "current streams=$s"
done
DOA_HIP_LIB=$PWD/_abevd/gr-doa_amd/lib/libdoa_hip.so timeout -k 5 200 python bench.py --steps 300 --warmup 30 --streams 4 --no-cpu-baseline --no-scan-roofline 2>/dev/null | show "old kernels streams=4"
done
