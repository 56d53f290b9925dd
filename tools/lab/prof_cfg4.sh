#!/bin/bash
# lab: per-kernel times of the cfg4 pipeline (N=16, M=3, P=4096) under rocprofv3, by ablation of the long scan kernel
# (1 = no Horner, 2 = no dB pass / row store, 4 = no peak pick; sums combine)
export TMPDIR=/tmp
export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for abl in ${ABLS:-0 1 2 4 6 7}; do
    export DOA_SCAN_LONG_ABLATE=$abl
    out=gpurun_out/prof_lab_cfg4_$abl
    rm -rf $out; mkdir -p $out
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_kernels.py --N 16 --M 3 --P 4096 --stages pipe --reps 20 > $out/run.log 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    python3 - "$f" $abl <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'doa::music_scan' in r['Name']: print(f"ablate {sys.argv[2]}: {float(r['AverageNs'])/1e3:9.2f} us  x{r['Calls']:>4}  {r['Name'][:70]}")
PY
done
