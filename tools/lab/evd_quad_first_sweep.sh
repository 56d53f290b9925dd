#!/bin/bash
# lab (round 4): the four-lane eigen stage's first-check rule -- "more than F steps to go: take the Jacobi now" -- by F, on four kinds of
# data (kernel averages by rocprofv3, us per 4096 items; the one-lane Jacobi reads 11.1-11.2 on all of them)
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
prof() {
    name=$1; shift
    d=gpurun_out/r04/evd_qf_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "evd" in r["Name"]:
        print(f'      {float(r["AverageNs"])/1e3:8.2f} us')
PY
}
for F in 3 4 6 8 11; do
    export DOA_EVD_QUAD_FIRST=$F
    echo "== F = $F"
    echo "   random directions 20 dB:"; prof r20_$F --N 4 --M 2 --snr 20 --stages pipe --reps 30
    echo "   random directions 10 dB:"; prof r10_$F --N 4 --M 2 --snr 10 --stages pipe --reps 30
    echo "   random directions 5 dB:";  prof r5_$F --N 4 --M 2 --snr 5 --stages pipe --reps 30
    echo "   flowgraph shape (sim_source, forward-backward):"; prof fg_$F --M 2 --K 2048 --ovl 512 --fb 1 --stages pipe --reps 30
done
