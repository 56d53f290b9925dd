#!/bin/bash
# lab (round 4): the wave-per-item subspace eigen stage (N = 8 / 16) at low SNR: step limit and early bail-out of hopeless items;
# rocprofv3 kernel averages, us per 4096 items (random directions per snapshot)
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
prof() {
    name=$1; shift
    d=gpurun_out/r04/evd_wide_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "evd" in r["Name"]:
        print(f'{float(r["AverageNs"])/1e3:8.2f}', end=" ")
PY
}
for cfg in "20 0" "20 1" "32 1" "48 1" "64 1"; do
    set -- $cfg
    export DOA_EVD_WIDE_MAX_STEPS=$1 DOA_EVD_WIDE_BAIL=$2
    for shape in "8 2 1024" "16 3 4096"; do
        set -- $shape
        echo -n "steps $DOA_EVD_WIDE_MAX_STEPS bail $DOA_EVD_WIDE_BAIL  N=$1 M=$2  at 20 / 10 / 5 / 0 dB: "
        for snr in 20 10 5 0; do prof n$1_${snr}_${DOA_EVD_WIDE_MAX_STEPS}_$DOA_EVD_WIDE_BAIL --N $1 --M $2 --P $3 --snr $snr --stages pipe --reps 12; done
        echo
    done
done
