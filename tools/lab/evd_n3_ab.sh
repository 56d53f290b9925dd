#!/bin/bash
# lab (round 4): which eigen stage for which small shape -- four lanes per item (N = 4, M = 2 only, as shipped) against the one-lane
# Jacobi (DOA_EVD_QUAD=0), random directions per snapshot at 20 and 5 dB; kernel averages by rocprofv3
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
prof() {
    name=$1; shift
    d=gpurun_out/r04/evd_n3_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "evd" in r["Name"]:
        print(f'   {r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
}
for snr in 20 5; do
    for q in 1 0; do
        export DOA_EVD_QUAD=$q
        echo "== N=4 M=2, $snr dB, DOA_EVD_QUAD=$q"; prof n4m2_${snr}_$q --N 4 --M 2 --snr $snr --stages pipe --reps 40
        echo "== N=3 M=2, $snr dB, DOA_EVD_QUAD=$q"; prof n3m2_${snr}_$q --N 3 --M 2 --snr $snr --stages pipe --reps 40
    done
done
