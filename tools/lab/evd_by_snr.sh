#!/bin/bash
# lab (round 4): eigen-stage kernel time by SNR of the data (random directions per snapshot), rocprofv3 kernel averages, us per 4096 items
export TMPDIR=/tmp
prof() {
    name=$1; shift
    d=gpurun_out/r04/evd_snr_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "evd" in r["Name"]:
        print(f'      {r["Name"][10:52]:42s} {float(r["AverageNs"])/1e3:8.2f} us')
PY
}
for shape in "4 1 1024" "4 2 1024" "8 2 1024" "16 3 4096"; do
    set -- $shape
    for snr in 20 10 5 0; do
        echo "== N=$1 M=$2, $snr dB"; prof n$1m$2_$snr --N $1 --M $2 --P $3 --snr $snr --stages pipe --reps 20
    done
done
