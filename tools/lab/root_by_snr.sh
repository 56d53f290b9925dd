#!/bin/bash
# lab (round 4): Root-MUSIC kernel time by SNR of the data (random directions per snapshot), rocprofv3 averages, us per 4096 items
export TMPDIR=/tmp
prof() {
    name=$1; shift
    d=gpurun_out/r04/root_snr_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "root_music" in r["Name"]:
        print(f'{float(r["AverageNs"])/1e3:8.2f}', end=" ")
PY
}
for shape in "4 2" "4 3" "8 2" "16 3"; do
    set -- $shape
    echo -n "N=$1 M=$2 at 30 / 20 / 10 / 5 / 0 dB: "
    for snr in 30 20 10 5 0; do prof n$1m$2_$snr --N $1 --M $2 --snr $snr --stages rootpipe --reps 12; done
    echo
done
