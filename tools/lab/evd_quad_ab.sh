#!/bin/bash
# lab (round 4): N <= 4, M >= 2 eigen stage: four lanes per item (music_evd_quad_kernel) against the one-lane Jacobi, kernel
# averages by rocprofv3 for the flowgraph's shape and configs[2]'s, lab build (DOA_EVD_QUAD = 0 / 1)
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for q in 0 1; do
    export DOA_EVD_QUAD=$q
    for cfg in "flowgraph --M 2 --K 2048 --ovl 512 --fb 1 --stages pipe --reps 40" "cfg3 --M 2 --stages cov,root --reps 40" "m3 --M 3 --stages pipe --reps 40"; do
        set -- $cfg; name=$1; shift
        d=gpurun_out/r04/evd_quad_${name}_$q
        rm -rf $d
        rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
        f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
        echo "== quad $q $name"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "evd" in n or "root_music" in n or "cov_" in n or "scan" in n:
        print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
    done
done
