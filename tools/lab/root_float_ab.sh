#!/bin/bash
# lab (round 4): Root-MUSIC kernel with / without the float phase (lab build), kernel averages by rocprofv3 + accuracy
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for q in 0 1; do
    export DOA_ROOT_FLOAT_PHASE=$q
    for cfg in "cfg3 --M 2 --stages rootpipe --reps 40" "n8 --N 8 --M 2 --stages cov,root --reps 20" "n16 --N 16 --M 3 --stages cov,root --reps 10"; do
        set -- $cfg; name=$1; shift
        d=gpurun_out/r04/root_float_${name}_$q; rm -rf $d
        rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
        f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
        echo "== float phase $q $name"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "root_music" in r["Name"]: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
    done
done
