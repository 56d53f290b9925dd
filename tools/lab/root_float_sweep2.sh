#!/bin/bash
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for cfg in "0 0 0" "1 12 33" "1 16 33" "1 20 33" "1 30 33" "1 16 26" "1 20 26" "1 30 26" "1 30 40" "1 20 40"; do
    set -- $cfg
    export DOA_ROOT_FLOAT_PHASE=$1 DOA_ROOT_FLOAT_ITERS=$2 DOA_ROOT_FLOAT_TOL_LOG2=$3
    for w in "cfg3 --M 2 --stages cov,root --reps 40" "n8 --N 8 --M 2 --stages cov,root --reps 20" "n16 --N 16 --M 3 --stages cov,root --reps 10"; do
        set -- $w; name=$1; shift
        echo "phase $DOA_ROOT_FLOAT_PHASE iters $DOA_ROOT_FLOAT_ITERS tol 2^-$DOA_ROOT_FLOAT_TOL_LOG2 $name: $(python3 tools/bench_kernels.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("root_us", [round(x,2) for x in d["root_us"]])')"
    done
done
