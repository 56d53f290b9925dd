#!/bin/bash
# lab (round 4): the float phase's two knobs (iteration cap, squared relative step at which it is left: 2^-x)
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for cfg in "0 0 0" "1 12 33" "1 8 33" "1 6 33" "1 4 33" "1 12 20" "1 12 14" "1 8 20" "1 8 14" "1 6 20" "1 6 14" "1 5 10" "1 3 10"; do
    set -- $cfg
    export DOA_ROOT_FLOAT_PHASE=$1 DOA_ROOT_FLOAT_ITERS=$2 DOA_ROOT_FLOAT_TOL_LOG2=$3
    for w in "cfg3 --M 2 --stages cov,root --reps 40" "n8 --N 8 --M 2 --stages cov,root --reps 20"; do
        set -- $w; name=$1; shift
        echo "phase $DOA_ROOT_FLOAT_PHASE iters $DOA_ROOT_FLOAT_ITERS tol 2^-$DOA_ROOT_FLOAT_TOL_LOG2 $name: $(python3 tools/bench_kernels.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("root_us", [round(x,2) for x in d["root_us"]])')"
    done
done
