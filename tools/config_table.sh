#!/bin/bash
# Regenerates the numbers of DESIGN.md section 6's "other configs" table (one JSON line per config and mode)
# into gpurun_out/<tag>/config_table.jsonl.     usage (on the GPU box): bash tools/config_table.sh r02
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
: > $out/config_table.jsonl
run() { echo "# $*" >> $out/config_table.jsonl; timeout -k 10 300 python3 tools/bench_kernels.py "$@" >> $out/config_table.jsonl; }
run --M 1 --stages cov,music,peak,pipe
run --M 1 --stages mpipe --streams 4
run --M 2 --stages cov,music,peak,root,pipe
run --M 2 --stages mroot --streams 4
run --M 2 --K 2048 --stages cov,music,peak,pipe
run --M 2 --K 2048 --stages mpipe --streams 4
run --M 2 --K 2048 --ovl 512 --fb 1 --stages cov,music,peak,pipe
run --M 2 --K 2048 --ovl 512 --fb 1 --stages mpipe --streams 4
run --N 8 --M 2 --stages cov,music,peak,root,pipe
run --N 8 --M 2 --stages mpipe --streams 4
run --N 16 --M 3 --P 4096 --stages cov,music,peak,root,pipe --reps 20
run --N 16 --M 3 --P 4096 --stages mpipe --streams 4 --reps 20
echo done
