#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box into gpurun_out/prof_<tag>/ (scratch);
# tools/summarize_profiles.py <tag> then copies the judged summaries into profiles/ (tracked).
# Every rocprofv3 command has the program itself directly behind `--` (no env/bash hop: switches are exported first), and
# the PMC passes are separate from the kernel-trace passes.   usage (on the box):  bash tools/profile_round.sh r04
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
run() {   # name, rocprof args..., -- program...
    name=$1; shift
    echo "$name: ${PRE:-}rocprofv3 $*" >> $out/commands.txt
    rocprofv3 "$@" > $out/$name.log 2>&1 || echo "$name FAILED rc=$?" >> $out/commands.txt
}
T="--kernel-trace --stats --output-format csv"
B="--no-cpu-baseline --no-scan-roofline --no-other-configs"
BENCH_SERIAL="python3 bench.py --steps 200 --warmup 20 --streams 1 $B"
BENCH_OVL="python3 bench.py --steps 200 --warmup 20 $B"
SCAN_L="python3 tools/profile_scan.py --batch 262144 --reps 40"
SCAN_B="python3 tools/profile_scan.py --batch 4096 --reps 200"
run trace_serial  $T -d $out/trace_serial  -- $BENCH_SERIAL
run trace_overlap $T -d $out/trace_overlap -- $BENCH_OVL
run trace_driver20 $T -d $out/trace_driver20 -- python3 bench.py --steps 20 --warmup 5 $B
run trace_scan262144 $T -d $out/trace_scan262144 -- $SCAN_L
run trace_scan4096   $T -d $out/trace_scan4096   -- $SCAN_B
# the two ablations of the graded kernel (lab build of the library, results invalid: DESIGN.md section 3)
if [ -f _ab/libdoa_hip_lab.so ]; then
    export DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
    export DOA_SCAN_ABLATE=1; PRE="DOA_HIP_LIB=_ab/libdoa_hip_lab.so DOA_SCAN_ABLATE=1 "
    run trace_scan262144_ablate_no_row_stores $T -d $out/trace_scan262144_ablate_no_row_stores -- $SCAN_L
    export DOA_SCAN_ABLATE=2; PRE="DOA_HIP_LIB=_ab/libdoa_hip_lab.so DOA_SCAN_ABLATE=2 "
    run trace_scan262144_ablate_row_stores_only $T -d $out/trace_scan262144_ablate_row_stores_only -- $SCAN_L
    unset DOA_SCAN_ABLATE DOA_HIP_LIB; PRE=""
fi
for ctr in FETCH_SIZE WRITE_SIZE; do
    run pmc_${ctr}_bench4096    --pmc $ctr --output-format csv -d $out/pmc_${ctr}_bench4096    -- python3 bench.py --steps 40 --warmup 5 --streams 1 $B
    run pmc_${ctr}_scan262144   --pmc $ctr --output-format csv -d $out/pmc_${ctr}_scan262144   -- python3 tools/profile_scan.py --batch 262144 --reps 12
done
# the other BASELINE.json configs (parity-test cases, not bench lines)
run trace_cfg3_root   $T -d $out/trace_cfg3_root   -- python3 tools/bench_kernels.py --M 2 --stages rootpipe --reps 40
run trace_cfg4_n16    $T -d $out/trace_cfg4_n16    -- python3 tools/bench_kernels.py --N 16 --M 3 --P 4096 --stages pipe --reps 20
run trace_flowgraph   $T -d $out/trace_flowgraph   -- python3 tools/bench_kernels.py --M 2 --K 2048 --ovl 512 --fb 1 --stages pipe --reps 40
run trace_n8          $T -d $out/trace_n8          -- python3 tools/bench_kernels.py --N 8 --M 2 --stages pipe --reps 40
# the stand-alone blocks on cfg4's spectra (MUSIC_lin_array: EVD + spectrum-only scan; find_local_max on 4096-element vectors)
run trace_k5_long     $T -d $out/trace_k5_long     -- python3 tools/bench_kernels.py --N 16 --M 3 --P 4096 --stages music,peak --reps 40
for ctr in FETCH_SIZE WRITE_SIZE; do
    run pmc_${ctr}_cfg4 --pmc $ctr --output-format csv -d $out/pmc_${ctr}_cfg4 -- python3 tools/bench_kernels.py --N 16 --M 3 --P 4096 --stages pipe --reps 6
    run pmc_${ctr}_k5long --pmc $ctr --output-format csv -d $out/pmc_${ctr}_k5long -- python3 tools/bench_kernels.py --N 16 --M 3 --P 4096 --stages music,peak --reps 6
done
cat $out/commands.txt
