#!/bin/bash
# lab (round 4): num_targets = N - 1 on N <= 4: the one-noise-vector inverse iteration (evd_small_noise_vector) against what ran
# before -- the one-lane Jacobi for (4, 3), the four-lane subspace iteration for (3, 2) -- kernel averages by rocprofv3 on
# random-direction data at 20 dB and on a fixed scenario, fall-back counts; and the benchmark's own eigen stage (M = 1), which
# shares the kernel
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
prof() {  # name, bench_kernels args...
    name=$1; shift
    d=gpurun_out/r04/evd_nv_$name; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
    f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "evd" in r["Name"]:
        print(f'   {r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
}
for nv in 1 0; do
    export DOA_EVD_NOISE_VECTOR=$nv
    echo "== DOA_EVD_NOISE_VECTOR=$nv: N=4 M=3"; prof n4m3_$nv --N 4 --M 3 --stages pipe --reps 40
    echo "== DOA_EVD_NOISE_VECTOR=$nv DOA_EVD_QUAD=0: N=3 M=2"; DOA_EVD_QUAD=0 prof n3m2_$nv --N 3 --M 2 --stages pipe --reps 40
    echo "== DOA_EVD_NOISE_VECTOR=$nv: N=2 M=1 and N=4 M=1 (the signal-subspace form, unchanged code path)"; prof n2m1_$nv --N 2 --M 1 --stages pipe --reps 40; prof n4m1_$nv --N 4 --M 1 --stages pipe --reps 40
done
unset DOA_EVD_NOISE_VECTOR
echo "== N=3 M=2 as shipped (four lanes per item)"; prof n3m2_quad --N 3 --M 2 --stages pipe --reps 40
python3 tools/lab/evd_noise_vector_diag.py
