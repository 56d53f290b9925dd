#!/bin/bash
# lab (round 4): long-spectrum scan kernel, one wave per item against two waves per item (lab build, DOA_SCAN_LONG2 = 0 / 1)
export TMPDIR=/tmp DOA_HIP_LIB=$PWD/_ab/libdoa_hip_lab.so
for q in 0 1; do
    export DOA_SCAN_LONG2=$q
    for cfg in "cfg4 --N 16 --M 3 --P 4096 --stages pipe --reps 20" "n8p4096 --N 8 --M 2 --P 4096 --stages pipe --reps 20" "n16p2112 --N 16 --M 3 --P 2112 --stages pipe --reps 20" "block --N 16 --M 3 --P 4096 --stages music --reps 20"; do
        set -- $cfg; name=$1; shift
        d=gpurun_out/r04/long2_${name}_$q; rm -rf $d
        rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_kernels.py "$@" > $d.log 2>&1
        f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1)
        echo "== long2 $q $name"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "scan" in r["Name"]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
    done
done
unset DOA_SCAN_LONG2
for q in 0 1; do DOA_SCAN_LONG2=$q python3 - <<'PY'
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gr-doa_amd", "python")]
import torch, doa, bench
out = bench.other_configs(doa, torch, torch.cuda.current_stream(), lanes=4, check=False)
v = out["cfg4_n16"]
print("long2", os.environ["DOA_SCAN_LONG2"], "cfg4 step: serial", round(v["us_per_step_serial"], 1), "overlapped", round(v["us_per_step_overlapped"], 1))
PY
done
