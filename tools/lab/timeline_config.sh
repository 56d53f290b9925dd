#!/bin/bash
# lab (round 4): timelines of the secondary configs' overlapped step (last of four identical calls of 12 batches)
export TMPDIR=/tmp
for cfg in "cfg4 4" "cfg3 4" "cfg3 6" "flowgraph 4"; do
    set -- $cfg
    out=gpurun_out/r04/timeline_$1_$2; rm -rf $out
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/lab/timeline_config.py $1 $2 > $out.log 2>&1
    f=$(ls $out/*/*kernel_trace.csv | head -n 1)
    echo "== $1, $2 lanes"
    python3 - "$f" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "doa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k, v in (("cov_mfma", "K1m"), ("cov_wave", "K1"), ("cov_piece", "K1p"), ("cov_combine", "K1c"), ("music_evd", "EVD"), ("music_scan", "SCAN"), ("root_music", "ROOT"), ("find_local_max", "PEAK")):
        if k in n: return v
    return n[:16]
per = {"cfg4": 3}
# the last call: the last 12 batches' kernels
kinds = sorted(set(short(r["Kernel_Name"]) for r in rows))
n_per_batch = len([k for k in kinds if k not in ("sim_source_kernel",) and not k.startswith("doa::sim")])
last = rows[-12 * n_per_batch:]
t0 = int(last[0]["Start_Timestamp"])
span = (max(int(r["End_Timestamp"]) for r in last) - t0) / 1e3
print("kernels per batch:", n_per_batch, "| dispatches:", len(last), "| span us:", round(span, 1), "| per batch:", round(span / 12, 1))
for r in last:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{short(r['Kernel_Name']):5s} q{r.get('Queue_Id','?'):>3s} start {s:8.1f} end {e:8.1f} dur {e-s:6.1f}")
P
done
