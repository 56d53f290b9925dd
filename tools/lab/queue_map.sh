#!/bin/bash
# lab (round 4): hardware queue per lane, sim_source handle alive / released at lane creation
export TMPDIR=/tmp
for mode in alive freed; do
    out=gpurun_out/r04/queue_map_$mode; rm -rf $out
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/lab/queue_map.py $mode 2>/dev/null | grep "us/step" | sed "s|$| (under the profiler)|"
    f=$(ls $out/*/*kernel_trace.csv | head -n 1)
    python3 - "$f" <<'P'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "doa::" in r["Kernel_Name"] and "sim_source" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-160:]                                  # the last call: 40 batches x 4 kernels
q = collections.Counter(r["Queue_Id"] for r in rows)
print("   dispatches per Queue_Id:", dict(q))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
print("   GPU span of the call: %.1f us = %.2f per batch" % ((t1 - t0) / 1e3, (t1 - t0) / 1e3 / 40))
for name in ("cov_piece", "cov_combine", "music_evd", "music_scan"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    print("   %-12s mean %.1f us" % (name, sum(d) / len(d)))
P
done
