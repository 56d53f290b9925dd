#!/bin/bash
# kernel timeline of the driver-style run (20 timed steps): start/end of every dispatch
out=gpurun_out/timeline; rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scan-roofline > $out/bench.log 2>&1
f=$(ls $out/t/*/*kernel_trace.csv | head -n 1)
python3 - "$f" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "doa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return "K1" if "cov_wave" in n else "EVD" if "music_evd" in n else "SCAN" if "music_scan" in n else "PEAK" if "find_local_max" in n else n[:20]
# the timed region: 6 set-up + 5 warm-up + 20 timed steps of (K1, EVD, SCAN) = dispatches 33..92 of the pipeline kernels
pipe = [r for r in rows if short(r["Kernel_Name"]) in ("K1", "EVD", "SCAN")]
timed = pipe[33:93]
t0 = int(timed[0]["Start_Timestamp"])
print("timed dispatches:", len(timed), "span us:", (max(int(r["End_Timestamp"]) for r in timed) - t0) / 1e3)
for r in timed:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{short(r['Kernel_Name']):5s} q{r.get('Queue_Id','?'):>3s} start {s:8.1f} end {e:8.1f} dur {e-s:6.1f}")
P
