"""Copies the judged rocprofv3 summaries of one round from gpurun_out/prof_<tag>/ (scratch, written on the GPU
box by tools/profile_round.sh) into profiles/ (tracked).   usage: python tools/summarize_profiles.py r02

  profiles/<tag>_kernel_stats_<run>.csv     rocprofv3 --kernel-trace --stats rows of this library's kernels, one
                                            file per profiled command; first line = "# command: ..."
  profiles/<tag>_pmc_hbm_traffic.csv        per (kernel, batch): FETCH_SIZE / WRITE_SIZE means from the separate --pmc
                                            passes, corrected as MI355X_MICROARCH.md (section HBM) prescribes
                                            (FETCH_SIZE x 2 on gfx950 for 16-B-per-lane streaming reads, WRITE_SIZE exact,
                                            both in KiB), HBM bytes per launch, the algorithmic bytes next to them,
                                            and the command line of the pass
  profiles/<tag>_bench_under_rocprof_<run>.json   the bench line printed under the profiler (profiled clocks)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

commands = {}
cmd_file = os.path.join(src, "commands.txt")
if os.path.exists(cmd_file):
    for line in open(cmd_file):
        m = re.match(r"(\w+): (.*rocprofv3 .*)", line.strip())
        if m:
            commands[m.group(1)] = m.group(2)


def keep(name):
    return "doa::" in name


def find_csv(run, suffix):
    fs = glob.glob(os.path.join(src, run, "**", f"*{suffix}"), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None          # (a re-run leaves the older pid's files next to the new ones)


# ---- kernel-trace stats ---------------------------------------------------------------------------------
for run in sorted(commands):
    if not run.startswith("trace_"):
        continue
    f = find_csv(run, "kernel_stats.csv")
    if not f:
        print("no kernel stats for", run)
        continue
    rows = [r for r in csv.DictReader(open(f)) if keep(r["Name"])]
    out = os.path.join(dst, f"{tag}_kernel_stats_{run[len('trace_'):]}.csv")
    with open(out, "w", newline="") as fo:
        fo.write(f"# command: {commands[run]}\n")
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    log = os.path.join(src, run + ".log")
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith("{") and '"metric"' in line:
                open(os.path.join(dst, f"{tag}_bench_under_rocprof_{run[len('trace_'):]}.json"), "w").write(line)
    print("wrote", out)

# ---- PMC: HBM traffic per launch ---------------------------------------------------------------------------
# what each profiled workload launches, so that a row can state its batch and its algorithmic bytes
N, K, P = 4, 1024, 1024
WORKLOADS = {
    "bench4096": {"batch": 4096, "alg": {"cov_wave_kernel": 4096 * (N * K * 8 + N * N * 8),
                                           "music_evd_kernel": 4096 * (N * N * 8 + 2 * N * 8),
                                           "music_scan_peak1_kernel": 4096 * (2 * N * 8 + P * 4 + 8)}},
    "scan262144": {"batch": 262144, "alg": {"music_scan_peak1_kernel": 262144 * (2 * N * 8 + P * 4 + 8)}},
    "cfg4": {"batch": 4096, "alg": {"cov_mfma_kernel": 4096 * (16 * 1024 * 8 + 256 * 8),
                                      "music_evd_block16_kernel": 4096 * (256 * 8 + 32 * 8),
                                      "music_evd_subspace_kernel": 4096 * (256 * 8 + 32 * 8),
                                      "music_scan_stream_kernel": 4096 * (32 * 8 + 4096 * 4),
                                      "music_scan_peak_long_kernel": 4096 * (32 * 8 + 4096 * 4 + 24),
                                      "find_local_max_stream_kernel": 4096 * (4096 * 4 + 24)}},
    "k5long": {"batch": 4096, "alg": {"find_local_max_blocked_kernel": 4096 * (4096 * 4 + 24),
                                        "music_scan_stream_kernel": 4096 * (32 * 8 + 4096 * 4),
                                        "music_evd_subspace_kernel": 4096 * (256 * 8 + 32 * 8)}},
}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))       # (workload, kernel) -> counter -> values
for run in sorted(commands):
    m = re.match(r"pmc_(FETCH_SIZE|WRITE_SIZE)_(\w+)", run)
    if not m:
        continue
    f = find_csv(run, "counter_collection.csv")
    if not f:
        print("no counters for", run)
        continue
    for r in csv.DictReader(open(f)):
        if keep(r["Kernel_Name"]) and r["Counter_Name"] == m.group(1):
            pmc[(m.group(2), r["Kernel_Name"].split("(")[0])][m.group(1)].append(float(r["Counter_Value"]))
if pmc:
    out = os.path.join(dst, f"{tag}_pmc_hbm_traffic.csv")
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "batch", "dispatches", "FETCH_SIZE_KiB_raw_mean", "FETCH_bytes_corrected_x2", "WRITE_SIZE_KiB_mean",
                    "WRITE_bytes", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "command"])
        for (wl, k), c in sorted(pmc.items()):
            # the workload's own launches only: a one-off set-up launch of another size would skew the mean, so
            # outliers further than 2x from the median are dropped
            def mean(v):
                if not v:
                    return float("nan")
                med = sorted(v)[len(v) // 2]
                v = [x for x in v if med / 2 <= x <= med * 2] or v
                return sum(v) / len(v)
            fe, wr = mean(c["FETCH_SIZE"]), mean(c["WRITE_SIZE"])
            fb, wb = fe * 1024 * 2, wr * 1024
            alg = next((v for kk, v in WORKLOADS.get(wl, {}).get("alg", {}).items() if kk in k), "")
            cmd = commands.get(f"pmc_FETCH_SIZE_{wl}", "").replace("--pmc FETCH_SIZE", "--pmc FETCH_SIZE|WRITE_SIZE (two passes)")
            cmd = re.sub(r"pmc_FETCH_SIZE_", "pmc_<counter>_", cmd)
            w.writerow([k, WORKLOADS.get(wl, {}).get("batch", ""), max(len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"])), f"{fe:.1f}",
                        f"{fb:.0f}", f"{wr:.1f}", f"{wb:.0f}", f"{fb + wb:.0f}", alg, cmd])
    print("wrote", out)
