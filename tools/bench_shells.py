"""Host-fed rate of the C++ block shells (not the graded bench): N stream files ->
gr::doa::music_pipeline (or the three chained blocks) under the gnuradio_lite scheduler, wall time inside the
blocks' work() calls only (host buffers in, host buffers out).
usage: python tools/bench_shells.py [--snapshots 16384] [--multiple 256 64 16 1]"""
import argparse, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
ap = argparse.ArgumentParser()
ap.add_argument("--snapshots", type=int, default=16384)
ap.add_argument("--multiple", type=int, nargs="*", default=[1024, 256, 64, 16, 1])
a = ap.parse_args()
N, K, M, P, d = 4, 1024, 1, 1024, 0.5
from doa import sim
x, _ = sim.make_batch_streams(N, K, a.snapshots, d, M, 20.0, seed=5)
exe = os.path.join(ROOT, "gr-doa_amd", "lib", "run_flowgraph")
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    pre = os.path.join(tmp, "in")
    for k in range(N):
        x[k].tofile(f"{pre}.ch{k}.c64")
    for mode in ("pipeline", "music"):
        for mult in a.multiple:
            env = dict(os.environ, DOA_GR_OUTPUT_MULTIPLE=str(mult), DOA_GR_MIN_OUTPUT_BUFFER=str(max(512, 2 * mult)))
            r = subprocess.run([exe, mode, pre, os.path.join(tmp, "out"), str(N), str(K), "0", "0", str(d), str(M), str(P), "8"],
                               capture_output=True, text=True, env=env, timeout=600)
            lines = [l for l in r.stdout.splitlines() if "snapshots_per_s" in l]
            print(f"{mode:9s} output_multiple={mult:5d}:", " | ".join(lines) if r.returncode == 0 else r.stderr[-300:], flush=True)
