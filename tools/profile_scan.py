"""Only the K4+K5 scan kernel, back to back, at a given batch (for rocprofv3 --kernel-trace / --pmc runs):
coefficient records come from one real K1 -> EVD pass over 64-sample snapshots, then the pipeline handle is told
to launch the scan stage only.  usage: python3 tools/profile_scan.py [--batch 262144] [--reps 40] [--M 1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch
import doa
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--M", type=int, default=1)
ap.add_argument("--P", type=int, default=1024)
a = ap.parse_args()
N, K, B = 4, 64, a.batch
st = torch.cuda.current_stream()
pipe = doa.music_pipeline(N, K, 0, 0, 0.5, a.M, a.P, B)
s, _ = doa.sim.make_batch_streams_torch(N, K, B, 0.5, a.M, 20.0, seed=77, device="cuda")
nb = 2 if B > 65536 else 8
cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
spec = [torch.empty((B, a.P), dtype=torch.float32, device="cuda") for _ in range(nb)]
mx = torch.empty((B, a.M), dtype=torch.float32, device="cuda")
am = torch.empty((B, a.M), dtype=torch.float32, device="cuda")
run = lambda i: pipe.work_dev(B, [t.data_ptr() for t in s], cov.data_ptr(), spec[i % nb].data_ptr(), mx.data_ptr(), am.data_ptr(), st)
run(0)
torch.cuda.synchronize()
pipe.set_stages(cov=False, evd=False, scan=True)
for i in range(10):
    run(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(st)
for i in range(a.reps):
    run(i)
e1.record(st)
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.reps
nbytes = (2 * N * 8 + a.P * 4 + 8 * a.M) * B
print(f"scan-only launches: {a.reps} batch {B}: {us:.2f} us/launch, {nbytes / us / 1e3:.0f} GB/s algorithmic = {nbytes / us / 8e6:.3f} of 8 TB/s")
