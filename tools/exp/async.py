"""Experiment: two-stream (async) pipeline mode vs several caller streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa, numpy as np
N, K, P, M, B = 4, 1024, 1024, 1, 4096
nbuf = 6
streams = [doa.sim.make_batch_streams_torch(N, K, B, 0.5, M, 20.0, seed=b) for b in range(nbuf)]
ptrs = [[t.data_ptr() for t in s[0]] for s in streams]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]

def make(mode, ns):
    pipes = [doa.music_pipeline(N, K, 0, 0, 0.5, M, P, B) for _ in range(ns)]
    hs = [torch.cuda.Stream() for _ in range(ns)]
    if mode == "async":
        for p in pipes: p.set_async(True)
    def step(i):
        b, k = i % nbuf, i % ns
        pipes[k].work_dev(B, ptrs[b], cov[b].data_ptr(), spec[b].data_ptr(), mx[b].data_ptr(), am[b].data_ptr(), hs[k])
    def join():
        if mode == "async":
            for p, h in zip(pipes, hs): p.join(h)
        torch.cuda.synchronize()
    return step, join, pipes

def region(step, join, n):
    join(); t0 = time.perf_counter()
    for i in range(n): step(i)
    join()
    return (time.perf_counter() - t0) / n * 1e6

for mode, ns in (("multi", 4), ("async", 1), ("async", 2), ("multi", 3), ("async", 1), ("multi", 4), ("async", 2)):
    step, join, pipes = make(mode, ns)
    for i in range(12): step(i)
    join()
    r = [round(region(step, join, n), 2) for n in (20, 20, 300, 20, 300, 20)]
    # correctness of the last batch
    est = am[(300 - 1) % nbuf].cpu().numpy()[:, 0]; th = streams[(300 - 1) % nbuf][1][:, 0]
    print(f"{mode:6s} streams={ns}: us/step for regions (20,20,300,20,300,20) = {r}   max angle err {np.abs(est - th).max():.3f}", flush=True)
    del pipes
