"""Experiment: why does a 20-step timed region run slower than a 500-step one?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
N, K, P, M, B = 4, 1024, 1024, 1, 4096
ns = int(os.environ.get("NS", "4")); nbuf = 6
pipes = [doa.music_pipeline(N, K, 0, 0, 0.5, M, P, B) for _ in range(ns)]
hs = [torch.cuda.Stream() for _ in range(ns)]
streams = [doa.sim.make_batch_streams_torch(N, K, B, 0.5, M, 20.0, seed=b)[0] for b in range(nbuf)]
ptrs = [[t.data_ptr() for t in s] for s in streams]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
def step(i):
    b, k = i % nbuf, i % ns
    pipes[k].work_dev(B, ptrs[b], cov[b].data_ptr(), spec[b].data_ptr(), mx[b].data_ptr(), am[b].data_ptr(), hs[k])
def region(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
for i in range(6): step(i)
torch.cuda.synchronize()
out = []
for n in (5, 20, 20, 20, 100, 20, 500, 20, 20):
    out.append((n,) + tuple(round(x, 2) for x in region(n)))
time.sleep(0.5)
out.append(("after 0.5 s idle",))
for n in (20, 20, 20):
    out.append((n,) + tuple(round(x, 2) for x in region(n)))
print("streams", ns, "(n, us/step total, us/step host enqueue):", out)
