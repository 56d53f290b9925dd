"""Wall time of a 20-step region against its GPU span (events): how much of the driver-style figure is host latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
N, K, P, M, B, S = 4, 1024, 1024, 1, 4096, 4
pipes = [doa.music_pipeline(N, K, 0, 0, 0.5, M, P, B) for _ in range(S)]
sts = [torch.cuda.Stream() for _ in range(S)]
nbuf = 6
ins = [doa.sim.make_batch_streams_torch(N, K, B, 0.5, M, 20.0, seed=b)[0] for b in range(nbuf)]
ptrs = [[t.data_ptr() for t in s] for s in ins]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
def step(i):
    b, k = i % nbuf, i % S
    pipes[k].work_dev(B, ptrs[b], cov[b].data_ptr(), spec[b].data_ptr(), mx[b].data_ptr(), am[b].data_ptr(), sts[k])
for i in range(12): step(i)
torch.cuda.synchronize()
for steps in (20, 20, 20, 100, 20):
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(sts[0])
    for i in range(steps):
        step(i)
    th = time.perf_counter()
    for k in range(S): ends[k].record(sts[k])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    span = max(e0.elapsed_time(e) for e in ends) * 1e3
    # a trivial region: one empty-ish sync
    torch.cuda.synchronize(); ts = time.perf_counter(); torch.cuda.synchronize(); sync_idle = (time.perf_counter() - ts) * 1e6
    print(f"steps {steps}: wall {1e6*(t1-t0):.1f} us, host enqueue {1e6*(th-t0):.1f} us, GPU span {span:.1f} us, wall-span {1e6*(t1-t0)-span:.1f} us, idle sync {sync_idle:.1f} us")
