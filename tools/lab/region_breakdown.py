"""lab: where the time of a 20-step region goes -- host-measured elapsed against the GPU-side span (events on the four adopted
lane streams around the one detached batches call) and against the steady-state step of a 500-step call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch
import doa
N, K, P, M, B, d = 4, 1024, 1024, 1, 4096, 0.5
L = 4
nbuf = 8
bufs = [doa.sim.make_batch_streams_torch(N, K, B, d, M, 20.0, seed=b, device="cuda")[0] for b in range(nbuf)]
ptrs = [[t.data_ptr() for t in s] for s in bufs]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, B)
sts = [torch.cuda.Stream() for _ in range(L)]
pipe.set_lane_streams(sts)
def prep(n):
    idx = [i % nbuf for i in range(n)]
    return pipe.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                                [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], doa.DETACHED)
c8, c5, c20, c500 = prep(8), prep(5), prep(20), prep(500)
c8(); torch.cuda.synchronize()
for rep in range(6):
    c5(); torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(L): e0[l].record(sts[l])
    c20()
    for l in range(L): e1[l].record(sts[l])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # GPU-side: per lane start->end, and the span from the earliest start to the latest end (events of different streams share a clock)
    lane_ms = [e0[l].elapsed_time(e1[l]) for l in range(L)]
    span = max(e0[0].elapsed_time(e1[l]) for l in range(L)) - min(0.0, *[e0[0].elapsed_time(e0[l]) for l in range(L)])
    print(f"rep {rep}: host elapsed {t_all*1e6:7.1f} us ({t_all*1e6/20:.2f}/step), enqueue {t_enq*1e6:6.1f} us, GPU span {span*1e3:7.1f} us, per lane {[round(x*1e3,1) for x in lane_ms]}")
t0 = time.perf_counter(); c500(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"500 steps: {t*1e6/500:.2f} us/step")
