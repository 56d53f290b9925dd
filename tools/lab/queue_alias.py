"""lab (round 4): does the flowgraph-shape call slow down when other handles (with lane streams of their own) exist in the process?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
B, nbuf, N = 4096, 8, 4
st = torch.cuda.current_stream()
span = (B - 1) * 1536 + 2048
bufs = []
for b in range(nbuf):
    s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
    doa.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600 + b).work_dev(span, [t.data_ptr() for t in s], st)
    bufs.append(s)
ptrs = [[t.data_ptr() for t in s] for s in bufs]
cov = [torch.empty((B, 16), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
ang = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
spec = [torch.empty((B, 1024), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
c3 = [doa.sim.make_batch_streams_torch(N, 1024, B, 0.44, 2, 20.0, seed=400 + b, device="cuda")[0] for b in range(nbuf)]
c3p = [[t.data_ptr() for t in s] for s in c3]
steps = 100
idx = [i % nbuf for i in range(steps)]
def flow_pipe():
    p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, B); p.set_lanes(4)
    return p, p.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                                [mx[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx], doa.DETACHED)
def root_pipe():
    p = doa.root_pipeline(4, 1024, 0, 0, 0.44, 2, B); p.set_lanes(4)
    return p, p.prepare_batches(B, [c3p[b] for b in idx], [cov[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx], None, doa.DETACHED)
def measure(call, tag):
    call(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); call(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    print(f"{tag}: {best:6.2f} us/step", flush=True)
keep = []
p, c = flow_pipe(); keep.append(p); measure(c, "flow, first handle of the process")
first_flow = c
for k in range(1, 5):
    q, cq = root_pipe(); keep.append(q); measure(cq, f"root handle #{k}")
    p, c = flow_pipe(); keep.append(p); measure(c, f"flow, new handle after {2 * k} others")
    measure(first_flow, "flow, the first handle again")
# serial work_dev calls on the torch stream before the lanes call (what bench.py does)
p, c = flow_pipe(); keep.append(p)
for i in range(20):
    p.work_dev(B, ptrs[i % nbuf], cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(), ang[i % nbuf].data_ptr(), st)
torch.cuda.synchronize()
measure(c, "flow, new handle after 20 serial work_dev calls on it")
# a handle destroyed (its four lane streams with it) before the next one is created: what bench.py's other_configs does
import gc
print("-- now with handles destroyed in between", flush=True)
for k in range(3):
    q, cq = root_pipe(); measure(cq, f"root handle (to be destroyed) #{k}")
    del q, cq; gc.collect()
    p, c = flow_pipe(); measure(c, f"flow, new handle after {k + 1} destroyed")
    del p, c; gc.collect()
