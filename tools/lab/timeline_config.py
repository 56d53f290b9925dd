"""lab (round 4): kernel timeline (rocprofv3 --kernel-trace) of one detached work_dev_batches call of a secondary config:
which kernels run beside which, per lane.  usage (under rocprofv3): python3 tools/lab/timeline_config.py cfg4|cfg3|flowgraph [lanes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, nbuf, steps = 4096, 4, 12
st = torch.cuda.current_stream()
if which == "cfg4":
    N, K, ovl, fb, d, M, P = 16, 1024, 0, 0, 0.5, 3, 4096
elif which == "cfg3":
    N, K, ovl, fb, d, M, P = 4, 1024, 0, 0, 0.44, 2, 0
else:
    N, K, ovl, fb, d, M, P = 4, 2048, 512, 1, 0.4, 2, 1024
bufs = []
for b in range(nbuf):
    if ovl == 0:
        s, _ = doa.sim.make_batch_streams_torch(N, K, B, d, M, 20.0, seed=500 + b, device="cuda")
    else:
        span = (B - 1) * (K - ovl) + K
        s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
        doa.sim_source(N, d, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600 + b).work_dev(span, [t.data_ptr() for t in s], st)
    bufs.append(s)
ptrs = [[t.data_ptr() for t in s] for s in bufs]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
idx = [i % nbuf for i in range(steps)]
if which == "cfg3":
    ang = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    pipe = doa.root_pipeline(N, K, ovl, fb, d, M, B)
    pipe.set_lanes(lanes)
    call = pipe.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx], None, doa.DETACHED)
else:
    spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, B)
    pipe.set_lanes(lanes)
    call = pipe.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                                [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], doa.DETACHED)
for _ in range(3):
    call(); pipe.synchronize()
torch.cuda.synchronize()
print("MARK")
call(); pipe.synchronize()
torch.cuda.synchronize()
