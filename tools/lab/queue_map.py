"""lab (round 4): which hardware queue each lane's kernels run on (rocprofv3 --kernel-trace gives Queue_Id per dispatch), with the
last sim_source handle of the set-up kept alive (bench.py's pipeline_config before the fix) or released.  argv: alive|freed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
mode = sys.argv[1]
B, N, nbuf, steps = 4096, 4, 8, 40
st = torch.cuda.current_stream()
span = (B - 1) * 1536 + 2048
bufs = []
for b in range(nbuf):
    s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
    src = doa.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600 + b)   # the previous one dies AFTER this one exists
    src.work_dev(span, [t.data_ptr() for t in s], st)
    bufs.append(s)
if mode == "freed":
    src = None
ptrs = [[t.data_ptr() for t in s] for s in bufs]
cov = [torch.empty((B, 16), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
spec = [torch.empty((B, 1024), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
idx = [i % nbuf for i in range(steps)]
p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, B); p.set_lanes(4)
call = p.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                         [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], doa.DETACHED)
call(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e6)
print(f"{mode}: {best:6.2f} us/step", flush=True)
