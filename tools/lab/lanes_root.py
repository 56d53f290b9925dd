"""lab (round 4): cfg3 (doa_root_pipeline) and the flowgraph shape by lane count; us per step of one detached call, best of 5"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
B, nbuf = 4096, 8
st = torch.cuda.current_stream()
def streams(which):
    N = 4
    if which == "cfg3":
        return [doa.sim.make_batch_streams_torch(N, 1024, B, 0.44, 2, 20.0, seed=400 + b, device="cuda")[0] for b in range(nbuf)]
    out = []
    span = (B - 1) * 1536 + 2048
    for b in range(nbuf):
        s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
        doa.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600 + b).work_dev(span, [t.data_ptr() for t in s], st)
        out.append(s)
    return out
from doa._lib import lib
WHICH = tuple(os.environ.get("LANES_WHICH", "cfg3,flow").split(","))
LANES = tuple(int(x) for x in os.environ.get("LANES_LIST", "3,4,5,6,8").split(","))
for which in WHICH:
    bufs = streams(which)
    ptrs = [[t.data_ptr() for t in s] for s in bufs]
    cov = [torch.empty((B, 16), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
    ang = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    spec = [torch.empty((B, 1024), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    mx = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    for steps in (20, 100):
        idx = [i % nbuf for i in range(steps)]
        for lanes in LANES:
            if which == "cfg3":
                p = doa.root_pipeline(4, 1024, 0, 0, 0.44, 2, B); p.set_lanes(lanes)
                call = p.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx], None, doa.DETACHED)
            else:
                p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, B); p.set_lanes(lanes)
                call = p.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                                         [mx[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx], doa.DETACHED)
            call(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); call(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps * 1e6)
            print(f"{which} steps {steps:3d} lanes {lanes}: {best:6.2f} us/step   (lanes seen side by side: {lib.doa_hip_lane_streams_verified_debug()})", flush=True)
            del p
