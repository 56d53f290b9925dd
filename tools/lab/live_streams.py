"""lab (round 4): the flowgraph-shape call (4 lanes) with k other streams alive in the process when the handle creates its lanes
(each doa.sim_source handle owns one stream).  argv: k [n_lanes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
from flow_bisect_common import make, B, st
k = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = make(8)
extra = [doa.sim_source(4, 0.4, [30.0], [0.03125], None, None, 0.1, seed=1) for _ in range(k)]
res = []
for steps in (20, 100):
    idx = [i % 8 for i in range(steps)]
    p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, B); p.set_lanes(L)
    call = p.prepare_batches(B, [d["ptrs"][b] for b in idx], [d["cov"][b].data_ptr() for b in idx], [d["spec"][b].data_ptr() for b in idx],
                             [d["mx"][b].data_ptr() for b in idx], [d["am"][b].data_ptr() for b in idx], doa.DETACHED)
    call(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); call(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    res.append(best)
    del p, call
print(f"{k} live streams, {L} lanes: {res[0]:6.2f} us/step at 20 steps, {res[1]:6.2f} at 100", flush=True)
