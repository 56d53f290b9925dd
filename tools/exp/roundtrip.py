"""Launch + synchronize round trip of a tiny kernel (host latency floor of any timed region)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch, doa
blk = doa.find_local_max(1, 256, 0.0, 180.0)
x = torch.randn(1, 256, device="cuda"); a = torch.empty(1, 1, device="cuda"); b = torch.empty(1, 1, device="cuda")
st = torch.cuda.Stream()
def go(): blk.work_dev(1, x.data_ptr(), a.data_ptr(), b.data_ptr(), st)
for _ in range(20): go()
torch.cuda.synchronize()
for idle_us in (0, 50, 500, 5000):
    ts = []
    for _ in range(50):
        torch.cuda.synchronize()
        t_end = time.perf_counter() + idle_us * 1e-6
        while time.perf_counter() < t_end: pass
        t0 = time.perf_counter(); go(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"idle {idle_us:5d} us before: launch+sync round trip median {ts[25]:.1f} us, min {ts[0]:.1f}, p90 {ts[45]:.1f}")
# the same with an event poll instead of the blocking synchronize
ev = torch.cuda.Event()
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); go(); ev.record(st)
    while not ev.query(): pass
    ts.append((time.perf_counter() - t0) * 1e6)
ts.sort(); print(f"event-poll round trip median {ts[25]:.1f} us, min {ts[0]:.1f}")
