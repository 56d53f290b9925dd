"""lab: us per 4096-snapshot step of one pipeline config: serial, the library's lanes (own streams / adopted torch streams,
attached to the caller's stream / detached) and caller-side streams with one handle each.
usage: python tools/lab/lanes_sweep.py N K ovl fb d M P [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch
import doa
N, K, ovl, fb, d, M, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 40
B, nbuf = 4096, 4
st = torch.cuda.current_stream()
S = K - ovl
span = (B - 1) * S + K
bufs = []
for b in range(nbuf):
    if ovl == 0:
        s, _ = doa.sim.make_batch_streams_torch(N, K, B, d, M, 20.0, seed=500 + b, device="cuda")
    else:
        s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
        src = doa.sim_source(N, d, [30.0, 123.0, 75.0][:M], [0.03125, 0.0625, 0.11][:M], None, None, 0.1, seed=600 + b)
        src.work_dev(span, [t.data_ptr() for t in s], st)
    bufs.append(s)
ptrs = [[t.data_ptr() for t in s] for s in bufs]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]

def timed(fn, n, sync=torch.cuda.synchronize):
    fn(4); sync(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(n); sync()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    torch.cuda.synchronize()
    return best

def caller_streams(L):
    pipes = [doa.music_pipeline(N, K, ovl, fb, d, M, P, B) for _ in range(L)]
    sts = [torch.cuda.Stream() for _ in range(L)]
    cs = lambda n: [pipes[i % L].work_dev(B, ptrs[i % nbuf], cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(), am[i % nbuf].data_ptr(), sts[i % L]) for i in range(n)]
    return timed(cs, reps)
if os.environ.get("WARM"):
    x = torch.randn(8192, 8192, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(os.environ["WARM"]):
        y = x @ x
        torch.cuda.synchronize()
keep = []
for _ in range(int(os.environ.get("PRE_STREAMS", "0"))):
    s_ = torch.cuda.Stream(); keep.append(s_)
    with torch.cuda.stream(s_):
        torch.zeros(16, device="cuda").add_(1)
torch.cuda.synchronize()
print(f"  caller streams L=4 (FIRST):               {caller_streams(4):8.2f} us/step")
print(f"  caller streams L=4 (SECOND):              {caller_streams(4):8.2f} us/step")
print(f"  caller streams L=4 (THIRD):               {caller_streams(4):8.2f} us/step")
pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, B)
serial = lambda n: [pipe.work_dev(B, ptrs[i % nbuf], cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(), am[i % nbuf].data_ptr(), st) for i in range(n)]
print(f"N={N} K={K} ovl={ovl} fb={fb} M={M} P={P} reps={reps}")
print(f"  serial (one stream):                      {timed(serial, reps):8.2f} us/step")
def batches(p, stream, per):
    def fn(n):
        for i0 in range(0, n, per):
            idx = [i % nbuf for i in range(i0, min(n, i0 + per))]
            p.work_dev_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                               [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], stream)
    return fn
for L in (2, 4):
    for own in (True, False):
        p = doa.music_pipeline(N, K, ovl, fb, d, M, P, B)
        if own:
            p.set_lanes(L)
        else:
            p.set_lane_streams([torch.cuda.Stream() for _ in range(L)])
        kind = "own streams   " if own else "torch streams "
        for per in (reps, 20, 1):
            print(f"  lanes L={L} {kind} attached, calls of {per:3d}: {timed(batches(p, st, per), reps):8.2f} us/step")
            print(f"  lanes L={L} {kind} detached, calls of {per:3d}: {timed(batches(p, doa.DETACHED, per), reps, p.synchronize):8.2f} us/step")
for L in (2, 4):
    pipes = [doa.music_pipeline(N, K, ovl, fb, d, M, P, B) for _ in range(L)]
    sts = [torch.cuda.Stream() for _ in range(L)]
    cs = lambda n: [pipes[i % L].work_dev(B, ptrs[i % nbuf], cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(), am[i % nbuf].data_ptr(), sts[i % L]) for i in range(n)]
    print(f"  caller streams L={L}, one handle each:      {timed(cs, reps):8.2f} us/step")
