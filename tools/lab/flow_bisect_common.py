import torch, doa
B, N = 4096, 4
st = torch.cuda.current_stream()
span = (B - 1) * 1536 + 2048
def make(nbuf):
    bufs = []
    for b in range(nbuf):
        s = [torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)]
        s = doa.sim.stream_slab_torch(s)
        doa.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600 + b).work_dev(span, [t.data_ptr() for t in s], st)
        bufs.append(s)
    ptrs = [[t.data_ptr() for t in s] for s in bufs]
    cov = [torch.empty((B, 16), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
    spec = [torch.empty((B, 1024), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    mx = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    am = [torch.empty((B, 2), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    return dict(bufs=bufs, ptrs=ptrs, cov=cov, spec=spec, mx=mx, am=am, nbuf=nbuf)
