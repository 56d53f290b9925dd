"""lab (round 4): how many items of a 4096-item batch leave the quad subspace iteration for the Jacobi, by data shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import numpy as np, torch, doa
st = torch.cuda.current_stream()
B = 4096
def run(name, N, K, ovl, fb, d, M, P, streams):
    pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, B)
    cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
    spec = torch.empty((B, P), dtype=torch.float32, device="cuda")
    mx = torch.empty((B, M), dtype=torch.float32, device="cuda"); am = torch.empty((B, M), dtype=torch.float32, device="cuda")
    doa.evd_fallback_count(reset=True)
    pipe.work_dev(B, [t.data_ptr() for t in streams], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(), am.data_ptr(), st)
    torch.cuda.synchronize()
    fb_n = doa.evd_fallback_count(reset=True)
    # eigenvalue statistics of the items (host): ratio of the noise eigenvalues' half spread to the signal gap
    R = cov.cpu().numpy().reshape(B, N, N).transpose(0, 2, 1)
    R = np.triu(R) + np.conj(np.transpose(np.triu(R, 1), (0, 2, 1)))
    w = np.linalg.eigvalsh(R.astype(np.complex128))[:, ::-1]
    mu = w[:, M:].mean(axis=1)
    rate = np.abs(w[:, M:] - mu[:, None]).max(axis=1) / (w[:, M - 1] - mu)
    print(f"{name}: fall-backs {fb_n} of {B}; convergence rate per step: median {np.median(rate):.2e}, 90% {np.quantile(rate, 0.9):.2e}, "
          f"99% {np.quantile(rate, 0.99):.2e}, max {rate.max():.2e}; items with rate > 0.3: {(rate > 0.3).sum()}")
s, _ = doa.sim.make_batch_streams_torch(4, 1024, B, 0.44, 2, 20.0, seed=400, device="cuda")
run("cfg3 data (N=4, M=2, random directions per snapshot, 20 dB)", 4, 1024, 0, 0, 0.44, 2, 1024, s)
s, _ = doa.sim.make_batch_streams_torch(4, 1024, B, 0.5, 3, 20.0, seed=401, device="cuda")
run("N=4, M=3, random directions per snapshot, 20 dB", 4, 1024, 0, 0, 0.5, 3, 1024, s)
K, ovl = 2048, 512
span = (B - 1) * (K - ovl) + K
s = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(4)])
src = doa.sim_source(4, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, None, 0.1, seed=600)
src.work_dev(span, [t.data_ptr() for t in s], st)
run("flowgraph shape (sim_source, 30/123 deg, antenna noise 0.1, K=2048, overlap 512, FB)", 4, K, ovl, 1, 0.4, 2, 1024, s)
src = doa.sim_source(4, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, [5e-5, 5e-3], 0.0, seed=601)
src.work_dev(span, [t.data_ptr() for t in s], st)
run("flowgraph scenario itself (per-source noise 5e-5 / 5e-3, no antenna noise: rank-2 items)", 4, K, ovl, 1, 0.4, 2, 1024, s)
