"""lab (round 4): how many items of a 4096-item batch the one-noise-vector iteration hands to the Jacobi, by data"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import numpy as np, torch, doa
st = torch.cuda.current_stream()
B = 4096
def run(name, N, M, d, snr, K=1024):
    s, _ = doa.sim.make_batch_streams_torch(N, K, B, d, M, snr, seed=400 + N + M, device="cuda")
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, 1024, B)
    cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
    spec = torch.empty((B, 1024), dtype=torch.float32, device="cuda")
    mx = torch.empty((B, M), dtype=torch.float32, device="cuda"); am = torch.empty((B, M), dtype=torch.float32, device="cuda")
    doa.evd_fallback_count(reset=True)
    pipe.work_dev(B, [t.data_ptr() for t in s], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(), am.data_ptr(), st)
    torch.cuda.synchronize()
    n_fb = doa.evd_fallback_count(reset=True)
    R = cov.cpu().numpy().reshape(B, N, N).transpose(0, 2, 1)
    R = np.triu(R) + np.conj(np.transpose(np.triu(R, 1), (0, 2, 1)))
    w = np.linalg.eigvalsh(R.astype(np.complex128))
    gap = (w[:, 1] - w[:, 0]) / np.sqrt((w ** 2).sum(axis=1))
    print(f"{name}: fall-backs {n_fb} of {B}; (lambda_2 - lambda_1)/||A||: median {np.median(gap):.2e}, 1% {np.quantile(gap, 0.01):.2e}, "
          f"min {gap.min():.2e}; items below the certificate's 1e-5: {(gap < 1e-5).sum()}")
for N, M in ((4, 3), (3, 2), (2, 1)):
    for snr in (20.0, 5.0):
        run(f"N={N} M={M}, random directions per snapshot, {snr:.0f} dB", N, M, 0.5, snr)
