"""lab: checksum of the pipeline's outputs (spectrum, peak value, peak location) on one fixed batch, for bit-comparisons of
scan-kernel variants across processes (DOA_SCAN_VARIANT is read once per process)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import numpy as np
import torch
import doa
N, K, B, P, M = 4, 64, int(os.environ.get("SCAN_CHECK_BATCH", "20000")), 1024, 1
st = torch.cuda.current_stream()
pipe = doa.music_pipeline(N, K, 0, 0, 0.5, M, P, B)
s, _ = doa.sim.make_batch_streams_torch(N, K, B, 0.5, M, 20.0, seed=5, device="cuda")
# a few degenerate rows: constant spectrum (R = c I) and a non-finite item
for t in s:
    t.view(torch.float32)[: 2 * K * 2] = 0
for k in range(N):
    s[k][k:2 * K:N] = 2.0
s[0][5 * K + 3] = float("nan")
cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
spec = torch.full((B, P), -1.0, dtype=torch.float32, device="cuda")
mx = torch.empty((B, M), dtype=torch.float32, device="cuda")
am = torch.empty((B, M), dtype=torch.float32, device="cuda")
pipe.work_dev(B, [t.data_ptr() for t in s], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(), am.data_ptr(), st)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (spec, mx, am):
    h.update(t.cpu().numpy().tobytes())
sp = spec.cpu().numpy()
print("variant", os.environ.get("DOA_SCAN_VARIANT", "-"), "sha", h.hexdigest()[:16], "rows with max==0:", int((np.nanmax(sp, axis=1) == 0).sum()),
      "zeros:", int((sp == 0).sum()), "nan rows:", int(np.isnan(sp).any(axis=1).sum()))
