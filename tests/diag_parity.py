"""GPU diagnostic (not a test): per scenario, how far the HIP MUSIC path and the LAPACK-fp32 oracle
sit from the fp64 evaluation, for the projector, the null spectrum and the dB spectrum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import doa, doa_oracle as oracle
from scenarios import SCENARIOS, make_input

for name in SCENARIOS:
    c, x = make_input(name)
    N, M, P, n = c["N"], c["M"], c["P"], c["n"]
    R = oracle.autocorrelate(x, c["K"], c["ovl"], c["fb"], n)
    s32, q32, p32 = oracle.music_lin_array(R, c["d"], M, N, P, "f32", True)
    s64, q64, p64 = oracle.music_lin_array(R, c["d"], M, N, P, "f64", True)
    for bits in (64, 32):
        doa.set_internal_precision(bits)
        blk = doa.MUSIC_lin_array(c["d"], M, N, P)
        doa.set_internal_precision(64)
        spec = np.empty((n, P), np.float32); blk.work(n, [R], [spec])
        pn, q = blk.debug(R)
        rows = []
        for i in range(n):
            Ph = pn[i].reshape(N, N, order="F")
            qt = q64[i]; mx = qt.max()
            g2 = qt >= 1e-2 * mx; g1 = qt >= 1e-1 * mx
            rows.append([np.abs(Ph - p64[i]).max(), np.abs(p32[i] - p64[i]).max(),
                         (np.abs(q[i] - qt)[g2] / qt[g2]).max(), (np.abs(q32[i] - qt)[g2] / qt[g2]).max(),
                         (np.abs(q[i] - qt)[g1] / qt[g1]).max(), (np.abs(q32[i] - qt)[g1] / qt[g1]).max(),
                         np.abs(q[i] - qt).max() / mx, np.abs(q32[i] - qt).max() / mx,
                         (np.abs(q[i] - q32[i])[g2] / qt[g2]).max(),
                         np.abs(spec[i] - s64[i])[np.isfinite(s64[i])].max(), np.abs(s32[i] - s64[i])[np.isfinite(s64[i])].max(),
                         float(np.argmax(spec[i]) == np.argmax(s64[i]))])
        r = np.array(rows).max(axis=0)
        print(f"{name:18s} evd{bits} dP hip/ref {r[0]:.1e}/{r[1]:.1e}  relQ>1e-2 {r[2]:.1e}/{r[3]:.1e}  relQ>1e-1 {r[4]:.1e}/{r[5]:.1e}"
              f"  absQ/max {r[6]:.1e}/{r[7]:.1e}  hip-vs-ref32 {r[8]:.1e}  dB {r[9]:.3f}/{r[10]:.3f}  samebin(min) {np.array(rows)[:,11].min():.0f}")
