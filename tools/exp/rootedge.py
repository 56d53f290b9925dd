import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle")]
import doa, doa_oracle as oracle
def steer(N, d, th):
    loc = d * ((N - 1) / 2.0 - np.arange(N)); return np.exp(-2j * np.pi * np.cos(np.deg2rad(th)) * loc)
for (N, M, d, ths) in [(3, 2, 0.5, (60., 110.)), (4, 2, 0.5, (50., 120.)), (4, 3, 0.5, (40., 90., 130.)), (4, 2, 0.44, (30., 123.)), (5, 3, 0.5, (40, 80, 120)), (3,2,0.3,(20.,160.))]:
    items = []
    for k in range(40):
        th = [t + 0.37 * k for t in ths]
        A = np.stack([steer(N, d, t) for t in th], axis=1)
        p = np.diag(1.0 + 0.1 * np.arange(M) + 0.01 * k)
        R = A @ p @ A.conj().T                                # exactly rank M: double roots on the unit circle
        items.append(R.astype(np.complex64).reshape(-1, order="F"))
    R = np.stack(items)
    blk = doa.rootMUSIC_linear_array(d, M, N)
    ang, roots, status = blk.debug(R)
    inside = (1.0 - np.abs(roots) > 0).sum(axis=1)
    sel = []
    bad = 0
    for i in range(R.shape[0]):
        try:
            s = oracle.root_music_select(roots[i], d, M, "f64")
        except ValueError:
            s = np.full(M, np.nan, np.float32)
        ok = np.allclose(s, ang[i], atol=1e-5, equal_nan=True)
        bad += 0 if ok else 1
    print(N, M, d, "inside counts:", np.bincount(inside, minlength=2 * N - 1), "status!=0:", int((status != 0).sum()), "selection mismatches:", bad, "90s:", int((ang == 90.0).sum()), "nan:", int(np.isnan(ang).sum()))
