"""Golden-vector tests (tests/golden/*.npz, minted by tests/golden/make_golden.py).

CPU: the oracle still reproduces the committed fixtures (guards the checker against drift).
GPU: the HIP path, through the C ABI, against the committed expectations — no oracle involved."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    N, M, P, K, ovl, fb, n = (int(v) for v in z["cfg"])
    return z, dict(N=N, M=M, P=P, K=K, ovl=ovl, fb=fb, n=n, d=float(z["d"]))


def test_fixtures_exist():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    import doa_oracle as oracle
    z, c = _load(path)
    R32 = oracle.autocorrelate(z["x"], c["K"], c["ovl"], c["fb"], c["n"])
    assert np.abs(R32 - z["R32"]).max() <= 1e-6 * np.abs(z["R32"]).max()       # BLAS build may reorder sums
    spec64, Q64, PN64 = oracle.music_lin_array(z["R32"], c["d"], c["M"], c["N"], c["P"], "f64", return_parts=True)
    assert np.abs(PN64 - z["PN64"]).max() <= 1e-12
    assert np.all(np.abs(Q64 - z["Q64"]) <= 1e-9 * np.abs(z["Q64"]) + 1e-12)
    val64, loc64 = oracle.find_local_max(z["spec64"].astype(np.float32), c["M"], c["P"], 0.0, 180.0)
    assert np.array_equal(val64, z["val64"]) and np.array_equal(loc64, z["loc64"])
    root64 = oracle.root_music(z["R32"], c["d"], c["M"], c["N"], "f64")
    assert np.abs(root64 - z["root64"]).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(path):
    import doa
    z, c = _load(path)
    N, M, P, n = c["N"], c["M"], c["P"], c["n"]
    x = z["x"]
    # K1
    a = doa.autocorrelate(N, c["K"], c["ovl"], c["fb"])
    R = np.empty((n, N * N), np.complex64)
    a.general_work(n, [x[k] for k in range(N)], [R])
    assert np.abs(R - z["R64"]).max() <= 2e-6 * np.abs(z["R64"]).max()
    assert np.abs(R - z["R32"]).max() <= 1e-5 * np.abs(z["R32"]).max()
    # K2-K4 on the fixture's covariances (identical inputs)
    m = doa.MUSIC_lin_array(c["d"], M, N, P)
    spec = np.empty((n, P), np.float32)
    m.work(n, [z["R32"]], [spec])
    pn, q = m.debug(z["R32"])
    assert np.abs(pn.reshape(n, N, N).transpose(0, 2, 1) - z["PN64"]).max() <= 1e-7
    assert np.all(np.abs(q - z["Q64"]) <= 3e-7 * np.abs(z["Q64"]) + 2e-13 * np.abs(z["Q64"]).max())
    assert np.all(spec.max(axis=1) == 0.0)
    good = z["Q64"] >= 1e-2 * z["Q64"].max(axis=1, keepdims=True)
    diff = spec.astype(np.float64) - z["spec64"]
    for i in range(n):
        dg = diff[i][good[i] & np.isfinite(z["spec64"][i])]
        assert dg.max() - dg.min() <= 4e-5
    # K5 on the fixture's spectrum: bit exact
    f = doa.find_local_max(M, P, 0.0, 180.0)
    v0 = np.empty((n, M), np.float32)
    v1 = np.empty((n, M), np.float32)
    f.work(n, [z["spec32"]], [v0, v1])
    assert np.array_equal(v0, z["val32"]) and np.array_equal(v1, z["loc32"])
    # angles of the chained HIP path: same grid bins as the fp64 expectation (within one bin on the
    # rank-deficient fixtures), and within 180/P + 1e-3 deg of the fp32 expectation
    f.work(n, [spec], [v0, v1])
    assert np.abs(v1 - z["loc64"]).max() <= 180.0 / P + 1e-3
    # K6
    r = doa.rootMUSIC_linear_array(c["d"], M, N)
    ang = np.empty((n, M), np.float32)
    r.work(n, [z["R32"]], [ang])
    dev32 = np.abs(z["root32"].astype(np.float64) - z["root64"]).max()
    assert np.abs(ang - z["root64"]).max() <= max(1e-3, 2 * dev32)
