"""BASELINE.json configs[2] and configs[3] at their full sizes (batch 4096), device-resident, through
size-independent properties (every snapshot's generated directions are recovered; maxima at exactly
0 dB; outputs sorted as the blocks sort them) plus a sample of rows against the oracle."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_cfg3_root_music_full_batch():
    # 4-element ULA, 2 sources, Root-MUSIC: autocorrelate -> rootMUSIC_linear_array on 4096 snapshots
    N, K, M, B, d = 4, 1024, 2, 4096, 0.44
    streams, thetas = doa.sim.make_batch_streams_torch(N, K, B, d, M, 20.0, seed=31)
    st = torch.cuda.current_stream()
    cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
    ang = torch.empty((B, M), dtype=torch.float32, device="cuda")
    a = doa.autocorrelate(N, K, 0, 0)
    r = doa.rootMUSIC_linear_array(d, M, N)
    assert a.work_dev(B, [s.data_ptr() for s in streams], cov.data_ptr(), st) == B
    assert r.work_dev(B, cov.data_ptr(), ang.data_ptr(), st) == B
    torch.cuda.synchronize()
    got = ang.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.all(np.diff(got, axis=1) >= 0)                               # ascending (:144)
    # two sources at least 4 degrees apart, 20 dB, 1024 samples, 4 elements: Root-MUSIC resolves them
    err = np.abs(got - np.sort(thetas, axis=1))
    assert np.percentile(err, 99) <= 1.0 and err.max() <= 6.0
    pick = np.r_[0:24, B - 8:B]
    a64 = oracle.root_music(cov.cpu().numpy()[pick], d, M, N, "f64")
    assert np.abs(got[pick] - a64).max() <= 1e-3


def test_cfg4_sixteen_elements_full_batch():
    # 16-element ULA, 3 sources, 4096-point spectrum, MFMA covariance: the fused pipeline on 4096 snapshots
    N, K, M, P, B, d = 16, 1024, 3, 4096, 4096, 0.5
    streams, thetas = doa.sim.make_batch_streams_torch(N, K, B, d, M, 10.0, seed=32)
    st = torch.cuda.current_stream()
    cov = torch.empty((B, N * N), dtype=torch.complex64, device="cuda")
    spec = torch.empty((B, P), dtype=torch.float32, device="cuda")
    mx = torch.empty((B, M), dtype=torch.float32, device="cuda")
    am = torch.empty((B, M), dtype=torch.float32, device="cuda")
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, max_batch=B)
    assert pipe.work_dev(B, [s.data_ptr() for s in streams], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(),
                         am.data_ptr(), st) == B
    torch.cuda.synchronize()
    S = spec.cpu().numpy()
    assert np.all(S.max(axis=1) == 0.0)
    v, loc = mx.cpu().numpy(), am.cpu().numpy()
    assert np.all(v[:, 0] == 0.0) and np.all(np.diff(v, axis=1) <= 0)      # values descending, best peak = the maximum
    assert np.all(np.diff(loc, axis=1) <= 0)                               # locations sorted descending on their own
    err = np.abs(np.sort(loc, axis=1) - np.sort(thetas, axis=1))
    assert np.percentile(err, 99) <= 0.5                                   # 16 elements at 10 dB: well inside a degree
    # a handful of snapshots draw two sources 4 degrees apart near end-fire, which MUSIC does not resolve at
    # 10 dB: fewer peaks than sources -> the reference's fill rule.  That is the estimator, not the port: those
    # rows must still be exactly what the oracle's find_local_max makes of the same spectrum.
    bad = np.nonzero(err.max(axis=1) > 1.0)[0]
    assert bad.size <= B // 200
    if bad.size:
        b0, b1 = oracle.find_local_max(S[bad[:32]], M, P, 0.0, 180.0)
        assert np.array_equal(v[bad[:32]], b0) and np.array_equal(loc[bad[:32]], b1)
    # covariance (MFMA kernel) and spectrum of a sample of rows against the oracle
    pick = np.r_[0:6, 2047:2050, B - 3:B]
    x = np.stack([s.cpu().numpy() for s in streams])
    for i in pick:
        w = x[:, i * K:(i + 1) * K].astype(np.complex128)
        Ri = w @ w.conj().T / K
        assert np.abs(cov[i].cpu().numpy().reshape(N, N, order="F") - Ri).max() <= 3e-6 * np.abs(Ri).max()
    R = cov.cpu().numpy()[pick]
    s64 = oracle.music_lin_array(R, d, M, N, P, "f64")
    assert np.all(np.abs(S[pick] - s64) <= 1e-4 + 2e-6 * np.abs(s64))
    o0, o1 = oracle.find_local_max(S[pick], M, P, 0.0, 180.0)
    assert np.array_equal(v[pick], o0) and np.array_equal(loc[pick], o1)
