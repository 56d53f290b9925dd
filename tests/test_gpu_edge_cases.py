"""GPU edge cases beyond the reference's own tests: grid-stride paths (more items than resident
waves), degenerate / extreme covariance matrices, non-finite input (no hang, no crash), scaling."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def test_autocorrelate_many_small_windows_grid_stride():
    # 20000 windows >> 2048 resident waves: every wave loops over ~10 snapshots
    N, K, ovl, n = 4, 64, 16, 20000
    S = K - ovl
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((N, (n - 1) * S + K)) + 1j * rng.standard_normal((N, (n - 1) * S + K))).astype(np.complex64)
    blk = doa.autocorrelate(N, K, ovl, 1)
    R = np.empty((n, N * N), np.complex64)
    blk.general_work(n, [x[k] for k in range(N)], [R])
    idx = np.r_[0:8, 2040:2056, 9990:10010, n - 8:n]
    ref = oracle.autocorrelate(x[:, idx[0] * S:], K, ovl, 1, 8, precision="f64")
    assert np.abs(R[:8] - ref).max() <= 2e-6 * np.abs(ref).max()
    for i in idx:
        w = x[:, i * S:i * S + K].astype(np.complex128)
        Ri = w @ w.conj().T / K
        J = np.fliplr(np.eye(N))
        Ri = 0.5 * Ri + (0.5 / K) * J @ np.conj(Ri) @ J
        assert np.abs(R[i].reshape(N, N, order="F") - Ri).max() <= 2e-6 * np.abs(Ri).max(), i


def test_music_many_items_grid_stride_and_sixteen_antennas():
    # 5000 items with N=16 (MFMA covariance upstream not needed here): group Jacobi + generic scan
    N, M, P, n = 16, 2, 256, 600
    c = dict(d=0.5)
    x = doa.sim.make_streams(N, n * 128, [50.0, 120.0], 0.5, snr_db=10.0, seed=4)
    R = oracle.autocorrelate(x, 128, 0, 0, n)
    blk = doa.MUSIC_lin_array(0.5, M, N, P)
    spec = np.empty((n, P), np.float32)
    blk.work(n, [R], [spec])
    pick = [0, 1, 63, 64, 300, n - 1]
    s64 = oracle.music_lin_array(R[pick], 0.5, M, N, P, "f64")
    assert np.abs(spec[pick] - s64).max() <= 1e-3
    assert np.all(spec.max(axis=1) == 0.0)


def test_music_diagonal_and_scaled_covariances():
    # diagonal R with distinct entries: eigenvectors are unit vectors, the noise subspace is the
    # N-M smallest diagonal entries -> P_N is a 0/1 diagonal, Q = number of noise elements (constant)
    N, M, P = 4, 1, 64
    R = np.zeros((3, N, N), np.complex64)
    R[0] = np.diag([4.0, 1.0, 3.0, 2.0])
    R[1] = np.diag([1e-12, 3e-12, 2e-12, 9e-12])          # tiny scale
    R[2] = np.diag([5e12, 1e12, 3e12, 2e12])              # huge scale
    items = R.transpose(0, 2, 1).reshape(3, N * N).copy()
    blk = doa.MUSIC_lin_array(0.5, M, N, P)
    pn, q = blk.debug(items)
    for i, drop in enumerate([0, 3, 0]):                   # index of the largest entry = signal
        want = np.eye(N)
        want[drop, drop] = 0
        assert np.abs(pn[i].reshape(N, N, order="F") - want).max() <= 1e-7
        assert np.abs(q[i] - 3.0).max() <= 1e-5
    spec = np.empty((3, P), np.float32)
    blk.work(3, [items], [spec])
    assert np.abs(spec).max() <= 1e-4                      # flat spectrum: everything at the 0 dB maximum


def test_non_finite_input_terminates():
    # NaN / Inf covariances: the reference's LAPACK path returns garbage or raises; here the bounded
    # Jacobi / Aberth loops must simply terminate and the call return
    N, M, P = 4, 1, 128
    R = np.full((4, N * N), np.nan + 0j, np.complex64)
    R[1] = np.inf
    R[2] = 0
    R[3] = np.eye(N).reshape(-1)
    blk = doa.MUSIC_lin_array(0.5, M, N, P)
    spec = np.empty((4, P), np.float32)
    assert blk.work(4, [R], [spec]) == 4
    f = doa.find_local_max(2, P, 0.0, 180.0)
    v0 = np.empty((4, 2), np.float32)
    v1 = np.empty((4, 2), np.float32)
    assert f.work(4, [spec], [v0, v1]) == 4
    root = doa.rootMUSIC_linear_array(0.5, M, N)
    ang = np.empty((4, M), np.float32)
    with pytest.raises(doa.DoaError) as ei:                # rows 0 and 1: no root compares as "inside" (see
        root.work(4, [R], [ang])                           # test_gpu_root_music.py for the parity side of this)
    assert ei.value.status == -5 and np.isnan(ang[:2]).all()


def test_find_local_max_ties_and_flats_at_chunk_borders():
    # plateaus and equal peaks straddling lane (4-element) and chunk (256-element) boundaries
    L = 1024
    rows = []
    for start in (2, 3, 4, 254, 255, 256, 257, 510, 511, 512, 1019, 1020):
        for width in (2, 3, 5, 9):
            v = np.zeros(L, np.float32)
            v[start:start + width] = 1.0
            v[(start + 300) % (L - 12) + 1:(start + 300) % (L - 12) + 3] = 1.0       # an equal-height rival
            rows.append(v)
    v = np.stack(rows)
    for M in (2, 3):
        blk = doa.find_local_max(M, L, 0.0, 180.0)
        o0 = np.empty((v.shape[0], M), np.float32)
        o1 = np.empty((v.shape[0], M), np.float32)
        blk.work(v.shape[0], [v], [o0, o1])
        r0, r1 = oracle.find_local_max(v, M, L, 0.0, 180.0)
        assert np.array_equal(o0, r0) and np.array_equal(o1, r1)


def test_internal_precision_is_a_property_of_the_handle():
    """the process-wide doa_set_internal_precision is only the default a handle copies at create; a handle's
    own setter changes that handle and nothing else (two handles of different precisions side by side)."""
    import doa_oracle as oracle
    from scenarios import make_input
    c, x = make_input("bench_cfg2")
    R = oracle.autocorrelate(x, c["K"], c["ovl"], c["fb"], c["n"])
    a = doa.MUSIC_lin_array(c["d"], c["M"], c["N"], c["P"])
    b = doa.MUSIC_lin_array(c["d"], c["M"], c["N"], c["P"])
    b.set_internal_precision(32)
    assert doa.get_internal_precision() == 64                        # the default is untouched
    sa, sb, sa2 = (np.empty((c["n"], c["P"]), np.float32) for _ in range(3))
    a.work(c["n"], [R], [sa]); b.work(c["n"], [R], [sb])
    assert not np.array_equal(sa, sb) and np.abs(sa - sb).max() < 0.5  # two arithmetic paths, same spectrum shape
    b.set_internal_precision(64)
    b.work(c["n"], [R], [sa2])
    assert np.array_equal(sa, sa2)
    with pytest.raises(doa.DoaError):
        a.set_internal_precision(16)
    pipe = doa.music_pipeline(c["N"], c["K"], c["ovl"], c["fb"], c["d"], c["M"], c["P"], max_batch=c["n"])
    pipe.set_internal_precision(32)
    pipe.set_internal_precision(64)                                  # back to the parity configuration (records re-reserved)
    mx, am = np.empty((c["n"], c["M"]), np.float32), np.empty((c["n"], c["M"]), np.float32)
    sp = np.empty((c["n"], c["P"]), np.float32)
    assert pipe.work(c["n"], [x[k] for k in range(c["N"])], mx, am, spectrum_out=sp) == c["n"]
    fresh = doa.music_pipeline(c["N"], c["K"], c["ovl"], c["fb"], c["d"], c["M"], c["P"], max_batch=c["n"])
    mx2, am2, sp2 = np.empty_like(mx), np.empty_like(am), np.empty_like(sp)
    assert fresh.work(c["n"], [x[k] for k in range(c["N"])], mx2, am2, spectrum_out=sp2) == c["n"]
    assert np.array_equal(sp, sp2) and np.array_equal(am, am2)
