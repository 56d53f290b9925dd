"""GPU tests: (1) the C ABI is thread-safe across handles (GNU Radio runs one thread per block;
ctypes releases the GIL during the calls, so the Python threads below really overlap), and (2) a
seeded random sweep over block parameters against the oracle, beyond the named scenarios."""
import os
import threading

import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _chain(seed, reps, out, err):
    try:
        rng = np.random.default_rng(seed)
        N, K, ovl, fb, M, P, n = 4, 256, 32 * (seed % 3), seed % 2, 1 + seed % 2, 512, 24
        S = K - ovl
        x = doa.sim.make_streams(N, (n - 1) * S + K, [40.0 + 10 * seed, 130.0][:M], 0.5, snr_db=15.0, seed=seed)
        a = doa.autocorrelate(N, K, ovl, fb)
        m = doa.MUSIC_lin_array(0.5, M, N, P)
        f = doa.find_local_max(M, P, 0.0, 180.0)
        r = doa.rootMUSIC_linear_array(0.5, M, N)
        res = []
        for _ in range(reps):
            R = np.empty((n, N * N), np.complex64)
            a.general_work(n, [x[k] for k in range(N)], [R])
            S_ = np.empty((n, P), np.float32)
            m.work(n, [R], [S_])
            v0, v1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
            f.work(n, [S_], [v0, v1])
            ang = np.empty((n, M), np.float32)
            r.work(n, [R], [ang])
            res.append((R, S_, v0, v1, ang))
            assert doa.last_error() == ""                     # the error slot is per thread
        out[seed] = res
        del rng
    except Exception as e:                                    # pragma: no cover - reported by the main thread
        err[seed] = e


def test_handles_are_independent_across_threads():
    seeds, reps = list(range(6)), 8
    serial, par, err = {}, {}, {}
    for s in seeds:
        _chain(s, 1, serial, err)
    assert not err, err
    threads = [threading.Thread(target=_chain, args=(s, reps, par, err)) for s in seeds]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not err, err
    for s in seeds:
        for got in par[s]:
            for g, w in zip(got, serial[s][0]):
                assert np.array_equal(g, w, equal_nan=True), s


def test_error_message_is_thread_local():
    seen = {}

    def bad():
        try:
            doa.MUSIC_lin_array(0.5, 9, 4, 64)                # num_targets >= num_ant_ele
        except doa.DoaError as e:
            seen["bad"] = str(e)

    t = threading.Thread(target=bad)
    t.start()
    t.join()
    assert "num_targets" in seen["bad"]
    blk = doa.find_local_max(1, 64, 0.0, 180.0)               # a successful call on this thread
    assert doa.last_error() == "" and blk is not None


@pytest.mark.parametrize("seed", range(int(os.environ.get("DOA_TEST_SEEDS", "96"))))
def test_random_configuration_against_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(2, 17))
    M = int(rng.integers(1, min(N, 5)))
    K = int(rng.integers(2 * N, 700))
    ovl = int(rng.integers(0, K // 2)) if seed % 3 else int(rng.integers(K // 2, K - 1))     # every third: deep overlap (q >= 2)
    fb = int(rng.integers(0, 2))
    P = int(rng.choice([64, 180, 256, 500, 1024, 1500, 2048]))
    d = float(rng.choice([0.3, 0.4, 0.5]))
    n = int(rng.integers(3, 20))
    thetas = np.sort(rng.uniform(25.0, 155.0, size=M))
    thetas += np.arange(M) * 12.0
    thetas = np.clip(thetas, 10.0, 170.0)
    S = K - ovl
    x = doa.sim.make_streams(N, (n - 1) * S + K, list(thetas), d, snr_db=float(rng.choice([5.0, 15.0, 30.0])), seed=seed)
    # K1
    a = doa.autocorrelate(N, K, ovl, fb)
    R = np.empty((n, N * N), np.complex64)
    a.general_work(n, [x[k] for k in range(N)], [R])
    R64 = oracle.autocorrelate(x, K, ovl, fb, n, precision="f64")
    assert np.abs(R - R64).max() <= 3e-6 * np.abs(R64).max()
    # K2-K4 against the fp64 evaluation of the reference's formulas, on the same covariance items
    m = doa.MUSIC_lin_array(d, M, N, P)
    spec = np.empty((n, P), np.float32)
    m.work(n, [R], [spec])
    s64 = oracle.music_lin_array(R, d, M, N, P, "f64")
    assert np.all(spec.max(axis=1) == 0.0)
    assert np.abs(spec - s64).max() <= 1e-4 + 2e-6 * np.abs(s64).max(), (N, M, K, ovl, fb, P)
    # K5 bit for bit on the produced spectrum
    f = doa.find_local_max(M, P, 0.0, 180.0)
    v0, v1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    f.work(n, [spec], [v0, v1])
    o0, o1 = oracle.find_local_max(spec, M, P, 0.0, 180.0)
    assert np.array_equal(v0, o0) and np.array_equal(v1, o1)
    # the fused pipeline gives the same peaks
    pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, max_batch=n)
    p0, p1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    pspec = np.empty((n, P), np.float32)
    pipe.work(n, [x[k] for k in range(N)], p0, p1, spectrum_out=pspec)
    assert np.all(np.abs(pspec - spec) <= 2e-6 + 5e-7 * np.abs(spec))
    q0, q1 = oracle.find_local_max(pspec, M, P, 0.0, 180.0)
    assert np.array_equal(p0, q0) and np.array_equal(p1, q1)
    # K6
    r = doa.rootMUSIC_linear_array(d, M, N)
    ang = np.empty((n, M), np.float32)
    r.work(n, [R], [ang])
    a64 = oracle.root_music(R, d, M, N, "f64")
    # d < 0.5: a root with |arg z| > 2 pi d has no real angle; acos(> 1) is NaN in the reference too
    assert np.array_equal(np.isnan(ang), np.isnan(a64)), (N, M, K, d)
    ok = ~np.isnan(a64)
    assert np.abs(ang[ok] - a64[ok]).max() <= 2e-3, (N, M, K, d)
