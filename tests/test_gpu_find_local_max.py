"""GPU parity of K5 (doa.find_local_max): compare/index logic only, so the bar is BIT-EXACT
equality with the oracle on identical inputs (both output ports), including the reference's
edge cases: flats and plateau peaks, fewer peaks than requested (with the reference's list-position
fill quirk, lib/find_local_max_impl.cc:150-160), no peak at all, M == 1 (global arg-max)."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _run(v, M, L, x_min=0.0, x_max=180.0):
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, L)
    n = v.shape[0]
    blk = doa.find_local_max(M, L, x_min, x_max)
    o0 = np.empty((n, M), np.float32)
    o1 = np.empty((n, M), np.float32)
    assert blk.work(n, [v], [o0, o1]) == n
    return o0, o1


def _check(v, M, L, x_min=0.0, x_max=180.0):
    got0, got1 = _run(v, M, L, x_min, x_max)
    ref0, ref1 = oracle.find_local_max(v, M, L, x_min, x_max)
    assert np.array_equal(got0.view(np.uint32), ref0.view(np.uint32)), (got0[:3], ref0[:3])
    assert np.array_equal(got1.view(np.uint32), ref1.view(np.uint32)), (got1[:3], ref1[:3])


@pytest.mark.parametrize("L", [256, 512, 1000, 1024, 2048, 4096, 37, 1022, 8192])
@pytest.mark.parametrize("M", [1, 2, 3, 5])
def test_random_vectors(L, M):
    rng = np.random.default_rng(L * 10 + M)
    _check(rng.standard_normal((40, L)).astype(np.float32), M, L)


@pytest.mark.parametrize("L", [512, 1024, 4096, 37])
@pytest.mark.parametrize("M", [2, 4])
def test_quantised_vectors_full_of_flats(L, M):
    # few distinct levels -> long flat runs, plateau peaks, equal-valued peaks (ties)
    rng = np.random.default_rng(L + M)
    v = np.round(rng.standard_normal((60, L)) * 1.5).astype(np.float32)
    _check(v, M, L)
    v2 = rng.integers(0, 2, size=(60, L)).astype(np.float32)
    _check(v2, M, L)


@pytest.mark.parametrize("M", [1, 2, 3])
def test_degenerate_vectors(M):
    L = 1024
    t = np.arange(L, dtype=np.float32)
    rows = [
        np.zeros(L), np.ones(L) * -3.5,                    # constant: no peak -> global arg-max (index 0)
        t, -t,                                             # monotonic: end points are never peaks
        np.where(t < 512, t, 1023 - t),                    # one interior peak
        np.where((t > 100) & (t < 200), 5.0, 0.0),         # one plateau
        np.concatenate([np.zeros(L - 1), [1.0]]),          # maximum at the last sample
        np.concatenate([[1.0], np.zeros(L - 1)]),          # maximum at the first sample
        np.where(t % 2 == 0, 1.0, 0.0),                    # 511 equal peaks (ties)
        np.where(t >= 1000, 7.0, np.sin(t / 9.0)),         # flat run reaching the end of the vector
        np.full(L, -np.inf),                               # nothing above -inf
    ]
    _check(np.stack(rows).astype(np.float32), M, L)


def test_fewer_peaks_than_requested_uses_reference_fill_rule():
    L, M = 512, 4
    t = np.arange(L, dtype=np.float32)
    v = np.exp(-((t - 300) / 20.0) ** 2) + 0.5 * np.exp(-((t - 100) / 10.0) ** 2)   # exactly two peaks
    got0, got1 = _run(v, M, L)
    ref0, ref1 = oracle.find_local_max(v, M, L, 0.0, 180.0)
    assert np.array_equal(got0, ref0) and np.array_equal(got1, ref1)
    # best peak is the second entry of the ascending peak list -> the fill index is 1 (the quirk)
    x = oracle.find_local_max_x_axis(L, 0.0, 180.0)
    assert got0[0, 2] == v[1] and got0[0, 3] == v[1]
    assert x[1] in got1[0]


@pytest.mark.parametrize("which,L,M", [(1, 2 ** 11, 3), (2, 2 ** 12, 5)])
def test_reference_qa_signals(which, L, M):
    """python/qa_find_local_max.py:41-74,77-110 with python/test00{1,2}_findpeaks.m: analytic
    signals on t in [0, 2*pi]; expected = top-M of an independent peak finder, values to 5 decimals
    (the reference's tolerance) — here also bit-exact against the oracle."""
    from scipy.signal import find_peaks
    t = 2 * np.pi * np.linspace(0, 1, L)
    if which == 1:
        y = np.sin(3.14 * t) + 0.5 * np.cos(6.09 * t) + 0.1 * np.sin(10.11 * t + 1 / 6) + 0.1 * np.sin(15.3 * t + 1 / 3)
    else:
        y = np.sin(0.25 * 3.14 * t) + 5 * np.sin(6.09 * t) + 0.6 * np.cos(1.11 * t + 1 / 6) + 2 * np.sin(5.3 * t + 1 / 3)
    data = np.abs(y).astype(np.float32)
    got0, got1 = _run(data, M, L, 0.0, 2 * np.pi)
    idx, _ = find_peaks(data.astype(np.float64))
    order = np.argsort(-data[idx], kind="stable")[:M]
    assert np.allclose(got0[0], data[idx][order], atol=1e-5)
    assert np.allclose(np.sort(got1[0])[::-1], np.sort(t[idx][order])[::-1], atol=2 * np.pi / L * 1.01)
    _check(data, M, L, 0.0, 2 * np.pi)


def test_create_rejects_bad_arguments():
    for args in [(0, 64, 0.0, 1.0), (2, 0, 0.0, 1.0), (2, 64, 1.0, 1.0), (17, 64, 0.0, 1.0)]:
        with pytest.raises(doa.DoaError):
            doa.find_local_max(*args)


def test_full_batch_property():
    # 4096 vectors of 1024: port 0 is non-increasing, port 1 is non-increasing, every reported value
    # occurs in its input row
    rng = np.random.default_rng(9)
    v = rng.standard_normal((4096, 1024)).astype(np.float32)
    o0, o1 = _run(v, 3, 1024)
    assert np.all(np.diff(o0, axis=1) <= 0) and np.all(np.diff(o1, axis=1) <= 0)
    assert all(np.isin(o0[i], v[i]).all() for i in range(0, 4096, 97))
    r0, r1 = oracle.find_local_max(v[:256], 3, 1024, 0.0, 180.0)
    assert np.array_equal(o0[:256], r0) and np.array_equal(o1[:256], r1)


def test_fuzz_both_kernels_against_oracle():
    # tests/fuzz_find_local_max.py: random lengths 3..5000 (register kernel, streaming mask kernel, serial kernel),
    # num_max_vals 1..16, smooth / quantised / constant / spiky vectors with NaN and +-inf sprinkled in
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_find_local_max.py")
    spec = importlib.util.spec_from_file_location("fuzz_find_local_max", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(n_cases=250, seed=11, verbose=False) == 0
