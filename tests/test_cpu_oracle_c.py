"""CPU tests of the plain-C oracle (oracle/doa_oracle.c, the cpu_baseline "port") against the
numpy/LAPACK oracle, in both of its modes (built-in Jacobi / loops, and bound to the BLAS/LAPACK
routines Armadillo forwards to)."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

import doa_oracle as oracle
from test_cpu_oracle_pins import sim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liboracle_doa.so")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make"], cwd=os.path.join(ROOT, "oracle"), check=True, capture_output=True)
    L = ctypes.CDLL(SO)
    L.oracle_set_num_threads(2)
    return L


def _openblas():
    import scipy
    c = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    return c[0] if c else None


def _pipeline(lib, x, N, K, ovl, avg, d, M, P, n):
    ptrs = (ctypes.c_void_p * N)(*[x[k].ctypes.data for k in range(N)])
    R = np.empty((n, N * N), np.complex64)
    spec = np.empty((n, P), np.float32)
    mv = np.empty((n, M), np.float32)
    am = np.empty((n, M), np.float32)
    vp = ctypes.c_void_p
    rc = lib.oracle_music_pipeline(ptrs, N, K, ovl, avg, ctypes.c_float(d), M, P, n, vp(R.ctypes.data),
                                   vp(spec.ctypes.data), vp(mv.ctypes.data), vp(am.ctypes.data))
    assert rc == n
    return R, spec, mv, am


CASES = [(4, 1024, 0, 0, 0.5, 1, 1024, [57.3]), (4, 2048, 512, 1, 0.4, 2, 1024, [30.0, 123.0]),
         (8, 256, 32, 1, 0.4, 1, 1024, [23.0]), (16, 256, 32, 1, 0.5, 3, 512, [40.0, 90.0, 121.0])]


@pytest.mark.parametrize("mode", ["builtin", "lapack"])
@pytest.mark.parametrize("N,K,ovl,avg,d,M,P,th", CASES)
def test_c_pipeline_matches_numpy_oracle(lib, mode, N, K, ovl, avg, d, M, P, th):
    if mode == "lapack":
        path = _openblas()
        if path is None:
            pytest.skip("scipy's bundled OpenBLAS not found")
        os.environ["OPENBLAS_NUM_THREADS"] = "1"
        assert lib.oracle_use_lapack(path.encode(), b"scipy_") == 0
    else:
        lib.oracle_use_lapack(None, None)
    n = 6
    x = np.ascontiguousarray(sim.make_streams(N, (n - 1) * (K - ovl) + K, th, d, snr_db=15.0, seed=N + K))
    R0, s0, v0, l0 = oracle.music_pipeline(x, K, ovl, avg, d, M, P)
    R1, s1, v1, l1 = _pipeline(lib, x, N, K, ovl, avg, d, M, P, n)
    lib.oracle_use_lapack(None, None)
    assert np.abs(R1 - R0).max() <= 1e-5 * np.abs(R0).max()
    assert np.array_equal(l1, l0)                       # same peak bins
    # the dB spectra agree up to the normalisation noise both fp32 paths carry
    assert np.abs(s1 - s0).max() <= 0.5
    d01 = (s1 - s0)
    away = s0 < -10.0                                    # away from the nulls of Q (= peaks of the spectrum)
    assert np.all(np.abs(d01 - np.median(d01, axis=1, keepdims=True))[away] <= 0.02)


def test_c_find_local_max_bit_exact(lib):
    rng = np.random.default_rng(0)
    vp = ctypes.c_void_p
    for L, M in [(1024, 1), (1024, 3), (512, 5), (37, 2), (4096, 4)]:
        for kind in range(3):
            v = rng.standard_normal((30, L)).astype(np.float32)
            if kind == 1:
                v = np.round(v * 1.5).astype(np.float32)
            if kind == 2:
                v = rng.integers(0, 2, size=(30, L)).astype(np.float32)
            o0 = np.empty((30, M), np.float32)
            o1 = np.empty((30, M), np.float32)
            rc = lib.oracle_find_local_max(vp(v.ctypes.data), 30, M, L, ctypes.c_float(0.0), ctypes.c_float(180.0),
                                           vp(o0.ctypes.data), vp(o1.ctypes.data))
            assert rc == 30
            r0, r1 = oracle.find_local_max(v, M, L, 0.0, 180.0)
            assert np.array_equal(o0.view(np.uint32), r0.view(np.uint32)), (L, M, kind)
            assert np.array_equal(o1.view(np.uint32), r1.view(np.uint32)), (L, M, kind)


def test_c_tables_match_numpy(lib):
    N, P, d = 8, 1000, 0.4
    loc = np.empty(N, np.float32)
    th = np.empty(P, np.float32)
    A = np.empty(N * P, np.complex64)
    vp = ctypes.c_void_p
    lib.oracle_music_tables(ctypes.c_float(d), N, P, vp(loc.ctypes.data), vp(th.ctypes.data), vp(A.ctypes.data))
    assert np.array_equal(loc, oracle.music_array_loc(d, N))
    assert np.array_equal(th, oracle.music_theta_grid(P))
    assert np.abs(A.reshape(P, N).T - oracle.music_steering(d, N, P, "f32")).max() <= 3e-7
