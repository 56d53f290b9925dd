"""GPU parity of K1 (doa.autocorrelate) against the oracle, through the C ABI.

Tolerance: the reference accumulates in fp32 through BLAS cgemm in an unspecified order, so two
correct fp32 implementations differ by rounding only; we require
    |R_hip - R_f64| <= 2e-6 * max|R|   (fp64 evaluation of the same formula = truth), and
    |R_hip - R_oracle_f32| <= 1e-5 * max|R|  (north_star's 1e-5 relative),
and the reference's own QA floor |dR| <= 1.0 (python/qa_autocorrelate.py:82) holds trivially.
"""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _rand_streams(N, T, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((N, T)) + 1j * rng.standard_normal((N, T))).astype(np.complex64)


def _run_block(blk, x, n):
    N = x.shape[0]
    out = np.empty((n, N * N), dtype=np.complex64)
    produced, consumed = blk.general_work(n, [x[k] for k in range(N)], [out])
    assert produced == n
    assert consumed == n * (blk.snapshot_size - blk.overlap_size)
    return out


# (K, overlap, N, FB): the reference's QA configurations first (python/qa_autocorrelate.py:40-48,
# 87-95,133-141), then the benchmark shape, odd sizes, single channel, wide arrays.
CASES = [
    (2048, 512, 4, 0), (1024, 256, 8, 1), (256, 32, 4, 1),
    (1024, 0, 4, 0), (1024, 0, 4, 1), (1000, 1, 4, 1), (77, 10, 3, 1), (129, 0, 5, 0),
    (64, 63, 2, 1), (512, 128, 1, 0), (300, 100, 6, 1), (256, 0, 7, 0),
    (256, 32, 16, 1), (100, 7, 12, 1), (128, 0, 9, 0),
    # read-once piece path (K = q S + r): q >= 2, r == 0, tiny steps, odd r (falls back to the single kernel)
    (1024, 512, 4, 1), (1024, 768, 4, 0), (1000, 700, 3, 1), (96, 80, 8, 1), (100, 90, 2, 0),
    (1024, 1022, 4, 1), (514, 2, 5, 1), (1001, 333, 4, 0), (90, 45, 4, 1),
]


@pytest.mark.parametrize("K,ovl,N,fb", CASES)
def test_autocorrelate_matches_oracle(K, ovl, N, fb):
    n = 9
    S = K - ovl
    x = _rand_streams(N, (n - 1) * S + K, seed=K + 7 * N + fb)
    blk = doa.autocorrelate(N, K, ovl, fb)
    got = _run_block(blk, x, n)
    ref32 = oracle.autocorrelate(x, K, ovl, fb, n)
    ref64 = oracle.autocorrelate(x, K, ovl, fb, n, precision="f64")
    scale = np.abs(ref64).max()
    assert np.abs(got - ref64).max() <= 2e-6 * scale
    assert np.abs(got - ref32).max() <= 1e-5 * scale
    assert np.abs(got - ref32).max() <= 1.0           # reference QA floor


def test_autocorrelate_unaligned_streams_take_scalar_path():
    # stream pointers that are only 8-byte aligned (odd sample offset) must still be exact
    K, ovl, N, n = 256, 64, 4, 5
    S = K - ovl
    base = _rand_streams(N, (n - 1) * S + K + 1, seed=3)
    x = [base[k][1:] for k in range(N)]              # views offset by one complex sample
    blk = doa.autocorrelate(N, K, ovl, 1)
    out = np.empty((n, N * N), dtype=np.complex64)
    # general_work copies to contiguous arrays; go through the raw ABI to keep the odd alignment
    import ctypes as C
    from doa._lib import lib, ptr_array, check
    keep = [np.ascontiguousarray(v) for v in x]
    check(lib.doa_autocorrelate_work(blk._h, n, ptr_array([a.ctypes.data for a in keep]), C.c_void_p(out.ctypes.data)))
    ref = oracle.autocorrelate(np.stack(keep), K, ovl, 1, n, precision="f64")
    assert np.abs(out - ref).max() <= 2e-6 * np.abs(ref).max()


def test_autocorrelate_history_and_scheduler_chunks():
    # GNU Radio semantics: history = overlap+1, zero pre-roll, scheduler-sized calls, consume_each
    K, ovl, N = 128, 32, 4
    S = K - ovl
    n_total = 11
    x_new = _rand_streams(N, n_total * S, seed=5)
    tb = doa.runtime.top_block(max_noutput_items=3)
    blk = doa.autocorrelate(N, K, ovl, 0)
    assert blk.history() == ovl + 1
    assert blk.forecast(5) == 5 * S
    sink = doa.runtime.vector_sink_c(N * N)
    for p in range(N):
        tb.connect((doa.runtime.vector_source_c(x_new[p]), 0), (blk, p))
    tb.connect((blk, 0), (sink, 0))
    tb.run()
    got = sink.data().reshape(-1, N * N)
    ref = oracle.autocorrelate(oracle.gr_history_prepend(x_new, ovl), K, ovl, 0, precision="f64")
    assert got.shape[0] == ref.shape[0] == n_total
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_autocorrelate_hermitian_and_psd():
    # size-independent properties at the benchmark size: R is Hermitian with a real diagonal and
    # x^H R x >= 0; FB output is persymmetric-conjugate up to the reference's 1/K factor
    K, N, n = 1024, 4, 64
    x = _rand_streams(N, n * K, seed=11)
    blk = doa.autocorrelate(N, K, 0, 0)
    R = _run_block(blk, x, n).reshape(n, N, N).transpose(0, 2, 1)   # column-major items -> [a, b]
    assert np.abs(R - R.conj().transpose(0, 2, 1)).max() == 0.0
    assert np.abs(np.diagonal(R, axis1=1, axis2=2).imag).max() == 0.0
    w = np.linalg.eigvalsh(R.astype(np.complex128))
    assert w.min() > -1e-5 * w.max()
    # linearity in power: scaling the input by 2 scales R by exactly 4
    R2 = _run_block(blk, (2 * x).astype(np.complex64), n).reshape(n, N, N).transpose(0, 2, 1)
    assert np.array_equal(R2, 4 * R)


def test_autocorrelate_create_rejects_bad_arguments():
    for args in [(0, 16, 0, 0), (4, 0, 0, 0), (4, 16, 16, 0), (4, 16, -1, 0), (17, 16, 0, 0)]:
        with pytest.raises(doa.DoaError):
            doa.autocorrelate(*args)


def test_autocorrelate_zero_items():
    blk = doa.autocorrelate(4, 64, 0, 0)
    out = np.empty((0, 16), dtype=np.complex64)
    x = _rand_streams(4, 64, 1)
    produced, consumed = blk.general_work(0, [x[k] for k in range(4)], [np.empty((1, 16), np.complex64)])
    assert produced == 0 and consumed == 0
