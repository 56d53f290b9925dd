"""GPU tests of the long-spectrum scan kernel (2048 < P <= 4096, double; csrc/music_scan_impl.hpp:
music_scan_peak_long_kernel) -- the kernel behind BASELINE.json configs[3] -- and of the lane-blocked form of the peak
pick it runs on its LDS row (csrc/peak_device.hpp: peak_pick_stream<true>).

find_local_max is integer/compare logic: whatever spectrum the scan produced, the peak values and locations must be the
reference's answer ON THAT SPECTRUM bit for bit (lib/find_local_max_impl.cc:93-160; oracle.find_local_max).  The blocked
form splits the vector into one 64-position block per lane, so the cases that matter are the ones where the reference's
"a flat takes the sign of the next non-zero difference to its right" rule has to cross block boundaries:
  * mirror-symmetric spectra (real covariance: Q(psi) = Q(-psi)) put an exact flat on the centre pair P/2-1, P/2, which is
    the last position of one block and the first of the next when P/2 is a multiple of 64;
  * constant spectra (R = c I): every difference is flat, the sign comes from beyond the end of the vector;
  * lengths that do not fill all 64 lanes (P = 2112, 3008).
The spectrum itself is held to the fp64 oracle as everywhere else (tests/test_gpu_music.py tolerances)."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _pipeline(x, N, K, d, M, P, n):
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, max_batch=n)
    mx, am = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    spec = np.full((n, P), -7.0, np.float32)
    cov = np.empty((n, N * N), np.complex64)
    assert pipe.work(n, [x[k] for k in range(N)], mx, am, cov_out=cov, spectrum_out=spec) == n
    return cov, spec, mx, am


@pytest.mark.parametrize("N,M", [(16, 3), (16, 1), (4, 2), (8, 4), (12, 2), (2, 1), (3, 1), (6, 2), (5, 2), (13, 3)])
@pytest.mark.parametrize("P", [2112, 3008, 4096])
def test_peaks_are_the_references_answer_on_the_kernels_own_spectrum(N, M, P):
    n, K, d = 96, 64, 0.5
    th = np.linspace(35.0, 145.0, M) + 3.0
    x = doa.sim.make_streams(N, n * K, list(th), d, snr_db=15.0, seed=100 * N + M)
    cov, spec, mx, am = _pipeline(x, N, K, d, M, P, n)
    o0, o1 = oracle.find_local_max(spec, M, P, 0.0, 180.0)
    assert np.array_equal(mx, o0) and np.array_equal(am, o1)
    # the spectrum against the fp64 evaluation of the reference's formulas on the same covariance items
    s64 = oracle.music_lin_array(cov, d, M, N, P, "f64")
    assert np.array_equal(np.argmax(spec, axis=1), np.argmax(s64, axis=1))
    assert np.abs(spec - s64).max() <= 1e-4
    assert np.all(spec.max(axis=1) == 0.0)


@pytest.mark.parametrize("M", [1, 2, 3])
@pytest.mark.parametrize("P", [2176, 4096])
def test_mirror_symmetric_spectra_put_a_flat_on_a_block_boundary(P, M):
    # small-integer REAL samples: K1 reproduces X X^T / K exactly, every covariance item is exactly real symmetric, so
    # Q(psi) = Q(-psi); the theta grid is float-accumulated and not exactly mirror symmetric, so equal neighbours are
    # not guaranteed at the centre -- what IS guaranteed is many near-ties and, over 64 rows, some exact flats
    N, n, K, d = 4, 64, 4, 0.5
    rng = np.random.default_rng(P + M)
    x = rng.integers(-3, 4, size=(N, n * K)).astype(np.float32)
    x[0] += 4.0
    x = x.astype(np.complex64)
    cov, spec, mx, am = _pipeline(x, N, K, d, M, P, n)
    assert np.all(cov.imag == 0.0)
    o0, o1 = oracle.find_local_max(spec, M, P, 0.0, 180.0)
    assert np.array_equal(mx, o0) and np.array_equal(am, o1)
    flats = int((np.diff(spec, axis=1) == 0.0).sum())
    print(f"P={P} M={M}: exact flats in {n} rows: {flats}")


@pytest.mark.parametrize("N", [4, 16])
@pytest.mark.parametrize("P", [2112, 4096])
def test_constant_and_stepwise_spectra(N, P):
    # R = c I: Q is the same at every angle, the whole row is one flat whose sign comes from beyond the end of the vector
    # (+1), no peak exists, index_max / the fill rule answer with bin 0 -- for every num_max_vals
    n, K, d = 5, 16, 0.5
    x = np.zeros((N, n * K), np.complex64)
    for k in range(N):
        x[k, k::K] = 4.0
    for M in (1, 2, 3):
        cov, spec, mx, am = _pipeline(x, N, K, d, M, P, n)
        assert np.all(spec == 0.0)
        o0, o1 = oracle.find_local_max(spec, M, P, 0.0, 180.0)
        assert np.array_equal(mx, o0) and np.array_equal(am, o1)


def test_full_batch_long_spectra_properties():
    """size-independent properties at BASELINE configs[3]'s full size (4096 snapshots, N = 16, P = 4096): every row's maximum
    is exactly 0 dB and sits where the strongest reported peak is; reported locations are grid values in descending order;
    a 256-row sample equals the oracle's find_local_max on the kernel's own rows."""
    torch = pytest.importorskip("torch")
    N, K, P, M, n, d = 16, 256, 4096, 3, 4096, 0.5
    s, th = doa.sim.make_batch_streams_torch(N, K, n, d, M, 20.0, seed=5, device="cuda")
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, n)
    spec = torch.empty((n, P), dtype=torch.float32, device="cuda")
    mx = torch.empty((n, M), dtype=torch.float32, device="cuda")
    am = torch.empty((n, M), dtype=torch.float32, device="cuda")
    pipe.work_dev(n, [t.data_ptr() for t in s], 0, spec.data_ptr(), mx.data_ptr(), am.data_ptr(), torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert bool((spec.max(dim=1).values == 0).all())
    arg = spec.argmax(dim=1)
    interior = (arg > 0) & (arg < P - 1)
    assert int(interior.sum()) >= n - 8
    assert bool((mx.max(dim=1).values[interior] == 0).all())         # the best peak is the row maximum
    a = am.cpu().numpy()
    assert np.all(np.diff(a, axis=1) <= 0)                            # locations in descending order
    grid = oracle.find_local_max_x_axis(P, 0.0, 180.0)
    assert np.isin(a, grid).all()
    k = 256
    o0, o1 = oracle.find_local_max(spec[:k].cpu().numpy(), M, P, 0.0, 180.0)
    assert np.array_equal(mx[:k].cpu().numpy(), o0) and np.array_equal(a[:k], o1)
