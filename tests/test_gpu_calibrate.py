"""GPU parity of doa.calibrate_lin_array against the oracle's literal two-EVD
restatement of lib/calibrate_lin_array_impl.cc:98-134, up to the unit-modulus factor LAPACK leaves
arbitrary (both sides rotated so that element 0 is real >= 0), and against the reference's QA check
(python/qa_calibrate_lin_array.py:40-91: the ratio true-perturbation / estimate is the same for all
antennas to 1 decimal)."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _pilot_covariances(N, d, pilot_deg, K, n, seed, snr_db=30.0):
    rng = np.random.default_rng(seed)
    gains = np.concatenate([[1.0], rng.uniform(0.3, 1.0, N - 1)])                # music_test_input_gen.m:42-49
    phases = np.concatenate([[1.0], np.exp(-1j * np.pi * rng.uniform(0, 1, N - 1))])
    pert = gains * phases
    x = doa.sim.make_streams(N, n * K, [pilot_deg], d, snr_db=None, seed=seed, freqs=[1.0 / 6.0])
    x = (pert[:, None] * x)
    x = x + 10 ** (-snr_db / 20) * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape)) / np.sqrt(2)
    return pert, oracle.autocorrelate(x.astype(np.complex64), K, 0, 0, n)


@pytest.mark.parametrize("N,d,pilot", [(4, 0.3, 30.0), (8, 0.5, 60.0), (2, 0.5, 45.0), (5, 0.4, 100.0), (16, 0.5, 75.0)])
def test_calibrate_matches_oracle_and_reference_qa(N, d, pilot):
    pert, R = _pilot_covariances(N, d, pilot, 1024, 10, seed=N)
    blk = doa.calibrate_lin_array(d, N, pilot)
    est = np.empty((R.shape[0], N), np.complex64)
    assert blk.work(R.shape[0], [R], [est]) == R.shape[0]
    ref64 = oracle.calibrate_normalise(oracle.calibrate_lin_array(R, d, N, pilot, "f64"))
    ref32 = oracle.calibrate_normalise(oracle.calibrate_lin_array(R, d, N, pilot, "f32"))
    assert np.all(est[:, 0].imag == 0) and np.all(est[:, 0].real >= 0)
    assert np.abs(np.linalg.norm(est, axis=1) - 1).max() <= 1e-6
    dev32 = np.abs(ref32 - ref64).max()
    assert np.abs(est - ref64).max() <= 2e-6 + 2 * dev32          # float pilot table (as the reference) vs the fp64 formula
    assert np.abs(est - ref32).max() <= 2e-6 + 2 * dev32
    # reference QA: ant_pert_vec ./ estimate has equal entries (diff ~ 0 to 1 decimal)
    ratio = pert[None, :] / est
    assert np.abs(np.diff(ratio, axis=1)).max() <= 0.05 * np.abs(ratio).max()


def test_calibrate_create_rejects_bad_arguments():
    for args in [(0.5, 1, 30.0), (0.6, 4, 30.0), (0.5, 17, 30.0)]:
        with pytest.raises(doa.DoaError):
            doa.calibrate_lin_array(*args)
