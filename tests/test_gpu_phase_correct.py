"""GPU parity of doa.phase_correct_hier (the per-stream phase correction that sits in front of
autocorrelate in the reference's X310 flowgraphs), standalone and folded into K1.  The oracle restates
python/phase_correct_hier.py:33-45 (the file parser with its quirk) and :86-104 (copy + multiply_const_vcc)."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _streams(N, T, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((N, T)) + 1j * rng.standard_normal((N, T))).astype(np.complex64)


@pytest.mark.parametrize("N,T", [(4, 4096), (2, 1001), (8, 513), (16, 64)])
def test_block_matches_oracle(tmp_path, N, T):
    rng = np.random.default_rng(N + T)
    phases = rng.uniform(-3.0, 3.0, N - 1)
    path = tmp_path / "phases.cfg"
    path.write_text("# measured offsets\n" + "".join(f"{float(p)!r}\n" for p in phases) + "\n")
    blk = doa.phase_correct_hier(N, str(path))
    assert blk.phases == oracle.phase_correct_phases(path.read_text())
    x = _streams(N, T, 3)
    out = [np.empty(T, np.complex64) for _ in range(N)]
    assert blk.work(T, [x[k] for k in range(N)], out) == T
    ref = oracle.phase_correct(x, blk.phases)
    got = np.stack(out)
    assert np.array_equal(got[0], x[0])                                   # port 0 is a copy
    assert np.abs(got - ref).max() <= 2.5e-7 * np.abs(ref).max()         # complex<float> product, fma vs separate rounding


def test_folded_into_the_covariance_kernel(tmp_path):
    N, K, ovl, n = 4, 512, 128, 9
    path = tmp_path / "p.cfg"
    path.write_text("0.7\n-2.1\n1.3\n")
    blk = doa.phase_correct_hier(N, str(path))
    x = _streams(N, (n - 1) * (K - ovl) + K, 5)
    ac = doa.autocorrelate(N, K, ovl, 1)
    ac.fuse_antenna_correction(blk.gains())
    R = np.empty((n, N * N), np.complex64)
    assert ac.general_work(n, [x[k] for k in range(N)], [R])[0] == n
    ref = oracle.autocorrelate(oracle.phase_correct(x, blk.phases), K, ovl, 1, n, precision="f64")
    assert np.abs(R - ref).max() <= 3e-6 * np.abs(ref).max()


def test_reference_parser_quirks_and_errors(tmp_path):
    p = tmp_path / "q.cfg"
    p.write_text("phase 1\n0.5\n0\n  -1.25  \nnan\n1e-3\n0.0\n")          # "0" and "0.0" parse to a falsy float: dropped
    assert doa.read_phase_config(str(p)) [:2] == [0.5, -1.25]
    got, want = doa.read_phase_config(str(p)), oracle.phase_correct_phases(p.read_text())
    assert len(got) == len(want) == 4 and got[:2] == want[:2] and np.isnan(got[2]) and np.isnan(want[2]) and got[3] == want[3]
    with pytest.raises(ValueError, match="Not valid number of phase estimates"):
        doa.phase_correct_hier(3, str(p))
    with pytest.raises(ValueError, match="not valid"):
        doa.phase_correct_hier(2, str(tmp_path / "missing.cfg"))
