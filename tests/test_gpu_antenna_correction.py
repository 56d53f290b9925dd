"""GPU parity of doa.antenna_correction (the block in front of autocorrelate),
standalone and fused into K1.  The reference has no QA test for this block; the oracle restates
lib/antenna_correction_impl.cc:56-73,85-99."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, gains, phases, name="antenna.cfg"):
    p = tmp_path / name
    p.write_text("".join(f"{g} {ph}\n" for g, ph in zip(gains, phases)))     # save_antenna_calib.py:70 format
    return str(p)


def _streams(N, T, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((N, T)) + 1j * rng.standard_normal((N, T))).astype(np.complex64)


@pytest.mark.parametrize("N,T", [(4, 4096), (4, 1001), (8, 513), (1, 64), (16, 2048)])
def test_standalone_block_matches_oracle(tmp_path, N, T):
    rng = np.random.default_rng(N * T)
    gains, phases = rng.uniform(0.5, 2.0, N), rng.uniform(-3.0, 3.0, N)
    path = _cfg(tmp_path, gains, phases)
    blk = doa.antenna_correction(N, path)
    g_ref = oracle.antenna_correction_gains(open(path).read(), N)
    assert np.abs(blk.gains() - g_ref).max() <= 2e-7 * np.abs(g_ref).max()      # libm cosf/sinf vs numpy: 1 ulp
    x = _streams(N, T, 1)
    out = [np.empty(T, np.complex64) for _ in range(N)]
    assert blk.work(T, [x[k] for k in range(N)], out) == T
    ref = oracle.antenna_correction(x, blk.gains())                              # same gains: isolates the multiply
    got = np.stack(out)
    assert np.abs(got - ref).max() <= 2.5e-7 * np.abs(ref).max()


def test_flowgraph_with_correction_block(tmp_path):
    N, K, ovl = 4, 256, 64
    path = _cfg(tmp_path, [1.0, 0.8, 1.3, 0.9], [0.0, 0.4, -1.1, 2.0])
    x = _streams(N, 10 * (K - ovl), 2)
    tb = doa.runtime.top_block(max_noutput_items=4)
    corr = doa.antenna_correction(N, path)
    ac = doa.autocorrelate(N, K, ovl, 1)
    sink = doa.runtime.vector_sink_c(N * N)
    for p in range(N):
        tb.connect((doa.runtime.vector_source_c(x[p]), 0), (corr, p))
        tb.connect((corr, p), (ac, p))
    tb.connect((ac, 0), (sink, 0))
    tb.run()
    got = sink.data().reshape(-1, N * N)
    xc = oracle.antenna_correction(x, corr.gains())
    ref = oracle.autocorrelate(oracle.gr_history_prepend(xc, ovl), K, ovl, 1, precision="f64")
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("N,K,ovl,fb", [(4, 1024, 0, 0), (4, 2048, 512, 1), (8, 256, 32, 1), (3, 77, 10, 1), (16, 256, 0, 1), (12, 100, 7, 0)])
def test_fused_into_autocorrelate_equals_block_chain(tmp_path, N, K, ovl, fb):
    rng = np.random.default_rng(N + K)
    path = _cfg(tmp_path, rng.uniform(0.5, 2.0, N), rng.uniform(-3.0, 3.0, N))
    corr = doa.antenna_correction(N, path)
    n, S = 7, K - ovl
    x = _streams(N, (n - 1) * S + K, 3)
    fused = doa.autocorrelate(N, K, ovl, fb)
    fused.fuse_antenna_correction(corr)
    Rf = np.empty((n, N * N), np.complex64)
    fused.general_work(n, [x[k] for k in range(N)], [Rf])
    ref = oracle.autocorrelate(oracle.antenna_correction(x, corr.gains()).astype(np.complex128), K, ovl, fb, n, precision="f64")
    assert np.abs(Rf - ref).max() <= 3e-6 * np.abs(ref).max()
    fused.fuse_antenna_correction(None)                         # and it can be removed again
    R0 = np.empty((n, N * N), np.complex64)
    fused.general_work(n, [x[k] for k in range(N)], [R0])
    ref0 = oracle.autocorrelate(x, K, ovl, fb, n, precision="f64")
    assert np.abs(R0 - ref0).max() <= 2e-6 * np.abs(ref0).max()


def test_config_file_errors(tmp_path):
    # the reference throws std::invalid_argument with these texts (lib/antenna_correction_impl.cc:58-73)
    with pytest.raises(ValueError, match="Cannot find configuration file."):
        doa.antenna_correction(4, str(tmp_path / "missing.cfg"))
    with pytest.raises(ValueError, match="too many inputs"):
        doa.antenna_correction(2, _cfg(tmp_path, [1, 1, 1], [0, 0, 0], "many.cfg"))
    with pytest.raises(ValueError, match="does not have enough inputs"):
        doa.antenna_correction(4, _cfg(tmp_path, [1, 1], [0, 0], "few.cfg"))
    assert doa.antenna_correction(2, _cfg(tmp_path, [2.0, 4.0], [0.0, 0.0], "ok.cfg")).gains().tolist() == [0.5 + 0j, 0.25 + 0j]
