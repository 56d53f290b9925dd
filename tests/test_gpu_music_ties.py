"""GPU tests of the arg-max tie rule of MUSIC_lin_array -> find_local_max.

Reference semantics (lib/MUSIC_lin_array_impl.cc:140-142, lib/find_local_max_impl.h:53-56): out = 1.0/Q is a
correctly rounded float, the spectrum is 10*log10(out/max(out)), so an angle is at exactly 0 dB iff its
rounded reciprocal EQUALS the largest one, and the reported angle is the first such bin.  Every scan kernel
(general register-resident, lean num_max_vals = 1, lean MULTI, long-spectrum streaming) must produce exactly
that tie set -- no wider window, no artificial flats -- on data with genuine near-ties: a real symmetric
covariance gives Q(psi) = Q(-psi), so mirror bins of the angle grid hold values a few ulp apart.
All internal arithmetic is double and differs between kernels at the 1e-15 level only, so after rounding Q to
float every kernel sees the same values: their outputs must agree bit for bit with the general kernel's, whose
Q is observable through doa_MUSIC_lin_array_debug."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu

N = 4


def _real_streams(seed, n_items, K=4):
    """Small-integer REAL samples: K1 reproduces X X^T / K exactly, so every covariance item is exactly real
    symmetric (u_l real, Q even in psi)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-3, 4, size=(N, n_items * K)).astype(np.float32)
    x[0] += 4.0                                                     # a dominant direction: distinct eigenvalues
    return x.astype(np.complex64)


def _check_against_own_q(spec, q, M, P, vals, locs):
    want = oracle.music_db_from_q(q, "f32")                        # the reference's rule on the device's own Q
    tie = np.flatnonzero(want == 0.0)
    zero = np.flatnonzero(spec == 0.0)
    assert spec.max() == 0.0 and np.array_equal(zero, tie), (zero, tie)
    rest = np.ones(P, bool)
    rest[tie] = False
    assert np.all(spec[rest] < 0.0)
    assert np.all(np.abs(spec[rest] - want[rest]) <= 4e-6 + 4e-7 * np.abs(want[rest]))
    o0, o1 = oracle.find_local_max(spec[None, :], M, P, 0.0, 180.0)
    assert np.array_equal(vals, o0[0]) and np.array_equal(locs, o1[0])
    if M == 1:
        assert locs[0] == oracle.find_local_max_x_axis(P, 0.0, 180.0)[tie[0]]      # first bin holding the maximum
    return len(tie)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("P", [256, 1024, 4096])
def test_tie_set_is_that_of_the_rounded_reciprocal(P, M):
    n, K, d = 48, 4, 0.5
    x = _real_streams(P + M, n, K)
    R = oracle.autocorrelate(x, K, 0, 0, n)
    assert np.all(R.imag == 0.0)
    blk = doa.MUSIC_lin_array(d, M, N, P)
    spec = np.empty((n, P), np.float32)
    assert blk.work(n, [R], [spec]) == n                            # general kernel (P <= 2048) / streaming kernel
    _pn, q = blk.debug(R)                                           # Q of the general kernel
    f = doa.find_local_max(M, P, 0.0, 180.0)
    v0, v1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    f.work(n, [spec], [v0, v1])
    n_multi = 0
    for i in range(n):
        n_multi += _check_against_own_q(spec[i], q[i], M, P, v0[i], v1[i]) > 1
    # the same through the pipeline: lean kernel (M = 1), lean MULTI (M = 2), streaming scan + K5 (P = 4096)
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, max_batch=n)
    p0, p1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    pspec = np.empty((n, P), np.float32)
    pcov = np.empty((n, N * N), np.complex64)
    pipe.work(n, [x[k] for k in range(N)], p0, p1, cov_out=pcov, spectrum_out=pspec)
    assert np.array_equal(pcov, R)                                  # K1 is exact on these samples
    for i in range(n):
        assert np.array_equal(np.flatnonzero(pspec[i] == 0.0), np.flatnonzero(spec[i] == 0.0)), i
        assert np.all(np.abs(pspec[i] - spec[i]) <= 4e-6 + 4e-7 * np.abs(spec[i]))     # two v_rcp/v_log routes to one value
    q0, q1 = oracle.find_local_max(pspec, M, P, 0.0, 180.0)          # K5 fused into the scan: bit for bit on ITS spectrum
    assert np.array_equal(p0, q0) and np.array_equal(p1, q1)
    if M == 1:
        assert np.array_equal(p1, v1)                               # same tie set => same first bin
    print(f"P={P} M={M}: items with more than one bin at the maximum: {n_multi}/{n}")


@pytest.mark.parametrize("M", [1, 2, 3])
@pytest.mark.parametrize("P", [256, 1024, 4096])
def test_constant_null_spectrum_ties_everywhere(P, M):
    # R = c I (and any matrix with N equal eigenvalues): ranks go by index, P_N = diag(1, .., 1, 0, ..), Q = N - M at
    # every angle: every bin ties, the spectrum is 0 dB everywhere and index_max / the fill rule pick bin 0
    d, n, K = 0.5, 6, 4
    x = np.zeros((N, n * K), np.complex64)
    for k in range(N):
        x[k, k::K] = np.sqrt(np.float32(K)) * 2.0                   # orthogonal impulses: R = 4 I exactly
    R = oracle.autocorrelate(x, K, 0, 0, n)
    assert np.array_equal(R, np.tile((4.0 * np.eye(N)).reshape(1, -1), (n, 1)).astype(np.complex64))
    blk = doa.MUSIC_lin_array(d, M, N, P)
    spec = np.full((n, P), -1.0, np.float32)
    blk.work(n, [R], [spec])
    assert np.all(spec == 0.0)
    pipe = doa.music_pipeline(N, K, 0, 0, d, M, P, max_batch=n)
    p0, p1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    pspec = np.full((n, P), -1.0, np.float32)
    pipe.work(n, [x[k] for k in range(N)], p0, p1, spectrum_out=pspec)
    assert np.all(pspec == 0.0)
    o0, o1 = oracle.find_local_max(pspec, M, P, 0.0, 180.0)
    assert np.array_equal(p0, o0) and np.array_equal(p1, o1)
    s32 = oracle.music_lin_array(R, d, M, N, P, "f32")
    assert np.abs(s32).max() <= 1e-5                                # (the oracle's own a^H P a carries rounding noise)
