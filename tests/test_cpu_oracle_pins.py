"""CPU tests that pin the oracle (oracle/doa_oracle.py) to what the reference's own tests fix:
their deterministic find_local_max signals, their QA scenarios and tolerances for autocorrelate /
MUSIC / Root-MUSIC, and the constructor-time tables.  No GPU, no product code."""
import importlib.util
import os

import numpy as np
import pytest

import doa_oracle as oracle

# the signal generator is product code without device dependencies; load it without importing the
# package (whose import requires the built HIP library)
_spec = importlib.util.spec_from_file_location(
    "doa_sim_standalone", os.path.join(os.path.dirname(__file__), "..", "gr-doa_amd", "python", "doa", "sim.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


# ---- find_local_max: the reference's fully deterministic QA signals ---------------------------------
@pytest.mark.parametrize("which,L,M", [(1, 2 ** 11, 3), (2, 2 ** 12, 5)])
def test_find_local_max_reference_signals(which, L, M):
    """python/qa_find_local_max.py:41-74,77-110; python/test001_findpeaks.m:5-12, test002: expected
    = top-M of findpeaks() by height, compared to 5 decimals (the reference's assertAlmostEqual)."""
    from scipy.signal import find_peaks
    t = 2 * np.pi * np.linspace(0, 1, L)
    if which == 1:
        y = np.sin(3.14 * t) + 0.5 * np.cos(6.09 * t) + 0.1 * np.sin(10.11 * t + 1 / 6) + 0.1 * np.sin(15.3 * t + 1 / 3)
    else:
        y = np.sin(0.25 * 3.14 * t) + 5 * np.sin(6.09 * t) + 0.6 * np.cos(1.11 * t + 1 / 6) + 2 * np.sin(5.3 * t + 1 / 3)
    data = np.abs(y)
    idx, _ = find_peaks(data)
    order = np.argsort(-data[idx], kind="stable")[:M]
    vals, locs = oracle.find_local_max(data.astype(np.float32), M, L, 0.0, 2 * np.pi)
    np.testing.assert_allclose(vals[0], data[idx][order], atol=1e-5)
    # locations: the block's x axis is x_min + i*(x_max-x_min)/L, the .m file's is linspace(0,1,L):
    # they differ by i*2pi/(L(L-1)) <= one grid step
    np.testing.assert_allclose(np.sort(locs[0])[::-1], np.sort(t[idx][order])[::-1], atol=2 * np.pi / L * 1.01)


def test_find_local_max_ports_are_sorted_independently():
    # port 0: descending value; port 1: descending location — not index-aligned (:186-188)
    L = 64
    v = np.zeros(L, np.float32)
    v[10], v[40] = 5.0, 9.0
    vals, locs = oracle.find_local_max(v, 2, L, 0.0, 64.0)
    assert list(vals[0]) == [9.0, 5.0]
    assert list(locs[0]) == [40.0, 10.0]
    v[10], v[40] = 9.0, 5.0
    vals, locs = oracle.find_local_max(v, 2, L, 0.0, 64.0)
    assert list(vals[0]) == [9.0, 5.0] and list(locs[0]) == [40.0, 10.0]


def test_find_local_max_flat_rules():
    # plateau followed by a fall is a peak at its first sample; plateau followed by a rise is not;
    # end points never are (:92-114)
    v = np.array([0, 1, 2, 2, 2, 1, 0, 3, 3, 4, 4], np.float32)
    vals, locs = oracle.find_local_max(v, 2, v.size, 0.0, float(v.size))
    assert vals[0, 0] == 2.0 and locs[0].max() == 2.0
    # only one peak for M=2 -> the fill index is the best peak's position in the peak list (0)
    assert vals[0, 1] == v[0]


# ---- autocorrelate --------------------------------------------------------------------------------
@pytest.mark.parametrize("K,ovl,N,fb", [(2048, 512, 4, 0), (1024, 256, 8, 1), (256, 32, 4, 1)])
def test_autocorrelate_reference_qa_configs(K, ovl, N, fb):
    """python/qa_autocorrelate.py:40-48,87-95,133-141: complex unit-normal input; the reference only
    requires |expected - observed| <= 1.0 per element (:82); the fp32 and fp64 oracle paths agree far
    inside that, and the window placement follows the block (history pre-roll), not the .m file."""
    rng = np.random.default_rng(K + N)
    n = 20
    S = K - ovl
    x = (rng.standard_normal((N, n * S)) + 1j * rng.standard_normal((N, n * S))).astype(np.complex64)
    xh = oracle.gr_history_prepend(x, ovl)
    R32 = oracle.autocorrelate(xh, K, ovl, fb)
    R64 = oracle.autocorrelate(xh, K, ovl, fb, precision="f64")
    assert R32.shape == (n, N * N)
    assert np.abs(R32 - R64).max() <= 1e-5 * np.abs(R64).max()
    # Octave model (examples/@wpi_twinrx_doa_testbench/autocorrelate.m:36-44) on the window the block
    # uses: transpose(x)*conj(x)/K, FB with its second division by K
    i = 3
    w = xh[:, i * S:i * S + K].astype(np.complex128).T
    S_x = w.T @ np.conj(w) / K
    if fb:
        J = np.fliplr(np.eye(N))
        S_x = 0.5 * S_x + 0.5 * J @ np.conj(S_x) @ J / K
    got = R64[i].reshape(N, N, order="F")
    assert np.abs(got - S_x).max() <= 1e-6
    assert np.abs(R32[i].reshape(N, N, order="F") - S_x).max() <= 1.0
    # first window starts with `overlap` zeros (set_history(overlap+1), lib/autocorrelate_impl.cc:57)
    assert np.all(xh[:, :ovl] == 0)


# ---- MUSIC tables -----------------------------------------------------------------------------------
def test_theta_grid_float_accumulation():
    """lib/MUSIC_lin_array_impl.cc:64-72: exact for power-of-two lengths, drifts for P=1000."""
    for P in (1024, 4096):
        th = oracle.music_theta_grid(P)
        np.testing.assert_array_equal(th, (np.pi * (np.arange(P) * (180.0 / P)) / 180.0).astype(np.float32))
    th = np.rad2deg(oracle.music_theta_grid(1000).astype(np.float64))
    drift = np.abs(th - np.arange(1000) * 0.18).max()
    assert 1e-4 < drift < 3e-3


def test_steering_matches_octave_manifold():
    """examples/@wpi_twinrx_doa_testbench/wpi_twinrx_doa_testbench.m:60-64:
    amv(theta) = exp(-1i*2*pi*cos(theta)*array_loc), array_loc = d*((N-1)/2:-1:-(N-1)/2)'."""
    N, d, P = 8, 0.4, 1024
    A32 = oracle.music_steering(d, N, P, "f32")
    A64 = oracle.music_steering(d, N, P, "f64")
    loc = np.float32(d).astype(np.float64) * ((N - 1) / 2.0 - np.arange(N))
    theta = oracle.music_theta_grid(P).astype(np.float64)
    ref = np.exp(-1j * 2 * np.pi * np.cos(theta)[None, :] * loc[:, None])
    assert np.abs(A64 - ref).max() <= 1e-14
    assert np.abs(A32 - ref).max() <= 2e-6


# ---- MUSIC / Root-MUSIC: the reference's QA scenarios and 2.0 degree tolerance -------------------------
MUSIC_QA = [dict(N=8, d=0.4, th=23.0, K=256, ovl=32), dict(N=16, d=0.5, th=121.0, K=256, ovl=32)]
ROOT_QA = [dict(N=8, d=0.5, th=23.0, K=256, ovl=32), dict(N=4, d=0.5, th=52.0, K=1024, ovl=64)]


@pytest.mark.parametrize("c", MUSIC_QA)
@pytest.mark.parametrize("snr", [None, 10.0])
def test_music_oracle_reference_qa(c, snr):
    """python/qa_MUSIC_lin_array.py:46-99,102-155 (FB on, P=1024, 1 source): every snapshot's
    find_local_max arg-max within 2.0 deg; fp32 and fp64 oracle paths within one grid bin."""
    n, P = 10, 1024
    S = c["K"] - c["ovl"]
    x = sim.make_streams(c["N"], (n - 1) * S + c["K"], [c["th"]], c["d"], snr_db=snr, seed=5)
    R = oracle.autocorrelate(x, c["K"], c["ovl"], 1)
    for prec in ("f32", "f64"):
        spec = oracle.music_lin_array(R, c["d"], 1, c["N"], P, prec)
        assert np.all(spec.max(axis=1) == 0.0)
        _, loc = oracle.find_local_max(spec.astype(np.float32), 1, P, 0.0, 180.0)
        assert np.all(np.abs(loc - c["th"]) <= 2.0)
        assert np.all(np.abs(loc - c["th"]) <= 180.0 / P + (0.6 if snr is not None else 0.0))


@pytest.mark.parametrize("c", ROOT_QA)
@pytest.mark.parametrize("snr", [None, 10.0])
def test_root_music_oracle_reference_qa(c, snr):
    """python/qa_rootMUSIC_linear_array.py:41-91,94-145: within 2.0 deg."""
    n = 10
    S = c["K"] - c["ovl"]
    x = sim.make_streams(c["N"], (n - 1) * S + c["K"], [c["th"]], c["d"], snr_db=snr, seed=6)
    R = oracle.autocorrelate(x, c["K"], c["ovl"], 1)
    a64 = oracle.root_music(R, c["d"], 1, c["N"], "f64")
    assert np.all(np.abs(a64 - c["th"]) <= (2.0 if snr is not None else 1e-2))
    if snr is not None:   # the float path is only meaningful on noisy data (near-double roots, H2)
        a32 = oracle.root_music(R, c["d"], 1, c["N"], "f32")
        assert np.all(np.abs(a32 - c["th"]) <= 2.0)
        assert np.abs(a32 - a64).max() <= 0.2


def test_simulation_flowgraph_scenario():
    """apps/run_MUSIC_lin_array_simulation.grc: 4 elements, sources at 30 and 123 deg, d=0.4, K=2048,
    ovl=512, FB, P=1024, noise added per source before the manifold -> both peaks on their bins."""
    N, K, ovl, P = 4, 2048, 512, 1024
    n = 6
    x = sim.make_streams(N, (n - 1) * (K - ovl) + K, [30.0, 123.0], 0.4, snr_db=None, seed=1,
                         per_source_noise=[5e-5, 5e-3])
    _, spec, vals, locs = oracle.music_pipeline(x, K, ovl, 1, 0.4, 2, P, precision="f64")
    assert np.all(np.abs(np.sort(locs, axis=1) - np.array([30.0, 123.0])[None, :]) <= 0.2)
    assert np.all(np.diff(vals, axis=1) <= 0) and np.all(np.diff(locs, axis=1) <= 0)


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors (philox4x32 10 rounds): zero, all-ones and the pi-digits vectors
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = oracle.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert [int(v) for v in got] == list(want)


def test_sim_source_oracle_has_the_flowgraph_statistics():
    # the generator restatement against the model it claims: tones through the manifold give the
    # covariance A diag(p) A^H (+ noise terms), and the noise streams are unit-variance, uncorrelated
    N, d, thetas, freqs = 4, 0.4, [30.0, 123.0], [0.03125, 0.0625]
    T = 1 << 15
    x = oracle.sim_source(N, d, thetas, freqs, T, None, [0.05, 0.1], 0.2, seed=11).astype(np.complex128)
    R = x @ x.conj().T / T
    th = np.deg2rad(np.array(thetas))
    loc = d * ((N - 1) / 2.0 - np.arange(N))
    A = np.exp(-2j * np.pi * np.cos(th)[None, :] * loc[:, None])
    p = np.array([1 + 2 * 0.05 ** 2, 1 + 2 * 0.1 ** 2])          # tone power + complex noise of variance 2 a^2
    want = A @ np.diag(p) @ A.conj().T + 0.2 ** 2 * np.eye(N)
    assert np.abs(R - want).max() <= 0.02
    g = oracle.sim_noise_stream(11, 3, 0, 1 << 16)
    assert abs(g.real.std() - 1) < 0.01 and abs(g.imag.std() - 1) < 0.01 and abs(np.mean(g * g)) < 0.02
    h = oracle.sim_noise_stream(11, 4, 0, 1 << 16)
    assert abs(np.mean(g * h.conj())) < 0.02
    # independent of where a range starts
    assert np.array_equal(oracle.sim_noise_stream(11, 3, 1000, 64), g[1000:1064])


def test_compass_mean_oracle():
    a = np.array([[10.0, 100.0], [20.0, 110.0], [60.0, 150.0]], np.float32)
    assert np.array_equal(oracle.compass_mean(a, 2), np.array([30.0, 120.0], np.float32))
    assert np.isnan(oracle.compass_mean(np.empty((0, 3), np.float32), 3)).all()


# ---- the reference's golden model (Octave, examples/@wpi_twinrx_doa_testbench) restated in numpy -------
# The reference's QA tests compare the C++ blocks against these .m files at run time (through oct2py,
# python/qa_MUSIC_lin_array.py:63-71, python/qa_rootMUSIC_linear_array.py:58-66).  Octave is not
# installed, so the .m files are restated here line by line (double precision, as Octave computes)
# and the oracle's fp64 path is held to them.
def _octave_amv(theta, d, N):
    loc = d * ((N - 1) / 2.0 - np.arange(N))                       # wpi_twinrx_doa_testbench.m:60-64
    return np.exp(-1j * 2 * np.pi * np.cos(theta) * loc)


def _octave_music(S_x, d, N, M, P, theta=None):
    """MUSIC.m:21-52 for one snapshot: theta = 0:180/P:180-180/P; [V,~] = eig(S_x) (ascending for a
    Hermitian matrix); U_N = V(:,1:N-M); Q = 1/(v' U_N U_N' v); 10 log10(Q / max Q)."""
    if theta is None:
        theta = np.arange(P) * (180.0 / P) * np.pi / 180.0
    w, V = np.linalg.eigh(S_x)
    U_N = V[:, :N - M]
    U_N_sq = U_N @ U_N.conj().T
    Q = np.empty(P)
    for ii in range(P):
        v = _octave_amv(theta[ii], d, N)
        Q[ii] = 1.0 / np.real(v.conj() @ U_N_sq @ v)
    return 10 * np.log10(Q / Q.max())


def _octave_rmusic(S_x, d, N, M):
    """rMUSIC.m:24-59 for one snapshot: u(l+N) = sum(diag(U_N_sq, l)), l = -N+1..N-1; flipud; roots(u/u(1));
    dist = 1 - |root|; drop dist < 0; the M smallest dist; acos(angle(psi)/(2 pi d)) in degrees."""
    w, V = np.linalg.eigh(S_x)
    U_N = V[:, :N - M]
    U_N_sq = U_N @ U_N.conj().T
    u = np.array([np.trace(U_N_sq, offset=l) for l in range(-N + 1, N)])
    u = u[::-1]
    r = np.roots(u / u[0])
    dist = 1 - np.abs(r)
    keep = dist >= 0
    r, dist = r[keep], dist[keep]
    psi = []
    for _ in range(M):
        k = int(np.argmin(dist))
        psi.append(r[k])
        r, dist = np.delete(r, k), np.delete(dist, k)
    return np.rad2deg(np.arccos(np.angle(np.array(psi)) / (2 * np.pi * d)))


@pytest.mark.parametrize("N,d,thetas,K,ovl,snr", [(8, 0.5, [23.0], 256, 32, 10.0), (16, 0.5, [121.0], 256, 32, 10.0),
                                                   (4, 0.5, [52.0], 1024, 64, 20.0), (4, 0.5, [30.0, 123.0], 2048, 512, 15.0),
                                                   (6, 0.25, [60.0, 100.0], 512, 0, 10.0)])
def test_oracle_f64_equals_octave_golden_model(N, d, thetas, K, ovl, snr):
    M, P, n = len(thetas), 1024, 4
    x = sim.make_streams(N, (n - 1) * (K - ovl) + K, thetas, d, snr_db=snr, seed=N + K)
    R = oracle.autocorrelate(x, K, ovl, 1, n, precision="f64")
    spec = oracle.music_lin_array(R, d, M, N, P, "f64")
    ang = oracle.root_music(R, d, M, N, "f64")
    for i in range(n):
        S_x = R[i].reshape(N, N, order="F").astype(np.complex128)
        # the Hermitian matrix both sides decompose: the upper triangle of the item (what cheevd 'U' reads)
        S_x = np.triu(S_x, 1) + np.triu(S_x, 1).conj().T + np.diag(np.real(np.diag(S_x)))
        # (a) on the C++ block's angle grid (theta is a float member there, MUSIC_lin_array_impl.cc:64-72:
        #     the radians are rounded to float32): the same formulas in double must agree to rounding
        grid = oracle.music_theta_grid(P).astype(np.float64)
        want = _octave_music(S_x, float(np.float32(d)), N, M, P, theta=grid)
        assert np.abs(spec[i] - want).max() <= 1e-8, (i, float(np.abs(spec[i] - want).max()))
        # (b) on the .m file's own double grid: the float rounding of theta (6e-8 relative) moves Q at the
        #     null by 1e-6..1e-4 relative (sharper nulls move more), i.e. the dB spectrum by 1e-5..3e-3 dB (a uniform shift: the maximum sits on the null) -- the C++/Octave gap the reference's
        #     own QA absorbs in its tolerance
        want = _octave_music(S_x, float(np.float32(d)), N, M, P)
        assert np.abs(spec[i] - want).max() <= 1e-2, (i, float(np.abs(spec[i] - want).max()))
        assert int(np.argmax(spec[i])) == int(np.argmax(want))
        doa_deg = np.sort(_octave_rmusic(S_x, float(np.float32(d)), N, M))
        assert np.abs(np.sort(ang[i]) - doa_deg).max() <= 2e-4       # output is float32 degrees


def test_root_selection_padding_rule_of_the_reference():
    """lib/rootMUSIC_linear_array_impl.cc:131-141 restated once on explicit root sets (an oracle self-check; the device's
    selection stage is run against this rule on hand-made roots in tests/test_gpu_root_music.py): with fewer than
    num_targets interior roots the remaining picks hit an "inf" entry, arg(inf + 0i) = 0, i.e. 90 degrees; with none,
    index_min runs on an empty vector (an Armadillo error)."""
    roots = np.array([0.5 * np.exp(1j * 1.0), 2.0 * np.exp(1j * 1.0), 1.5, 1.25 * np.exp(-0.3j)])
    got = oracle.root_music_select(roots, 0.5, 3, "f64")
    want = np.sort(np.array([np.degrees(np.arccos(1.0 / np.pi)), 90.0, 90.0], np.float32))
    assert np.allclose(got, want, atol=1e-5)
    with pytest.raises(ValueError):
        oracle.root_music_select(np.array([1.5, 2.0 + 1j]), 0.5, 1, "f64")
    # a root exactly on the circle is not "inside" (dist > 0, :125), in either precision of dist
    for prec in ("f32", "f64"):
        got = oracle.root_music_select(np.array([1.0 + 0j, 0.5j, 3.0]), 0.5, 1, prec)
        assert np.allclose(got, [np.degrees(np.arccos(0.5))], atol=1e-4)
