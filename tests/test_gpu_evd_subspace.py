"""GPU tests of the signal-subspace path of K2+K3 (csrc/evd_subspace.hpp): for 4 < N <= 16, M <= 4 (one wave per item),
for N <= 4, M = 1 (one lane per item) and -- round 4 -- for N = 4, M = 2 (four lanes per item, music_evd_quad_kernel) the noise projector is I - X X^H with X from a shifted orthogonal iteration, every item checked by its residual and by a certificate that X
spans the M LARGEST eigenvalues; items that fail a check take the full Jacobi EVD on the same wave.

What is pinned here, through the C ABI, against the fp64 oracle (numpy eigh on the same covariance items -- the
reference's eig_sym + U_N U_N^H, lib/MUSIC_lin_array_impl.cc:128-133):
  * the projector itself (doa_MUSIC_lin_array_debug) to 1e-7, the bound the Jacobi kernels are held to;
  * which path ran: doa_hip_evd_fallback_count;
  * adversarial inputs must take the fall-back and still match: equal eigenvalues across the signal/noise boundary
    (R = c I, block-diagonal with repeated values), the start columns orthogonal to a dominant eigenvector, zero matrices,
    non-finite items, signal and noise eigenvalues a factor 1.001 apart;
"""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu


def _items(mats):
    return np.stack([np.asarray(A, np.complex64).reshape(-1, order="F") for A in mats])


def _pn64(Rc64, N, M):
    out = []
    for it in Rc64:
        A = it.reshape(N, N, order="F").astype(np.complex128)
        A = np.triu(A) + np.triu(A, 1).conj().T                  # uplo = 'U'
        A[np.diag_indices(N)] = A.diagonal().real
        w, V = np.linalg.eigh(A)
        out.append(V[:, : N - M] @ V[:, : N - M].conj().T)
    return np.stack(out)


def _array_cov(rng, N, M, snr_db, K):
    th = np.sort(rng.uniform(15.0, 165.0, M)) + 6.0 * np.arange(M)
    A = np.exp(-2j * np.pi * 0.5 * np.cos(np.deg2rad(th))[None, :] * np.arange(N)[:, None])
    s = (rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))) / np.sqrt(2)
    w = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))) / np.sqrt(2) * 10 ** (-snr_db / 20)
    x = A @ s + w
    return x @ x.conj().T / K


@pytest.mark.parametrize("N,M", [(2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (4, 3), (5, 1), (5, 2), (6, 3), (8, 1), (8, 2), (8, 4), (9, 3), (12, 4), (16, 1),
                                 (16, 3), (16, 4)])
def test_projector_of_the_fast_path_matches_eigh(N, M):
    rng = np.random.default_rng(100 * N + M)
    mats = [_array_cov(rng, N, M, snr, K) for snr, K in ((20.0, 1024), (10.0, 512), (3.0, 256)) for _ in range(24)]
    R = _items(mats)
    blk = doa.MUSIC_lin_array(0.5, M, N, 256)
    doa.evd_fallback_count(reset=True)
    pn, _q = blk.debug(R)
    n_fb = doa.evd_fallback_count(reset=True)
    want = _pn64(R, N, M)
    got = pn.reshape(-1, N, N).transpose(0, 2, 1)                 # column-major items
    err = np.abs(got - want).max(axis=(1, 2))
    assert err.max() <= 1e-7, (N, M, err.max())
    # the fast path is the one that ran.  Measured fall-back counts on these 72 items (round 4): 0 for every pair except the
    # ones with 2M = N -- (12, 4) 5, (8, 4) 3 -- and (6, 3) / (9, 3) / (16, 3) / (16, 4) 1;
    # (4, 3) and (3, 2) have no fast path (one noise eigenvalue: 8 of 72 (4, 3) items fell back when it had, and took their waves
    # with them; at N = 3 the one-lane Jacobi beats the four-lane iteration 5.4 to 14.7 us per 4096 items).  Bounds = the measured
    # count + 2 (a regression that sends 15 % of the items down the slow path fails)
    print("fall-backs", (N, M), n_fb, "of", len(mats))
    # (4, 2): four lanes per item, and the Jacobi runs for a whole wave or not at all -- the 24 items at 3 dB SNR are more than
    # 11 steps from convergence at the first check and take their wave-mates with them (24 of 72 measured)
    assert n_fb <= {(12, 4): 7, (8, 4): 5, (3, 2): 0, (4, 3): 0, (4, 2): 26}.get((N, M), 2), (N, M, n_fb)
    if (N, M) == (4, 2):                              # ... while at 20 dB SNR the fast path is the one that runs
        doa.evd_fallback_count(reset=True)
        blk.debug(_items([_array_cov(rng, N, M, 20.0, 1024) for _ in range(64)]))
        assert doa.evd_fallback_count(reset=True) <= 2
    # production call (coefficient records only, a different kernel instantiation): spectra from both paths agree
    spec = np.empty((len(mats), 256), np.float32)
    assert blk.work(len(mats), [R], [spec]) == len(mats)
    s64 = oracle.music_lin_array(R, 0.5, M, N, 256, "f64")
    assert np.array_equal(np.argmax(spec, axis=1), np.argmax(s64, axis=1))
    assert np.abs(spec - s64).max() <= 1e-4


def _check_fallback(R, N, M, expect_all=True, tol=1e-7):
    blk = doa.MUSIC_lin_array(0.5, M, N, 64)
    doa.evd_fallback_count(reset=True)
    pn, _ = blk.debug(R)
    n_fb = doa.evd_fallback_count(reset=True)
    if expect_all:
        assert n_fb == R.shape[0], (n_fb, R.shape[0])
    return pn.reshape(-1, N, N).transpose(0, 2, 1), n_fb


@pytest.mark.parametrize("N", [4, 8, 16])
def test_equal_eigenvalues_across_the_boundary_take_the_jacobi_path(N):
    # R = c I and diag(5, 5, 5, 1, ...): with M = 1 or 2 the boundary falls between EQUAL eigenvalues; the certificate
    # cannot hold, the Jacobi kernel's rule (ranks by index among equals) decides, exactly as before this path existed
    mats = [3.0 * np.eye(N), np.diag([5.0, 5.0, 5.0] + [1.0] * (N - 3))]
    R = _items(mats)
    for M in (1, 2):
        pn, _ = _check_fallback(R, N, M)
        # ranks by index among equal eigenvalues: noise set = the N-M eigenvalues of lowest (value, index)
        for k, A in enumerate(mats):
            lam = np.real(np.diag(A))
            order = sorted(range(N), key=lambda i: (lam[i], i))
            want = np.zeros((N, N)); want[order[: N - M], order[: N - M]] = 1.0
            assert np.abs(pn[k] - want).max() <= 1e-12, (N, M, k)


def test_start_vectors_orthogonal_to_a_dominant_direction():
    # the iteration starts from the first M columns of A; a dominant eigenvector with zeros in its first M entries is
    # invisible to that start in exact arithmetic.  Whatever happens (fall-back, or rounding finds it), the result is eigh's
    N, M = 8, 2
    v1 = np.zeros(N, complex); v1[4:] = np.exp(1j * np.arange(4)) / 2.0          # first M entries exactly zero
    v2 = np.ones(N, complex) / np.sqrt(N)
    v2 -= v1 * (v1.conj() @ v2); v2 /= np.linalg.norm(v2)
    A = 9.0 * np.outer(v1, v1.conj()) + 4.0 * np.outer(v2, v2.conj()) + 0.01 * np.eye(N)
    R = _items([A, A * 3.0])
    pn, n_fb = _check_fallback(R, N, M, expect_all=False)
    assert np.abs(pn - _pn64(R, N, M)).max() <= 1e-7


@pytest.mark.parametrize("N,M", [(8, 2), (4, 2), (3, 2), (4, 3)])
def test_degenerate_items_take_the_fallback_and_keep_its_semantics(N, M):
    rng = np.random.default_rng(1)
    good = _array_cov(rng, N, M, 20.0, 256)
    zero = np.zeros((N, N))
    bad = good.copy(); bad[1, N - 1] = np.nan
    inf = good.copy(); inf[0, 0] = np.inf
    R = _items([good, zero, bad, inf, good])
    blk = doa.MUSIC_lin_array(0.5, M, N, 64)
    doa.evd_fallback_count(reset=True)
    pn, q = blk.debug(R)
    # (one noise eigenvalue on N <= 4 has no fast path since round 4's measurements: the Jacobi runs directly and nothing is counted)
    # (N = 4, M = 2: four lanes per item; the two good items share the wave of the three bad ones and take the Jacobi with them)
    assert doa.evd_fallback_count(reset=True) == {(4, 3): 0, (3, 2): 0, (4, 2): 5}.get((N, M), 3)
    pn = pn.reshape(-1, N, N).transpose(0, 2, 1)
    assert np.abs(pn[[0, 4]] - _pn64(R[[0, 4]], N, M)).max() <= 1e-7
    assert not np.isfinite(pn[2]).all() and not np.isfinite(pn[3]).all()       # non-finite in, non-finite out (never a plausible record)
    # zero matrix: all eigenvalues equal -> ranks by index: the first N-M unit vectors
    want = np.zeros((N, N)); want[np.arange(N - M), np.arange(N - M)] = 1.0
    assert np.abs(pn[1] - want).max() <= 1e-12


def test_nearly_equal_signal_and_noise_eigenvalues():
    # lambda_M / lambda_{M+1} = 1.001: the iteration would need thousands of steps; it gives up after 20, the Jacobi kernel
    # resolves the split (well defined in double: relative gap 1e-3)
    N, M = 16, 2
    rng = np.random.default_rng(9)
    Qm, _ = np.linalg.qr(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)))
    lam = np.concatenate([[2.0, 1.001], np.full(N - 2, 1.0) - 1e-4 * np.arange(N - 2)])
    A = (Qm * lam) @ Qm.conj().T
    R = _items([A])
    pn, n_fb = _check_fallback(R, N, M, expect_all=True)
    # float input rounding (6e-8) over a gap of 1e-3: eigh of the SAME float item is the reference point
    assert np.abs(pn - _pn64(R, N, M)).max() <= 1e-6


@pytest.mark.parametrize("N,M,P", [(8, 2, 1024), (16, 3, 4096)])
def test_pipeline_and_root_music_through_the_fast_path_full_batch(N, M, P):
    """size-independent properties at a full batch: every row's maximum is exactly 0 dB, estimated directions sit on the
    generated ones, Root-MUSIC agrees with the spectral peaks, and (almost) no item needed the fall-back"""
    torch = pytest.importorskip("torch")
    n, K = 4096, 256
    s, th = doa.sim.make_batch_streams_torch(N, K, n, 0.5, M, 20.0, seed=31, device="cuda")
    pipe = doa.music_pipeline(N, K, 0, 0, 0.5, M, P, n)
    cov = torch.empty((n, N * N), dtype=torch.complex64, device="cuda")
    spec = torch.empty((n, P), dtype=torch.float32, device="cuda")
    mx = torch.empty((n, M), dtype=torch.float32, device="cuda")
    am = torch.empty((n, M), dtype=torch.float32, device="cuda")
    doa.evd_fallback_count(reset=True)
    pipe.work_dev(n, [t.data_ptr() for t in s], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(), am.data_ptr(), torch.cuda.current_stream())
    torch.cuda.synchronize()
    n_fb = doa.evd_fallback_count(reset=True)
    assert n_fb <= n // 50, n_fb
    assert bool((spec.max(dim=1).values == 0).all())
    est = np.sort(am.cpu().numpy(), axis=1)
    assert np.abs(est - np.sort(th, axis=1)).max() <= 3.0           # MUSIC's own bias for sources 4 degrees apart at N = 8
    k = 96                                                          # and bin for bin against the fp64 oracle on a sample
    Rk = cov[:k].cpu().numpy()
    s64 = oracle.music_lin_array(Rk, 0.5, M, N, P, "f64")
    _, loc64 = oracle.find_local_max(s64.astype(np.float32), M, P, 0.0, 180.0)
    assert np.abs(np.sort(loc64, axis=1) - est[:k]).max() <= 180.0 / P + 1e-3
    root = doa.rootMUSIC_linear_array(0.5, M, N)
    ang = np.empty((n, M), np.float32)
    assert root.work(n, [cov.cpu().numpy()], [ang]) == n
    assert np.abs(ang - np.sort(th, axis=1)).max() <= 3.0
    assert np.abs(ang[:k] - oracle.root_music(Rk, 0.5, M, N, "f64")).max() <= 1e-3


def test_fallback_counter_lives_on_the_device_that_launches():
    """the fall-back counter a K2+K3 launch adds into must belong to the device the launch runs on (one
    counter per device since round 4).  On a one-GPU box: the counter the current device would use is allocated on that
    device; with two or more: a handle created on device 1 runs an item that must fall back (R = c I) and the count arrives."""
    import ctypes as C
    from doa import _lib
    torch = pytest.importorskip("torch")
    assert _lib.lib.doa_hip_evd_fallback_counter_device_debug() == torch.cuda.current_device()
    if doa.device_count() >= 2:
        with torch.cuda.device(1):
            assert _lib.lib.doa_hip_evd_fallback_counter_device_debug() == 1
            blk = doa.MUSIC_lin_array(0.5, 2, 8, 64)
            doa.evd_fallback_count(reset=True)
            blk.debug(_items([2.0 * np.eye(8)]))
            assert doa.evd_fallback_count(reset=True) == 1


def test_quad_path_on_forward_backward_flowgraph_items():
    """The simulation flowgraph's own items (N = 4, two sources, forward-backward averaging: the shape on which the one-lane
    iteration took 30 us against the Jacobi's 10): projector against eigh, fall-backs counted, and the coefficient records --
    u_l for Root-MUSIC, the lean scan's (A, B) record through the spectrum -- against the fp64 oracle."""
    from scenarios import make_input
    c, x = make_input("grc_music_sim")
    N, M, P, n = c["N"], c["M"], c["P"], c["n"]
    R = oracle.autocorrelate(x, c["K"], c["ovl"], c["fb"], n)
    blk = doa.MUSIC_lin_array(c["d"], M, N, P)
    doa.evd_fallback_count(reset=True)
    pn, _ = blk.debug(R)
    n_fb = doa.evd_fallback_count(reset=True)
    got = pn.reshape(-1, N, N).transpose(0, 2, 1)
    assert np.abs(got - _pn64(R, N, M)).max() <= 1e-7
    assert n_fb <= n // 2
    spec = np.empty((n, P), np.float32)
    blk.work(n, [R], [spec])
    s64 = oracle.music_lin_array(R, c["d"], M, N, P, "f64")
    assert np.abs(spec - s64).max() <= 2e-3 and np.array_equal(np.argmax(spec, axis=1), np.argmax(s64, axis=1))
