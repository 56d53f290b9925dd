"""GPU tests of the pipelines on OVERLAPPING windows (reference: lib/autocorrelate_impl.cc:83-118 with overlap_size > 0, the
simulation flowgraphs' setting): K1 then streams only the new samples of every window step (cov_piece_kernel) and a combine
launch forms the windows, and this happens inside the fused handles on their own workspaces and lanes.  Bars: the covariance
a pipeline hands out is the autocorrelate BLOCK's, bit for bit, for window steps that divide the window or not, with and
without forward-backward averaging and fused antenna gains, also for shapes that do NOT take the read-once path; spectra,
peaks and Root-MUSIC angles equal the chained blocks'; a ragged last wave stores nothing beyond its batch.
(Written for the round-4 experiment that formed the windows inside the eigen stage -- tools/lab/cov_fold_variant.diff.txt,
measured and not kept -- and kept for the shapes it covers.)"""
import numpy as np
import pytest
import torch

import doa

pytestmark = pytest.mark.gpu

SHAPES = [   # N, K, overlap, fb, d, M
    (4, 2048, 512, 1, 0.4, 2),      # the simulation flowgraph
    (4, 2048, 512, 0, 0.4, 2),
    (3, 1024, 256, 1, 0.5, 2),
    (4, 1000, 250, 1, 0.45, 2),     # S = 750, r = 250: q = 1
    (4, 512, 384, 0, 0.5, 2),       # S = 128, q = 4, r = 0
    (4, 600, 400, 1, 0.5, 2),       # S = 200, q = 3, r = 0
    (4, 1024, 0, 1, 0.5, 2),        # no overlap: one-kernel K1
    (4, 2048, 512, 1, 0.4, 1),      # one source: the one-lane eigen stage
    (4, 2048, 512, 1, 0.4, 3),      # three sources: Jacobi
    (4, 1001, 250, 0, 0.5, 2),      # odd window step: no read-once path at all
    (6, 1024, 256, 1, 0.5, 2),      # wider array: group kernels
]


def _streams(N, span, M, d, seed):
    thetas = [35.0, 100.0, 150.0][:M]
    x = doa.sim.make_streams(N, span, thetas, d, snr_db=15.0, seed=seed)
    return x, [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in x]


@pytest.mark.parametrize("N,K,ovl,fb,d,M", SHAPES)
@pytest.mark.parametrize("gains", [False, True])
def test_music_pipeline_on_overlapping_windows_equals_the_blocks(N, K, ovl, fb, d, M, gains):
    n, P = 300, 1024
    x, dx = _streams(N, (n - 1) * (K - ovl) + K, M, d, seed=N * K + ovl)
    st = torch.cuda.current_stream()
    pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, max_batch=n)
    blk = doa.autocorrelate(N, K, ovl, fb)
    if gains:
        g = (np.random.default_rng(5).standard_normal(N) + 1j * np.random.default_rng(6).standard_normal(N)).astype(np.complex64)
        pipe.fuse_antenna_correction(g)
        blk.fuse_antenna_correction(g)
    cov = torch.empty((n, N * N), dtype=torch.complex64, device="cuda")
    spec = torch.empty((n, P), dtype=torch.float32, device="cuda")
    mx = torch.empty((n, M), dtype=torch.float32, device="cuda")
    am = torch.empty((n, M), dtype=torch.float32, device="cuda")
    assert pipe.work_dev(n, [t.data_ptr() for t in dx], cov.data_ptr(), spec.data_ptr(), mx.data_ptr(), am.data_ptr(), st) == n
    torch.cuda.synchronize()
    R = np.empty((n, N * N), np.complex64)
    blk.general_work(n, [x[k] for k in range(N)], [R])
    assert np.array_equal(cov.cpu().numpy().view(np.uint8), R.view(np.uint8))        # the block's items, bit for bit
    S = np.empty((n, P), np.float32)
    doa.MUSIC_lin_array(d, M, N, P).work(n, [R], [S])
    v0, v1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    doa.find_local_max(M, P, 0.0, 180.0).work(n, [S], [v0, v1])
    assert np.all(np.abs(spec.cpu().numpy() - S) <= 2e-6 + 5e-7 * np.abs(S))
    assert np.array_equal(am.cpu().numpy(), v1)
    # without a covariance pointer (the handle's own buffer) and without a spectrum pointer: same peaks
    mx2, am2 = torch.empty_like(mx), torch.empty_like(am)
    assert pipe.work_dev(n, [t.data_ptr() for t in dx], 0, 0, mx2.data_ptr(), am2.data_ptr(), st) == n
    torch.cuda.synchronize()
    assert np.array_equal(am2.cpu().numpy(), v1) and np.array_equal(mx2.cpu().numpy().view(np.uint8), mx.cpu().numpy().view(np.uint8))


@pytest.mark.parametrize("N,K,ovl,fb,d,M", [s for s in SHAPES if s[5] == 2])
def test_root_pipeline_on_overlapping_windows_equals_the_blocks(N, K, ovl, fb, d, M):
    n = 300
    x, dx = _streams(N, (n - 1) * (K - ovl) + K, M, d, seed=7 + K)
    st = torch.cuda.current_stream()
    rp = doa.root_pipeline(N, K, ovl, fb, d, M, n)
    cov = torch.empty((n, N * N), dtype=torch.complex64, device="cuda")
    ang = torch.empty((n, M), dtype=torch.float32, device="cuda")
    assert rp.work_dev(n, [t.data_ptr() for t in dx], cov.data_ptr(), ang.data_ptr(), None, st) == n
    torch.cuda.synchronize()
    R = np.empty((n, N * N), np.complex64)
    doa.autocorrelate(N, K, ovl, fb).general_work(n, [x[k] for k in range(N)], [R])
    assert np.array_equal(cov.cpu().numpy().view(np.uint8), R.view(np.uint8))
    A = np.empty((n, M), np.float32)
    doa.rootMUSIC_linear_array(d, M, N).work(n, [R], [A])
    assert np.array_equal(ang.cpu().numpy().view(np.uint8), A.view(np.uint8))


def test_overlapping_windows_over_lanes_and_ragged_tail():
    """batches entry (lanes with their own piece workspaces) and a batch size that leaves idle quads in the last wave"""
    N, K, ovl, fb, d, M, P, n, steps = 4, 2048, 512, 1, 0.4, 2, 1024, 203, 6
    st = torch.cuda.current_stream()
    ins = [_streams(N, (n - 1) * (K - ovl) + K, M, d, seed=40 + b) for b in range(2)]
    pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, max_batch=n)
    pipe.set_lanes(3)
    cov = [torch.full((n + 1, N * N), 7 + 7j, dtype=torch.complex64, device="cuda") for _ in range(steps)]     # one guard row each
    spec = [torch.empty((n, P), dtype=torch.float32, device="cuda") for _ in range(steps)]
    mx = [torch.empty((n, M), dtype=torch.float32, device="cuda") for _ in range(steps)]
    am = [torch.empty((n, M), dtype=torch.float32, device="cuda") for _ in range(steps)]
    pipe.work_dev_batches(n, [[t.data_ptr() for t in ins[b % 2][1]] for b in range(steps)], [t.data_ptr() for t in cov],
                          [t.data_ptr() for t in spec], [t.data_ptr() for t in mx], [t.data_ptr() for t in am], st)
    torch.cuda.synchronize()
    blk = doa.autocorrelate(N, K, ovl, fb)
    for b in range(steps):
        R = np.empty((n, N * N), np.complex64)
        blk.general_work(n, [ins[b % 2][0][k] for k in range(N)], [R])
        got = cov[b].cpu().numpy()
        assert np.array_equal(got[:n].view(np.uint8), R.view(np.uint8))
        assert np.all(got[n] == 7 + 7j)                                        # nothing is stored beyond the batch (idle quads / lanes of the last wave)
        assert np.array_equal(am[b].cpu().numpy(), am[b % 2].cpu().numpy())
