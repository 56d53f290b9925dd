"""GPU tests of doa.root_pipeline: autocorrelate -> rootMUSIC_linear_array as one handle (the chain
apps/run_RootMUSIC_lin_array_simulation.grc wires; reference lib/autocorrelate_impl.cc:83-118 ->
lib/rootMUSIC_linear_array_impl.cc:90-152).  Bars:
  * bit-identical to the two block handles chained by hand on the device (same kernels, same launch shapes), for overlap /
    forward-backward / wide-array shapes and the reference's simulation flowgraph scenario;
  * angles within 1e-3 deg of the fp64 oracle on noisy data (the bar of tests/test_gpu_root_music.py);
  * the batches entry (lanes, attached / detached / adopted streams) = single calls, bit for bit;
  * the host-buffer entry (staged small calls, chunked large ones) = the device entry; an item without an interior root
    makes it return DOA_ERR_NUMERIC with the other rows intact;
  * an error return of either entry leaves nothing running (fault injection)."""
import numpy as np
import pytest
import torch

import doa
import doa_oracle as oracle
from scenarios import make_input, is_rank_deficient

pytestmark = pytest.mark.gpu

CASES = ["grc_root_sim", "bench_cfg3", "qa_root_aoa52", "qa_root_aoa23", "three_ant_fb", "five_ant", "two_ant", "bench_cfg4"]


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _chained(c, d_streams, n):
    """the reference's two blocks, chained by hand on device buffers"""
    N, M = c["N"], c["M"]
    st = torch.cuda.current_stream()
    cov = torch.empty((n, N * N), dtype=torch.complex64, device="cuda")
    ang = torch.empty((n, M), dtype=torch.float32, device="cuda")
    doa.autocorrelate(N, c["K"], c["ovl"], c["fb"]).work_dev(n, [t.data_ptr() for t in d_streams], cov.data_ptr(), st)
    doa.rootMUSIC_linear_array(c["d"], M, N).work_dev(n, cov.data_ptr(), ang.data_ptr(), st)
    torch.cuda.synchronize()
    return cov, ang


@pytest.mark.parametrize("name", CASES)
def test_root_pipeline_equals_chained_blocks_and_oracle(name):
    c, x = make_input(name)
    N, M, n = c["N"], c["M"], c["n"]
    d_in = [_dev(x[k]) for k in range(N)]
    cov_ref, ang_ref = _chained(c, d_in, n)
    pipe = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    cov = torch.empty((n, N * N), dtype=torch.complex64, device="cuda")
    ang = torch.full((n, M), -1.0, dtype=torch.float32, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    assert pipe.work_dev(n, [t.data_ptr() for t in d_in], cov.data_ptr(), ang.data_ptr(), status.data_ptr(),
                         torch.cuda.current_stream()) == n
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(cov), torch.view_as_real(cov_ref))
    assert torch.equal(ang, ang_ref)
    assert int(status.abs().sum()) == 0
    # without the optional outputs: same angles
    ang2 = torch.empty_like(ang)
    assert pipe.work_dev(n, [t.data_ptr() for t in d_in], 0, ang2.data_ptr(), None, torch.cuda.current_stream()) == n
    torch.cuda.synchronize()
    assert torch.equal(ang2, ang_ref)
    # and against the oracle, as the stand-alone block is held
    got = ang.cpu().numpy()
    R = cov.cpu().numpy()
    a64 = oracle.root_music(R, c["d"], M, N, "f64")
    assert np.all(np.diff(got, axis=1) >= 0)
    assert np.all(np.abs(got - np.sort(np.asarray(c["thetas"], np.float32))[None, :]) <= 2.0)
    if not is_rank_deficient(c):
        assert np.abs(got - a64).max() <= 1e-3, (name, float(np.abs(got - a64).max()))


@pytest.mark.parametrize("name,lanes", [("bench_cfg3", 4), ("grc_root_sim", 3), ("five_ant", 2), ("bench_cfg3", 1)])
def test_root_batches_entry_equals_single_calls_bit_for_bit(name, lanes):
    c, x = make_input(name)
    N, M, n = c["N"], c["M"], c["n"]
    nb = 7
    S = c["K"] - c["ovl"]
    span = (n - 1) * S + c["K"]
    rng = np.random.default_rng(4)
    xs = [(x * np.float32(1.0 + 0.25 * b) + (0.02 * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)).astype(np.complex64)
          for b in range(nb)]
    d_in = [[_dev(xb[k][:span]) for k in range(N)] for xb in xs]
    mk = lambda shape, dt: [torch.full(shape, 0 if dt == torch.complex64 else -3, dtype=dt, device="cuda") for _ in range(nb)]
    ref = dict(cov=mk((n, N * N), torch.complex64), ang=mk((n, M), torch.float32))
    got = dict(cov=mk((n, N * N), torch.complex64), ang=mk((n, M), torch.float32), st=mk((n,), torch.int32))
    one = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    st = torch.cuda.current_stream()
    want_cov = [b % 2 == 0 for b in range(nb)]
    want_st = [b % 3 != 1 for b in range(nb)]
    for b in range(nb):
        one.work_dev(n, [t.data_ptr() for t in d_in[b]], ref["cov"][b].data_ptr() if want_cov[b] else 0, ref["ang"][b].data_ptr(), None, st)
    torch.cuda.synchronize()
    pipe = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    pipe.set_lanes(lanes)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        # the fork orders the lanes behind what the caller's stream holds; the join lets it read the last batch right after
        fresh = [t.clone() for t in d_in[0]]
        for t in d_in[0]:
            t.zero_()
        torch.cuda._sleep(2_000_000)
        for t, f in zip(d_in[0], fresh):
            t.copy_(f)
        produced = pipe.work_dev_batches(n, [[t.data_ptr() for t in d_in[b]] for b in range(nb)],
                                         [got["cov"][b].data_ptr() if want_cov[b] else 0 for b in range(nb)],
                                         [t.data_ptr() for t in got["ang"]],
                                         [got["st"][b].data_ptr() if want_st[b] else 0 for b in range(nb)], side)
        last = got["ang"][nb - 1].clone()
    assert produced == nb * n
    side.synchronize()
    assert torch.equal(last, ref["ang"][nb - 1])
    torch.cuda.synchronize()
    for b in range(nb):
        assert torch.equal(got["ang"][b], ref["ang"][b]), b
        if want_cov[b]:
            assert torch.equal(torch.view_as_real(got["cov"][b]), torch.view_as_real(ref["cov"][b])), b
        assert int(got["st"][b].abs().sum()) == (0 if want_st[b] else 3 * n), b        # untouched when not asked for
    # whole optional arrays omitted; detached form; adopted streams
    for mode in ("detached", "adopted"):
        p2 = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
        if mode == "adopted":
            keep = [torch.cuda.Stream() for _ in range(3)]
            p2.set_lane_streams(keep)
        ang2 = [torch.empty((n, M), dtype=torch.float32, device="cuda") for _ in range(nb)]
        for half in (range(0, 3), range(3, nb)):          # two calls: the lane rotation carries over
            assert p2.work_dev_batches(n, [[t.data_ptr() for t in d_in[b]] for b in half], None, [ang2[b].data_ptr() for b in half], None,
                                       doa.DETACHED) == len(half) * n
        p2.synchronize()
        for b in range(nb):
            assert torch.equal(ang2[b], ref["ang"][b]), (mode, b)


def test_root_batches_entry_argument_checks():
    pipe = doa.root_pipeline(4, 64, 0, 0, 0.5, 2, max_batch=8)
    z = torch.zeros(8 * 64, dtype=torch.complex64, device="cuda")
    out = torch.zeros((8, 2), dtype=torch.float32, device="cuda")
    ins = [[z.data_ptr()] * 4]
    with pytest.raises(doa.DoaError):
        pipe.work_dev_batches(9, ins, None, [out.data_ptr()], None, torch.cuda.current_stream())          # > max_batch
    with pytest.raises(doa.DoaError):
        pipe.work_dev_batches(8, ins, None, [0], None, torch.cuda.current_stream())                       # no angle pointer
    with pytest.raises(doa.DoaError):
        pipe.set_lanes(0)
    with pytest.raises(doa.DoaError):
        pipe.set_lanes(9)
    assert pipe.work_dev_batches(8, [], None, [], None, torch.cuda.current_stream()) == 0
    for args in [(0, 16, 0, 0, 0.5, 1), (4, 16, 16, 0, 0.5, 1), (4, 16, 0, 0, 0.5, 4), (4, 16, 0, 0, 0.7, 1), (4, 16, 0, 0, 0.5, 0),
                 (1, 16, 0, 0, 0.5, 1)]:
        with pytest.raises(doa.DoaError):
            doa.root_pipeline(*args)


@pytest.mark.parametrize("name", ["grc_root_sim", "bench_cfg3", "five_ant"])
def test_root_host_entry_equals_device_entry(name):
    c, x = make_input(name)
    N, M, n = c["N"], c["M"], c["n"]
    d_in = [_dev(x[k]) for k in range(N)]
    cov_ref, ang_ref = _chained(c, d_in, n)
    pipe = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    ang, cov = np.empty((n, M), np.float32), np.empty((n, N * N), np.complex64)
    assert pipe.work(n, [x[k] for k in range(N)], ang, cov_out=cov) == n
    assert np.array_equal(ang, ang_ref.cpu().numpy()) and np.array_equal(cov.view(np.float32), cov_ref.cpu().numpy().view(np.float32))
    ang2 = np.empty((n, M), np.float32)
    assert pipe.work(n, [x[k] for k in range(N)], ang2) == n                       # angles only
    assert np.array_equal(ang2, ang)
    # as a flowgraph block (history / forecast / general_work as doa.autocorrelate)
    assert pipe.history() == c["ovl"] + 1 and pipe.forecast(3) == 3 * (c["K"] - c["ovl"])
    ang3 = np.empty((n, M), np.float32)
    produced, consumed = pipe.general_work(n, [x[k] for k in range(N)], [ang3])
    assert produced == n and consumed == n * (c["K"] - c["ovl"]) and np.array_equal(ang3, ang)


def test_root_host_entry_multi_chunk_and_numeric_error():
    """A call above the staging limit is cut into ~32 MiB chunks alternating over two streams; the rows must be those of
    small calls.  One snapshot made non-finite: that item has no interior root -> DOA_ERR_NUMERIC (the reference raises in
    arma::index_min), every other row intact."""
    N, K, M, d = 4, 1024, 2, 0.44
    n = 2600                                               # 85 MB of samples: three chunks
    x = doa.sim.make_streams(N, n * K, [30.0, 123.0], d, snr_db=20.0, seed=11)
    pipe = doa.root_pipeline(N, K, 0, 0, d, M, max_batch=n)
    ang = np.empty((n, M), np.float32)
    assert pipe.work(n, [x[k] for k in range(N)], ang) == n
    small = doa.root_pipeline(N, K, 0, 0, d, M, max_batch=64)
    for i0 in (0, 1000, 1100, n - 64):
        a = np.empty((64, M), np.float32)
        assert small.work(64, [x[k][i0 * K:] for k in range(N)], a) == 64
        assert np.array_equal(a, ang[i0:i0 + 64]), i0
    assert np.abs(ang - np.array([[30.0, 123.0]], np.float32)).max() <= 1.0
    bad = 1500
    xb = [x[k].copy() for k in range(N)]
    xb[1][bad * K + 5] = np.nan
    ang_b = np.full((n, M), -1.0, np.float32)
    with pytest.raises(doa.DoaError) as ei:
        pipe.work(n, xb, ang_b)
    assert ei.value.status == -5 and f"item {bad}" in str(ei.value)
    keep = np.ones(n, bool)
    keep[bad] = False
    assert np.array_equal(ang_b[keep], ang[keep]) and np.all(np.isnan(ang_b[bad]))
    assert pipe.lanes_idle()
    # the staged path reports it too
    a = np.empty((8, M), np.float32)
    with pytest.raises(doa.DoaError) as ei:
        small.work(8, [v[(bad - 3) * K:] for v in xb], a)
    assert ei.value.status == -5 and "item 3" in str(ei.value)
    assert np.array_equal(a[:3], ang[bad - 3:bad]) and np.array_equal(a[4:], ang[bad + 1:bad + 5])


@pytest.mark.parametrize("detached", [True, False])
def test_root_batches_entry_failure_leaves_nothing_running(detached):
    N, K, M, d, n, nb = 4, 1024, 2, 0.44, 512, 6
    x = doa.sim.make_streams(N, nb * n * K, [40.0, 100.0], d, snr_db=20.0, seed=12)
    d_in = [[_dev(x[k][b * n * K:(b + 1) * n * K]) for k in range(N)] for b in range(nb)]
    pipe = doa.root_pipeline(N, K, 0, 0, d, M, max_batch=n)
    ang = [torch.full((n, M), -9.0, dtype=torch.float32, device="cuda") for _ in range(nb)]
    st = doa.DETACHED if detached else torch.cuda.current_stream()
    call = lambda: pipe.work_dev_batches(n, [[t.data_ptr() for t in d_in[b]] for b in range(nb)], None, [t.data_ptr() for t in ang], None, st)
    assert call() == nb * n
    pipe.synchronize()
    torch.cuda.synchronize()
    ref = [t.clone() for t in ang]
    for t in ang:
        t.fill_(-9.0)
    pipe.inject_failure(4)
    with pytest.raises(doa.DoaError) as ei:
        call()
    assert "injected failure in batch 4" in str(ei.value)
    assert pipe.lanes_idle()                                   # an error return means nothing of the call still runs
    for b in range(4):
        assert torch.equal(ang[b], ref[b]), b
    assert bool((ang[4] == -9.0).all()) and bool((ang[5] == -9.0).all())
    assert call() == nb * n                                    # one-shot: the next call works
    pipe.synchronize()
    torch.cuda.synchronize()
    for b in range(nb):
        assert torch.equal(ang[b], ref[b]), b
    pipe.inject_failure(nb + 3)                                # an index no batch of the next call reaches: disarmed by that call
    assert call() == nb * n
    pipe.synchronize()
    pipe.inject_failure(-1)
    assert call() == nb * n
    pipe.synchronize()


def test_root_pipeline_fused_antenna_correction():
    """R[a,b] *= g_a conj(g_b) in K1's epilogue = correcting the streams first (lib/antenna_correction_impl.cc:85-99)."""
    c, x = make_input("bench_cfg3")
    N, M, n = c["N"], c["M"], c["n"]
    g = (np.array([1.0, 0.8, 1.3, 0.9]) * np.exp(1j * np.array([0.0, 0.4, -1.1, 2.0]))).astype(np.complex64)
    xd = [(x[k] / g[k]).astype(np.complex64) for k in range(N)]          # what a mis-calibrated front end delivers
    plain = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    a_ref = np.empty((n, M), np.float32)
    plain.work(n, [x[k] for k in range(N)], a_ref)
    fused = doa.root_pipeline(N, c["K"], c["ovl"], c["fb"], c["d"], M, max_batch=n)
    fused.fuse_antenna_correction(g)                       # (an antenna_correction block or its gains)
    a = np.empty((n, M), np.float32)
    fused.work(n, xd, a)
    assert np.all(np.isfinite(a_ref)) and np.abs(a - a_ref).max() <= 2e-3
    fused.fuse_antenna_correction(None)                    # off again: the distorted streams' own angles
    b, b_ref = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
    fused.work(n, xd, b)
    plain.work(n, xd, b_ref)
    assert np.array_equal(b, b_ref, equal_nan=True)
