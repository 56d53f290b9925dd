"""GPU tests of the product-level sharded run (doa.distributed.run_sharded): the ranks of an N-GPU job are played one
after the other on the one device of the test box (rank / world_size passed explicitly, no process group), and the
concatenation of their device results must equal the unsharded run BIT FOR BIT -- the path has no data-path collective,
so a shard plus its overlap halo is all a rank ever needs (reference: lib/autocorrelate_impl.cc:56-57, set_history).
The multi-process side of the same function (all_gather over gloo, launcher) is in test_cpu_distributed_gloo.py."""
import numpy as np
import pytest

import doa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, D = 4, 0.45


def _streams(n_snap, K, ovl, seed):
    src = doa.sim_source(N, D, [41.0, 117.0], [0.031, 0.047], None, None, 0.1, seed=seed)
    total = ovl + n_snap * (K - ovl)
    bufs = [torch.empty(total, dtype=torch.complex64, device="cuda") for _ in range(N)]
    src.work_dev(total, [b.data_ptr() for b in bufs], torch.cuda.current_stream())
    torch.cuda.synchronize()
    return bufs


def _music(K, ovl, fb, M, P):
    def compute(bufs, n_local):
        bufs = [b.clone() for b in bufs]                 # a rank holds its shard in its own (aligned) allocation
        out = torch.empty((n_local, 2 * M), dtype=torch.float32, device="cuda")
        if n_local == 0:
            return out
        pipe = doa.music_pipeline(N, K, ovl, fb, D, M, P, n_local)
        mx = torch.empty((n_local, M), dtype=torch.float32, device="cuda")
        am = torch.empty((n_local, M), dtype=torch.float32, device="cuda")
        assert pipe.work_dev(n_local, [b.data_ptr() for b in bufs], 0, 0, mx.data_ptr(), am.data_ptr(),
                             torch.cuda.current_stream()) == n_local
        torch.cuda.synchronize()
        out[:, :M], out[:, M:] = mx, am
        return out
    return compute


def _root(K, ovl, fb, M):
    def compute(bufs, n_local):
        bufs = [b.clone() for b in bufs]                 # a rank holds its shard in its own (aligned) allocation
        ang = torch.empty((n_local, M), dtype=torch.float32, device="cuda")
        if n_local == 0:
            return ang
        cov = torch.empty((n_local, N * N), dtype=torch.complex64, device="cuda")
        st = torch.cuda.current_stream()
        assert doa.autocorrelate(N, K, ovl, fb).work_dev(n_local, [b.data_ptr() for b in bufs], cov.data_ptr(), st) == n_local
        assert doa.rootMUSIC_linear_array(D, M, N).work_dev(n_local, cov.data_ptr(), ang.data_ptr(), st) == n_local
        torch.cuda.synchronize()
        return ang
    return compute


@pytest.mark.parametrize("K,ovl,fb,n_snap,world", [(1024, 0, 0, 600, 8), (2048, 512, 1, 203, 3), (64, 63, 0, 131, 4), (256, 32, 1, 5, 8)])
@pytest.mark.parametrize("path", ["music", "root"])
def test_serially_played_ranks_equal_the_unsharded_run(K, ovl, fb, n_snap, world, path):
    bufs = _streams(n_snap, K, ovl, seed=K + world)
    compute = _music(K, ovl, fb, 2, 1024) if path == "music" else _root(K, ovl, fb, 2)
    whole, sh = doa.distributed.run_sharded(bufs, n_snap, K, ovl, compute, rank=0, world_size=1, gather=False)
    assert sh.n_snapshots == n_snap and whole.shape[0] == n_snap
    parts, covered = [], 0
    for r in range(world):
        loc, s = doa.distributed.run_sharded(bufs, n_snap, K, ovl, compute, rank=r, world_size=world, gather=False)
        assert s.first_snapshot == covered and loc.shape[0] == s.n_snapshots
        # a rank is handed its own samples plus the overlap halo and nothing else
        assert s.n_samples == (ovl + s.n_snapshots * (K - ovl) if s.n_snapshots else 0)
        covered += s.n_snapshots
        parts.append(loc)
    assert covered == n_snap
    got = torch.cat(parts, dim=0)
    assert torch.equal(got, whole)                       # bit for bit, NaN-free by construction
    # and the job does what it is for: both sources found in every snapshot (each block's own output order)
    ang = torch.sort(whole[:, 2:] if path == "music" else whole, dim=1).values
    tol = 3.0 if K >= 256 else 12.0                      # 64-sample windows: the estimator's own spread
    assert float((ang[:, 0] - 41.0).abs().max()) <= tol and float((ang[:, 1] - 117.0).abs().max()) <= tol


def test_callable_stream_source_generates_only_the_shard():
    """The production shape: a rank never sees the whole stream, it generates (or ingests) [begin, end) only."""
    K, ovl, n_snap, world = 512, 128, 77, 3
    bufs = _streams(n_snap, K, ovl, seed=11)
    compute = _music(K, ovl, 0, 2, 512)
    whole, _ = doa.distributed.run_sharded(bufs, n_snap, K, ovl, compute, rank=0, world_size=1, gather=False)
    asked = []

    def source(begin, end):
        asked.append((begin, end))
        src = doa.sim_source(N, D, [41.0, 117.0], [0.031, 0.047], None, None, 0.1, seed=11)
        src.seek(begin)
        out = [torch.empty(end - begin, dtype=torch.complex64, device="cuda") for _ in range(N)]
        src.work_dev(end - begin, [b.data_ptr() for b in out], torch.cuda.current_stream())
        torch.cuda.synchronize()
        return out

    parts = [doa.distributed.run_sharded(source, n_snap, K, ovl, compute, rank=r, world_size=world, gather=False)[0]
             for r in range(world)]
    assert torch.equal(torch.cat(parts, dim=0), whole)
    step = K - ovl
    assert asked == [(0, 26 * step + ovl), (26 * step, 52 * step + ovl), (52 * step, 77 * step + ovl)]
