"""world_size-2 `gloo` test of the N>1 path on CPU: each rank takes its contiguous snapshot shard
(with the overlap halo), processes it independently — on CPU the oracle stands in for the device
kernels, the sharding/gather logic is the product's — and the gathered result equals the unsharded
one bit for bit.  Also checks the benchmark's max-over-ranks timing reduction."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, ovl, n_total, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "gr-doa_amd", "python"), os.path.join(root, "oracle"), os.path.join(root, "tests")]
    import doa_oracle as oracle
    from doa import sharding
    from test_cpu_oracle_pins import sim
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, S = 4, K - ovl
    x = sim.make_streams(N, (n_total - 1) * S + K, [40.0, 100.0], 0.45, snr_db=10.0, seed=3)   # same on every rank
    sh = sharding.shard_snapshots(n_total, world, rank, K, ovl)
    mine = oracle.autocorrelate(x[:, sh.sample_begin:sh.sample_end], K, ovl, 1, sh.n_snapshots)
    # gather of the small results only (variable shard sizes -> pad to the largest)
    biggest = max(s.n_snapshots for s in sharding.all_shards(n_total, world, K, ovl))
    buf = torch.zeros((biggest, N * N, 2), dtype=torch.float32)
    buf[: sh.n_snapshots] = torch.from_numpy(np.stack([mine.real, mine.imag], axis=-1))
    got = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(got, buf)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)       # pretend rank r took 1+r seconds
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        parts = []
        for r, g in enumerate(got):
            k = sharding.shard_snapshots(n_total, world, r, K, ovl).n_snapshots
            a = g[:k].numpy()
            parts.append((a[..., 0] + 1j * a[..., 1]).astype(np.complex64))
        full = oracle.autocorrelate(x, K, ovl, 1, n_total)
        np.save(os.path.join(out_dir, "ok.npy"),
                np.array([float(np.array_equal(np.concatenate(parts), full)), float(t.item()),
                          sharding.job_throughput([s.n_snapshots for s in sharding.all_shards(n_total, world, K, ovl)],
                                                  [1.0, 2.0])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,ovl,n_total", [(256, 64, 37), (128, 0, 10)])
def test_two_rank_sharding_equals_unsharded(tmp_path, K, ovl, n_total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, K, ovl, n_total, str(tmp_path)), nprocs=2, join=True)
    ok, tmax, rate = np.load(os.path.join(str(tmp_path), "ok.npy"))
    assert ok == 1.0
    assert tmax == 2.0                       # max over ranks
    assert rate == n_total / 2.0             # all units / slowest rank
