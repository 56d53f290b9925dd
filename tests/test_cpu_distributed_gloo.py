"""world_size-2 `gloo` tests of the N>1 path on CPU.

The product's sharded run (`doa.distributed.run_sharded`: contiguous snapshot shards with the
overlap halo -> per-rank pipeline -> all_gather of the results) is driven exactly as bench.py
drives it on GPUs; only the per-rank `compute` is injected — on CPU the oracle stands in for the
device kernels.  The gathered result must equal the unsharded one bit for bit.  Also covered: the
benchmark's max-over-ranks timing reduction, the launcher bench.py --gpus N uses to start its
ranks, and the dry run of that command line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, K, ovl, n_total, out_dir, by_callable):
    sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import doa_oracle as oracle
    from doa import distributed, sharding
    from test_cpu_oracle_pins import sim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    distributed.init_process_group("gloo")
    N, S = 4, K - ovl
    x = sim.make_streams(N, (n_total - 1) * S + K, [40.0, 100.0], 0.45, snr_db=10.0, seed=3)   # same on every rank
    seen = {}

    def compute(shard_streams, n_local):            # the rank's pipeline: here the oracle's autocorrelate
        xs = np.stack([np.asarray(s) for s in shard_streams])
        seen["samples"] = xs.shape[1]
        R = oracle.autocorrelate(xs, K, ovl, 1, n_local)
        return torch.from_numpy(np.stack([R.real, R.imag], axis=-1).astype(np.float32))

    streams = (lambda b, e: [x[k, b:e] for k in range(N)]) if by_callable else x
    got, sh = distributed.run_sharded(streams, n_total, K, ovl, compute)
    assert sh == sharding.shard_snapshots(n_total, world, rank, K, ovl)
    # the halo rule (reference lib/autocorrelate_impl.cc:56-57): a shard holds its windows' new samples plus
    # `overlap` samples of history in front of the first one
    assert seen["samples"] == (sh.n_snapshots - 1) * S + K == sh.n_snapshots * S + ovl
    tmax = distributed.max_over_ranks(1.0 + rank)   # pretend rank r took 1+r seconds
    if rank == 0:
        a = got.numpy()
        full = oracle.autocorrelate(x, K, ovl, 1, n_total)
        same = np.array_equal((a[..., 0] + 1j * a[..., 1]).astype(np.complex64), full)
        rate = sharding.job_throughput([s.n_snapshots for s in sharding.all_shards(n_total, world, K, ovl)], [1.0, 2.0])
        np.save(os.path.join(out_dir, "ok.npy"), np.array([float(same), tmax, rate, float(a.shape[0])]))
    dist.barrier()
    dist.destroy_process_group()


def _scatter_worker(rank, world, port, K, ovl, n_total, out_dir):
    sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import doa_oracle as oracle
    from doa import distributed, sharding
    from test_cpu_oracle_pins import sim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    distributed.init_process_group("gloo")
    N, S = 3, K - ovl
    x = None
    if rank == 0:                                   # only the ingest rank ever holds the stream
        x = sim.make_streams(N, (n_total - 1) * S + K, [55.0], 0.5, snr_db=10.0, seed=9)
    streams = [torch.from_numpy(np.ascontiguousarray(x[k])) for k in range(N)] if rank == 0 else None
    mine = distributed.scatter_shards(streams, N, n_total, K, ovl, src=0)
    sh = sharding.shard_snapshots(n_total, world, rank, K, ovl)
    assert len(mine) == N and all(t.shape[0] == sh.n_samples and t.dtype == torch.complex64 for t in mine)

    def compute(shard_streams, n_local):
        xs = np.stack([s.numpy() for s in shard_streams]) if n_local else np.zeros((N, 0), np.complex64)
        R = oracle.autocorrelate(xs, K, ovl, 0, n_local) if n_local else np.zeros((0, N * N), np.complex64)
        return torch.from_numpy(np.stack([R.real, R.imag], axis=-1).astype(np.float32))

    # the scattered shard is handed over as "this rank's slice": a callable that ignores the (global) range it is asked for
    got, _ = distributed.run_sharded(lambda b, e: mine, n_total, K, ovl, compute)
    if rank == 0:
        a = got.numpy()
        full = oracle.autocorrelate(x, K, ovl, 0, n_total)
        np.save(os.path.join(out_dir, "scatter_ok.npy"),
                np.array([float(np.array_equal((a[..., 0] + 1j * a[..., 1]).astype(np.complex64), full)), float(a.shape[0])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K,ovl,n_total", [(2, 128, 32, 21), (3, 64, 0, 4), (3, 32, 31, 2)])
def test_ingest_rank_scatters_shards_with_halo(tmp_path, world, K, ovl, n_total):
    """one rank ingests, point-to-point scatter of the shards (halo included), per-rank pipeline, gather."""
    from doa import launch
    mp.spawn(_scatter_worker, args=(world, launch.free_port(), K, ovl, n_total, str(tmp_path)), nprocs=world, join=True)
    ok, rows = np.load(os.path.join(str(tmp_path), "scatter_ok.npy"))
    assert ok == 1.0 and rows == n_total


@pytest.mark.parametrize("K,ovl,n_total,by_callable", [(256, 64, 37, False), (128, 0, 10, True), (64, 48, 5, True)])
def test_two_rank_run_sharded_equals_unsharded(tmp_path, K, ovl, n_total, by_callable):
    from doa import launch
    mp.spawn(_worker, args=(2, launch.free_port(), K, ovl, n_total, str(tmp_path), by_callable), nprocs=2, join=True)
    ok, tmax, rate, rows = np.load(os.path.join(str(tmp_path), "ok.npy"))
    assert ok == 1.0 and rows == n_total
    assert tmax == 2.0                       # max over ranks
    assert rate == n_total / 2.0             # all units / slowest rank


def test_run_sharded_single_process_needs_no_process_group():
    from doa import distributed
    x = np.arange(40, dtype=np.float32)[None, :]
    got, sh = distributed.run_sharded(x, 7, 10, 5, lambda s, n: torch.arange(n, dtype=torch.float32)[:, None])
    assert sh.n_snapshots == 7 and sh.sample_begin == 0 and sh.sample_end == 40
    assert got[:, 0].tolist() == list(range(7))


def test_bench_dry_run_prints_one_command_per_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 2
    for rank, l in enumerate(lines):
        assert f" RANK={rank} " in " " + l and "WORLD_SIZE=2" in l and "MASTER_ADDR=127.0.0.1" in l
        assert f"LOCAL_RANK={rank}" in l and "bench.py --gpus 2 --steps 3" in l and "--dry-run" not in l
    # the parent must be able to do this without loading torch or the HIP library
    probe = ("import sys, runpy; sys.argv=['bench.py','--gpus','2','--dry-run']; runpy.run_path(%r, run_name='__main__'); "
             "assert 'torch' not in sys.modules and 'doa' not in sys.modules" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_launcher_starts_ranks_relays_rank0_and_reports_failure(tmp_path):
    from doa import launch
    script = tmp_path / "child.py"
    script.write_text("import os, sys, json\n"
                      "r = int(os.environ['RANK'])\n"
                      "open(os.path.join(sys.argv[1], f'rank{r}.json'), 'w').write(json.dumps({k: os.environ[k] for k in "
                      "('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT')}))\n"
                      "print('line from rank', r)\n"
                      "sys.exit(int(sys.argv[2]) if r == 1 else 0)\n")
    probe = ("import sys; sys.path.insert(0, %r); import launch; sys.exit(launch.launch_ranks(%r, [%r, sys.argv[1]], 2, timeout=60))"
             % (os.path.join(ROOT, "gr-doa_amd", "python", "doa"), str(script), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", probe, "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "line from rank 0"          # only rank 0's stdout is relayed
    envs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [e["RANK"] for e in envs] == ["0", "1"] and envs[0]["MASTER_PORT"] == envs[1]["MASTER_PORT"]
    assert all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    r = subprocess.run([sys.executable, "-c", probe, "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3                                                       # a failing rank fails the job
    assert launch.rank_commands("b.py", ["--x"], 3, port=5)[2][1]["LOCAL_RANK"] == "2"
