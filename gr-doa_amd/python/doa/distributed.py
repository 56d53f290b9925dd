"""The sharded run: one process per GPU, contiguous snapshot shards, no data-path collective.

Every snapshot window of autocorrelate -> MUSIC / Root-MUSIC is independent (the reference has no
reduction across snapshots), so an N-rank job is

    shard the stream     each rank takes a contiguous range of snapshot indices and the samples
                         behind it, including the `overlap_size` samples in front of its first new
                         sample: the halo GNU Radio's set_history(overlap+1) hands the block
                         (reference lib/autocorrelate_impl.cc:56-57; `doa.sharding`);
    per-rank pipeline    the rank's own device pipeline over its shard (injected as `compute`);
    (shard scatter       optional, `scatter_shards`: when ONE rank ingests the whole stream it sends every other rank its
                         shard, halo included, point to point;)
    result gather        optional: one all_gather of the per-snapshot results (angles: a few bytes
                         per snapshot; RCCL over xGMI on GPUs, gloo on CPU).  Shards differ by at
                         most one snapshot, so the gather pads to the largest shard.

`run_sharded` is that sequence; bench.py (N > 1) and the world-size-2 gloo test both call it.  The
module never imports the oracle and holds no arithmetic of the path.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np

from . import sharding


def _dist():
    import torch.distributed as dist
    return dist


def world_info():
    """(rank, world_size, local_rank) from the torch.distributed launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: Optional[str] = None, device=None):
    """Joins the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  backend defaults
    to "nccl" (= RCCL on ROCm) when a device is given, else "gloo".  Returns torch.distributed."""
    dist = _dist()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dist.is_initialized():
        return dist
    if backend is None:
        backend = "nccl" if device is not None else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist


def gather_results(local, shards: Sequence[sharding.Shard], dist=None):
    """all_gather of per-snapshot result rows.  `local`: torch tensor [n_local, ...] on the device
    the process group communicates on (CUDA for nccl, CPU for gloo).  Returns the concatenation
    over ranks in snapshot order, [n_total, ...], on every rank."""
    import torch
    dist = dist or _dist()
    world = len(shards)
    biggest = max(s.n_snapshots for s in shards)
    tail = tuple(local.shape[1:])
    live = dist.is_available() and dist.is_initialized()
    dev = torch.device("cpu") if (live and dist.get_backend() == "gloo") else local.device   # gloo communicates host memory
    buf = torch.zeros((biggest,) + tail, dtype=local.dtype, device=dev)
    buf[: local.shape[0]] = local
    if world == 1 and not live:
        return buf[: shards[0].n_snapshots].clone()
    out = torch.empty((world * biggest,) + tail, dtype=local.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf.contiguous())
    parts = [out[r * biggest: r * biggest + shards[r].n_snapshots] for r in range(world)]
    return torch.cat(parts, dim=0)


def scatter_shards(streams, n_streams: int, n_snapshots: int, snapshot_size: int, overlap_size: int, *,
                   src: int = 0, device=None, dist=None):
    """The ingest rank hands every rank its shard: point-to-point sends of [shard samples incl. the overlap halo] per
    stream from rank `src` (which holds the N whole streams as complex64 torch tensors; the other ranks pass None), all
    posted at once (`batch_isend_irecv`: on an 8-GPU node the 7 shards leave over 7 xGMI links in parallel).  This is
    the only exchange the path has on its input side, and a job whose ranks ingest their own shards never needs it.
    Returns this rank's list of N complex64 tensors on `device` (CPU for gloo)."""
    import torch
    dist = dist or _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = sharding.all_shards(n_snapshots, world, snapshot_size, overlap_size)
    gloo = dist.get_backend() == "gloo"
    dev = torch.device("cpu") if gloo else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    me = shards[rank]
    # every rank enters one collective first: on RCCL a FIRST point-to-point batch in which not all ranks take part (a rank
    # whose shard is empty posts nothing below) is undefined behaviour; after a collective the communicator is up on all of them
    sync = torch.zeros(1, dtype=torch.float32, device=dev)
    dist.all_reduce(sync)
    if rank == src:
        if streams is None or len(streams) != n_streams:
            raise ValueError("the ingest rank must hold all n_streams streams")
        ops, keep = [], []
        for r, sh in enumerate(shards):
            if r == src or sh.n_samples == 0:
                continue
            for k in range(n_streams):
                piece = torch.view_as_real(streams[k][sh.sample_begin:sh.sample_end].to(dev).contiguous())
                keep.append(piece)
                ops.append(dist.P2POp(dist.isend, piece, r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return [streams[k][me.sample_begin:me.sample_end].to(dev).contiguous() for k in range(n_streams)]
    mine = [torch.empty((me.n_samples, 2), dtype=torch.float32, device=dev) for _ in range(n_streams)]
    if me.n_samples:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, src) for t in mine]):
            w.wait()
    return [torch.view_as_complex(t) for t in mine]


def run_sharded(streams, n_snapshots: int, snapshot_size: int, overlap_size: int,
                compute: Callable, *, rank: Optional[int] = None, world_size: Optional[int] = None,
                gather: bool = True, dist=None):
    """Shard -> per-rank pipeline -> (optional) result gather.

    streams   either an indexable [N][total_samples] array/tensor list every rank can see (tests), or a
              callable `streams(sample_begin, sample_end)` returning this rank's slice of every stream
              (a rank that ingests or generates only its own shard: the production shape).
              Sample 0 is the first history sample of the job, exactly as for the block's work().
    compute   `compute(shard_streams, n_local) -> torch tensor [n_local, ...]`: the rank's device
              pipeline (e.g. doa.music_pipeline.work_dev wrapped by the caller).
    Returns (results, shard): the gathered [n_snapshots, ...] tensor (or the local one when
    gather=False) and this rank's Shard."""
    if rank is None or world_size is None:
        d = dist or _dist()
        if d.is_available() and d.is_initialized():
            rank, world_size = d.get_rank(), d.get_world_size()
        else:
            rank, world_size = 0, 1
    shards = sharding.all_shards(n_snapshots, world_size, snapshot_size, overlap_size)
    sh = shards[rank]
    if callable(streams):
        mine = streams(sh.sample_begin, sh.sample_end)
    else:
        mine = [s[sh.sample_begin:sh.sample_end] for s in streams]
    local = compute(mine, sh.n_snapshots)
    if local.shape[0] != sh.n_snapshots:
        raise ValueError(f"compute returned {local.shape[0]} rows for a shard of {sh.n_snapshots} snapshots")
    if not gather:
        return local, sh
    return gather_results(local, shards, dist), sh


def max_over_ranks(seconds: float, device=None, dist=None) -> float:
    """The benchmark's timing reduction: the slowest rank's time, on every rank."""
    import torch
    dist = dist or _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


from .launch import free_port, rank_commands, launch_ranks  # noqa: E402,F401  (one child process per GPU)
