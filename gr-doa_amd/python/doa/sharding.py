"""Snapshot sharding across ranks (one process per GPU).

Every snapshot window is independent (the reference has no cross-snapshot reduction), so a stream
is split into contiguous ranges of snapshot indices, one per rank, with no data-path collective.
With overlapping windows (overlap_size > 0) a shard needs the `overlap_size` samples in front of
its first new sample — the same halo GNU Radio's set_history(overlap+1) provides
(reference lib/autocorrelate_impl.cc:57) — so a shard's sample range is
    [first_snapshot * S, (last_snapshot) * S + K),   S = K - overlap.
Only the small results (angles / spectra) are ever gathered, and only if the caller wants them in
one place.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    first_snapshot: int      # global index of this rank's first window
    n_snapshots: int
    sample_begin: int        # first sample (history included) this rank must hold, per stream
    sample_end: int          # one past the last sample

    @property
    def n_samples(self) -> int:
        return self.sample_end - self.sample_begin


def shard_snapshots(n_snapshots: int, world_size: int, rank: int, snapshot_size: int, overlap_size: int) -> Shard:
    """Contiguous, balanced split of n_snapshots windows over world_size ranks."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if not (0 <= overlap_size < snapshot_size):
        raise ValueError("need 0 <= overlap_size < snapshot_size")
    base, rem = divmod(n_snapshots, world_size)
    first = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    S = snapshot_size - overlap_size
    begin = first * S
    end = begin if count == 0 else (first + count - 1) * S + snapshot_size
    return Shard(rank, first, count, begin, end)


def all_shards(n_snapshots: int, world_size: int, snapshot_size: int, overlap_size: int):
    return [shard_snapshots(n_snapshots, world_size, r, snapshot_size, overlap_size) for r in range(world_size)]


def job_throughput(units_per_rank, seconds_per_rank) -> float:
    """Whole-job rate the benchmark reports: all units / the slowest rank's time."""
    return float(sum(units_per_rank)) / max(seconds_per_rank)
