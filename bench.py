#!/usr/bin/env python3
"""bench.py — DoA snapshots/s through autocorrelate -> MUSIC_lin_array -> find_local_max on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W.  N > 1: one rank per GPU, RCCL only for
the barrier / max-over-ranks; the ranks come either from a launcher (torch.distributed.run: WORLD_SIZE is
set) or, when there is none, from bench.py itself, which starts N child processes before it touches the
GPU (doa/launch.py).  Prints ONE JSON line from rank 0.

Workload = BASELINE.json configs[1]: 4-element ULA, 1 source, 1024-sample snapshots (overlap 0),
1024-point spectrum, batch = 4096 snapshots per step, complex fp32 streams resident in HBM,
synthetic seeded data (doa.sim).  One step = one pass of the whole hot path over one batch:
covariance (K1), Hermitian EVD + noise projector (K2+K3), spectrum scan (K4), peak pick (K5); the
4 KiB spectrum of every snapshot is written out as the flowgraph does.  Steps rotate over several
distinct input/output batches whose total footprint exceeds the 256 MiB Infinity Cache, so every
step streams from HBM (a single 144 MiB working set would be served on-die).

The K timed steps go through ONE pipeline handle in ONE call of doa_music_pipeline_work_dev_batches (K batches): the
library spreads the batches over its own lanes (default 4 HIP streams + workspaces owned by the handle), so that the
HBM-bound covariance of one batch overlaps the issue-bound EVD / scan of its neighbours; the call is made in its
detached form and joined by the device synchronize that closes the timed region.  --mode streams is the round-1/2 arrangement for comparison: four handles on four
caller-created streams, one work_dev call per step.

Multi-GPU (weak scaling): snapshots are independent, so every rank owns its own batch and there is
no data-path collective; value = (steps * batch * world) / max-over-ranks time.  Timing: barrier + synchronize, clock
on, K steps, synchronize, clock off, barrier, MAX over ranks of the per-rank times (the closing collective's own latency is
not charged to the steps; the slowest rank decides).  After the timed region the
ranks also run the product's sharded driver (doa.distributed.run_sharded: one stream cut into per-rank
shards with their overlap halo, results gathered over RCCL) and rank 0 reports it as "sharded_run".
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]

N_ANT, K_SNAP, P_SPEC, M_SRC, BATCH = 4, 1024, 1024, 1, 4096
NORM_SPACING, SNR_DB = 0.5, 20.0
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def algorithmic_bytes():
    """Per snapshot, SURVEY §8(d): what each stage must read + write."""
    cov = N_ANT * K_SNAP * 8 + N_ANT * N_ANT * 8           # K1: 32768 + 128
    evd = N_ANT * N_ANT * 8 * 2                              # K2+K3: 128 + 128
    scan = N_ANT * N_ANT * 8 + P_SPEC * 4                    # K4: 128 + 4096
    peak = P_SPEC * 4 + 2 * M_SRC * 4                        # K5: 4096 + 8
    # what the fused pipeline actually has to move: K5 runs on the spectrum in registers (no re-read) and the
    # EVD hands the scan a 2N-double coefficient record instead of the N x N projector
    rec = 2 * N_ANT * 8
    fused = cov + (N_ANT * N_ANT * 8 + rec) + (rec + P_SPEC * 4) + 2 * M_SRC * 4
    return {"cov": cov, "evd": evd, "scan": scan, "peak": peak, "total": cov + evd + scan + peak,
            "fused_total": fused, "scan_fused": rec + P_SPEC * 4 + 2 * M_SRC * 4}


SCAN_KERNEL = "music_scan_peak1_kernel<4, 4, double, false, true, true, 0>"
COV_KERNEL = "cov_wave_kernel<4, true, 4, true>"


def scan_kernel_roofline(doa, torch, st, batch, reps=30):
    """The spectrum-scan kernel (K4, with K5 fused: the kernel the north star grades) in isolation: coefficient
    records are produced once by a real K1 -> EVD pass over short (64-sample) snapshots, then only the scan launch
    is repeated (doa_music_pipeline_set_stages) and timed with HIP events on its launch stream.  batch = 4096 is
    the benchmark step's own launch (ramp/tail-bound: 1024 waves x 4 items); batch = 262144 (1 GiB of spectra) is
    the size at which launch ramp and tail stop mattering."""
    k_short = 64
    pipe = doa.music_pipeline(N_ANT, k_short, 0, 0, NORM_SPACING, M_SRC, P_SPEC, batch)
    nb = 2 if batch > 65536 else 8                      # rotate outputs: > 256 MiB at either size
    s, _ = doa.sim.make_batch_streams_torch(N_ANT, k_short, batch, NORM_SPACING, M_SRC, SNR_DB, seed=77, device="cuda")
    ptrs = [t.data_ptr() for t in s]
    cov = torch.empty((batch, N_ANT * N_ANT), dtype=torch.complex64, device="cuda")
    spec = [torch.empty((batch, P_SPEC), dtype=torch.float32, device="cuda") for _ in range(nb)]
    mx = torch.empty((batch, M_SRC), dtype=torch.float32, device="cuda")
    am = torch.empty((batch, M_SRC), dtype=torch.float32, device="cuda")
    run = lambda i=0: pipe.work_dev(batch, ptrs, cov.data_ptr(), spec[i % nb].data_ptr(), mx.data_ptr(), am.data_ptr(), st)
    run()
    torch.cuda.synchronize()
    pipe.set_stages(cov=False, evd=False, scan=True)   # only the scan launch from here on, on the valid records
    try:
        for i in range(30):                      # sustained warm-up: the first ~50 launches run measurably slower
            run(i)
        groups = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(st)
            for i in range(reps):
                run(i)
            e1.record(st)
            torch.cuda.synchronize()
            groups.append(e0.elapsed_time(e1) * 1e3 / reps)
    finally:
        pipe.set_stages()
    us = sorted(groups)[len(groups) // 2]             # median of 5 groups of `reps` back-to-back launches
    nbytes = algorithmic_bytes()["scan_fused"] * batch
    gbs = nbytes / (us * 1e-6) / 1e9
    traffic = pmc_traffic(SCAN_KERNEL, batch)
    return {"batch": batch, "avg_launch_us": us, "algorithmic_bytes_per_launch": nbytes, "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "spectra_per_s": batch / (us * 1e-6),
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None,
            "traffic_command": traffic["command"] if traffic else None, "group_avgs_us": groups}


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` at `batch` items from the committed rocprofv3 PMC summary of this round
    (profiles/*_pmc_hbm_traffic.csv: separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per
    MI355X_MICROARCH.md, section HBM; every row carries the command line it was collected with).  PMC counters cannot
    be collected from inside this process; a row that does not name this kernel AND this batch is not used."""
    import csv
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.csv")), reverse=True):
        rows = list(csv.DictReader(open(f)))
        if not rows or "command" not in rows[0] or "batch" not in rows[0]:
            continue
        for r in rows:
            if kernel in r["kernel"] and int(r["batch"]) == int(batch):
                return {"hbm_bytes_per_launch": float(r["hbm_bytes_per_launch"]), "source": os.path.relpath(f, ROOT),
                        "command": r["command"]}
    return None


def cpu_baseline(budget_s=12.0):
    """Times the C restatement of the reference path (oracle/doa_oracle.c, kind "port") on this
    box's host cores over a bounded sample of the same workload.  When scipy's bundled OpenBLAS is
    present the port calls the very BLAS/LAPACK routines Armadillo forwards to (cgemm, cheevd)."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_doa.so")
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    lapack = False
    try:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        if cands and lib.oracle_use_lapack(cands[0].encode(), b"scipy_") == 0:
            lapack = True
    except Exception:
        pass
    from doa import sim
    n_cal = 512
    x, _ = sim.make_batch_streams(N_ANT, K_SNAP, n_cal, NORM_SPACING, M_SRC, SNR_DB, seed=7)

    def run(xs, n, threads):
        lib.oracle_set_num_threads(threads)
        ptrs = (ctypes.c_void_p * N_ANT)(*[xs[k].ctypes.data for k in range(N_ANT)])
        R = np.empty((n, N_ANT * N_ANT), np.complex64)
        spec = np.empty((n, P_SPEC), np.float32)
        mv = np.empty((n, M_SRC), np.float32)
        am = np.empty((n, M_SRC), np.float32)
        t0 = time.perf_counter()
        rc = lib.oracle_music_pipeline(ptrs, N_ANT, K_SNAP, 0, 0, ctypes.c_float(NORM_SPACING), M_SRC, P_SPEC, n,
                                       R.ctypes.data_as(ctypes.c_void_p), spec.ctypes.data_as(ctypes.c_void_p),
                                       mv.ctypes.data_as(ctypes.c_void_p), am.ctypes.data_as(ctypes.c_void_p))
        dt = time.perf_counter() - t0
        assert rc == n, rc
        return dt

    cores, host = host_cores()
    cores = max(1, min(cores, int(os.environ.get("DOA_CPU_THREADS", str(cores)))))
    run(x, n_cal, cores)                                   # warm-up (page faults, OpenMP pool)
    rate1 = n_cal / run(x, n_cal, 1)
    rate_all_est = n_cal / run(x, n_cal, cores)
    # bounded sample: ~budget_s of CPU work split between the all-core, the GNU Radio model and the single-thread run
    n_all = int(min(max(rate_all_est * budget_s * 0.5, n_cal), 65536))
    reps = max(1, n_all // n_cal)
    xs, _ = sim.make_batch_streams(N_ANT, K_SNAP, n_cal * min(reps, 16), NORM_SPACING, M_SRC, SNR_DB, seed=8)
    n_run = xs.shape[1] // K_SNAP
    times = sorted(run(xs, n_run, cores) for _ in range(max(3, reps // 16)))
    rate_all = n_run / times[len(times) // 2]
    n1 = int(min(max(rate1 * budget_s * 0.2, 128), n_run))
    rate1 = n1 / run(xs[:, : n1 * K_SNAP].copy(), n1, 1)
    n_gr = int(min(max(rate1 * budget_s * 0.3, 256), n_run))
    gr_rate, gr_stage = gr_model(lib, xs, n_gr)
    return {"value": rate_all, "unit": "snapshots/s", "cores": cores, "kind": "port",
            "sample": f"{n_run} snapshots x{len(times)} (median), same N=4/K=1024/P=1024 workload, all {cores} host "
                      f"threads via OpenMP; eig/gemm = {'LAPACK cheevd + BLAS cgemm (scipy OpenBLAS)' if lapack else 'built-in Jacobi / loops'}",
            "host": host, "single_thread_value": rate1,
            "gr_model_value": gr_rate,
            "gr_model": f"GNU Radio's execution model (thread per block): autocorrelate | MUSIC_lin_array | find_local_max as three "
                        f"pipelined single-threaded stages over {n_gr} snapshots in scheduler-sized calls of 16 items; "
                        f"per-stage busy fractions {gr_stage}"}


def host_cores():
    """(threads this process may really use, description).  A one-GPU box sees every logical CPU of its host but is
    held to a cgroup CPU quota; more threads than that quota only get throttled."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    use = min(n, quota) if quota else n
    return use, f"{model}; {n} logical CPUs visible, cgroup cpu quota {quota if quota else 'none'} -> {use} threads used"


def gr_model(lib, xs, n, chunk=16):
    """the reference's blocks under GNU Radio's thread-per-block scheduler -- one thread per block,
    three blocks pipelined through bounded buffers.  Each stage is the C port's own per-block function, single-threaded;
    ctypes releases the GIL during the calls, so the three Python threads really run side by side."""
    import queue
    import threading
    import numpy as np
    K, N, P, M = K_SNAP, N_ANT, P_SPEC, M_SRC
    n = (n // chunk) * chunk
    R = np.empty((n, N * N), np.complex64)
    spec = np.empty((n, P), np.float32)
    mv = np.empty((n, M), np.float32)
    am = np.empty((n, M), np.float32)
    q1, q2 = queue.Queue(maxsize=4), queue.Queue(maxsize=4)      # the scheduler's output buffers
    busy = [0.0, 0.0, 0.0]
    vp = ctypes.c_void_p

    def stage_cov():
        lib.oracle_set_num_threads(1)
        for c0 in range(0, n, chunk):
            ptrs = (vp * N)(*[xs[k].ctypes.data + 8 * c0 * K for k in range(N)])
            t = time.perf_counter()
            rc = lib.oracle_autocorrelate(ptrs, N, K, 0, 0, chunk, vp(R.ctypes.data + 8 * N * N * c0))
            busy[0] += time.perf_counter() - t
            assert rc == chunk, rc
            q1.put(c0)
        q1.put(None)

    def stage_music():
        lib.oracle_set_num_threads(1)
        while (c0 := q1.get()) is not None:
            t = time.perf_counter()
            rc = lib.oracle_music_lin_array(vp(R.ctypes.data + 8 * N * N * c0), chunk, ctypes.c_float(NORM_SPACING), M, N, P,
                                            vp(spec.ctypes.data + 4 * P * c0))
            busy[1] += time.perf_counter() - t
            assert rc == chunk, rc
            q2.put(c0)
        q2.put(None)

    def stage_peak():
        lib.oracle_set_num_threads(1)
        while (c0 := q2.get()) is not None:
            t = time.perf_counter()
            rc = lib.oracle_find_local_max(vp(spec.ctypes.data + 4 * P * c0), chunk, M, P, ctypes.c_float(0.0),
                                           ctypes.c_float(180.0), vp(mv.ctypes.data + 4 * M * c0), vp(am.ctypes.data + 4 * M * c0))
            busy[2] += time.perf_counter() - t
            assert rc == chunk, rc

    th = [threading.Thread(target=f) for f in (stage_cov, stage_music, stage_peak)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return n / dt, [round(b / dt, 2) for b in busy]


def sharded_run(doa, torch, dist, world, rank, local_rank, per_rank=4096, K=1024, ovl=256, theta=(41.0, 117.0)):
    """The product's sharded driver (doa.distributed.run_sharded) over ONE simulated stream of world x per_rank overlapping
    windows (two fixed sources, forward-backward averaging), as a measurement: every rank generates only its own shard +
    halo (seekable device generator), runs its pipeline, and the per-snapshot angle pairs meet in one all_gather --
    generation, compute and gather timed separately (max over ranks).  With world > 1 the ingest-rank variant is timed too:
    rank 0 holds the whole stream and doa.distributed.scatter_shards sends every other rank its shard, halo included
    (configs[4]'s "RCCL scatter/gather over xGMI").  Correctness = every gathered pair sits on the two source directions
    and the scattered shard equals the generated one bit for bit."""
    n_total, S = world * per_rank, K - ovl
    dev = torch.device("cuda", local_rank)
    st = torch.cuda.current_stream()
    local_error = []
    t = {}

    def sync_time():
        torch.cuda.synchronize()
        return time.perf_counter()

    def generate(begin, end):
        bufs = [torch.zeros(end - begin, dtype=torch.complex64, device=dev) for _ in range(N_ANT)]
        try:
            src = doa.sim_source(N_ANT, 0.45, list(theta), [0.031, 0.047], None, None, 0.1, seed=99)
            src.seek(begin)
            src.work_dev(end - begin, [b.data_ptr() for b in bufs], st)
        except Exception as e:
            local_error.append(repr(e))
        return bufs

    def my_samples(begin, end):
        t0 = sync_time()
        bufs = generate(begin, end)
        t["generation_s"] = sync_time() - t0
        t["kept"] = bufs
        return bufs

    pipe_box = {}

    def compute(bufs, n_local):
        # never raises: a rank that failed locally must still enter the all_gather its peers are waiting in
        am = torch.full((n_local, 2), float("nan"), dtype=torch.float32, device=dev)
        try:
            pipe = pipe_box.setdefault("p", doa.music_pipeline(N_ANT, K, ovl, 1, 0.45, 2, P_SPEC, max(n_local, 1)))
            mx = torch.empty((n_local, 2), dtype=torch.float32, device=dev)
            args = (n_local, [b.data_ptr() for b in bufs], 0, 0, mx.data_ptr(), am.data_ptr(), st)
            pipe.work_dev(*args)                          # first call: lazy workspaces
            t0 = sync_time()
            pipe.work_dev(*args)
            t["compute_s"] = sync_time() - t0
        except Exception as e:
            local_error.append(repr(e))
        return am

    local, shard = doa.distributed.run_sharded(my_samples, n_total, K, ovl, compute, dist=dist, gather=False)
    shards = doa.sharding.all_shards(n_total, world, K, ovl)
    doa.distributed.gather_results(local, shards, dist)                 # first gather: communicator set-up
    t0 = sync_time()
    angles = doa.distributed.gather_results(local, shards, dist)
    t["gather_s"] = sync_time() - t0
    a = angles.cpu().numpy()
    err = float(max(abs(a[:, 0] - max(theta)).max(), abs(a[:, 1] - min(theta)).max()))     # port 1 is sorted descending
    red = lambda v: doa.distributed.max_over_ranks(float(v), device="cuda", dist=dist)
    out = {"snapshots": n_total, "snapshots_per_rank": per_rank, "ranks": world, "halo_samples": ovl,
           "shard_samples_rank0": shard.n_samples if rank == 0 else None,
           "gathered_rows": int(a.shape[0]), "max_angle_error_deg": err, "ok": bool(a.shape[0] == n_total and err <= 1.0),
           "generation_s": red(t.get("generation_s", float("nan"))), "compute_s": red(t.get("compute_s", float("nan"))),
           "gather_s": red(t.get("gather_s", float("nan"))),
           "gather_bytes_per_rank": int(per_rank * 2 * 4)}
    out["compute_snapshots_per_s"] = n_total / out["compute_s"] if out["compute_s"] == out["compute_s"] else None
    if world > 1:
        # ingest-rank variant: rank 0 generates the WHOLE stream, the shards travel (input side of configs[4])
        try:
            whole = generate(0, (n_total - 1) * S + K) if rank == 0 else None
            doa.distributed.scatter_shards(whole, N_ANT, n_total, K, ovl, src=0, device=dev, dist=dist)      # set-up pass
            t0 = sync_time()
            mine = doa.distributed.scatter_shards(whole, N_ANT, n_total, K, ovl, src=0, device=dev, dist=dist)
            t_sc = red(sync_time() - t0)
            same = all(bool(torch.equal(torch.view_as_real(m).cpu(), torch.view_as_real(k).cpu())) for m, k in zip(mine, t["kept"]))
            sent = sum(sh.n_samples for sh in shards[1:]) * N_ANT * 8
            out["scatter"] = {"seconds": t_sc, "bytes_sent_by_rank0": int(sent), "GBs": sent / t_sc / 1e9,
                              "shards_equal_generated": bool(same)}
            del whole, mine
        except Exception as e:
            out["scatter"] = {"error": repr(e)}
    if local_error:
        out["rank0_error"] = local_error[0]
    return out


def other_configs(doa, torch, st, lanes=4, check=True):
    """Driver-timed figures for the other BASELINE.json configs (parity-test cases, never the headline): configs[2] (Root-MUSIC,
    N=4, 2 sources), configs[3] (N=16, 3 sources, P=4096, MFMA covariance) and the simulation flowgraph's shape (K=2048,
    overlap 512, forward-backward, 2 sources).  Each: us per 4096-snapshot step serial (one stream) and overlapped (the
    handle's lanes: doa_music_pipeline / doa_root_pipeline work_dev_batches) -- as ONE call of 20 (cfg4: 8) batches, the figure
    comparable with round 3, and as one call of 100 (cfg4: 40), where fill, drain and the host's launch / synchronise latency are
    amortised as they are in a running stream --, items/s; preceded -- outside any timing -- by a
    16-row spot check against the oracle (the checker leg of this script, like cpu_baseline: fp64 restatement of the
    reference's formulas on the same samples)."""
    import numpy as np
    B = 4096
    out = {}

    def timed(fn, reps):
        fn(reps)                                      # warm-up incl. any argument marshalling the closure caches
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            fn(reps)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        return best

    def spot(name, streams, N, K, ovl, fb, d, M, P, cov_t, am_t, root_t):
        if not check:
            return None
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import doa_oracle as oracle
        n = 16
        span = (n - 1) * (K - ovl) + K
        x = np.stack([t[:span].cpu().numpy() for t in streams])
        R64 = oracle.autocorrelate(x, K, ovl, fb, n, precision="f64")
        Rg = cov_t[:n].cpu().numpy()
        res = {"rows": n, "cov_rel_err": float(np.abs(Rg - R64).max() / np.abs(R64).max())}
        ok = res["cov_rel_err"] <= (2e-6 if N <= 8 else 4e-6)       # (N > 8: fp32 MFMA accumulation over K products)
        if am_t is not None:
            s64 = oracle.music_lin_array(Rg, d, M, N, P, "f64")
            _, loc = oracle.find_local_max(s64.astype(np.float32), M, P, 0.0, 180.0)
            res["peak_location_max_diff_deg"] = float(np.abs(am_t[:n].cpu().numpy() - loc).max())
            ok = ok and res["peak_location_max_diff_deg"] <= 180.0 / P + 1e-3
        if root_t is not None:
            a64 = oracle.root_music(Rg, d, M, N, "f64")
            res["root_angle_max_diff_deg"] = float(np.abs(root_t[:n].cpu().numpy() - a64).max())
            ok = ok and res["root_angle_max_diff_deg"] <= 1e-3
        res["ok"] = bool(ok)
        return res

    def pipeline_config(name, N, K, ovl, fb, d, M, P, nbuf, reps, steady, desc):
        S = K - ovl
        span = (B - 1) * S + K
        bufs = []
        for b in range(nbuf):
            if ovl == 0:
                s, _ = doa.sim.make_batch_streams_torch(N, K, B, d, M, SNR_DB, seed=500 + b, device="cuda")
            else:           # one continuous stream with overlapping windows: the simulation flowgraph's own generator
                s = [torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)]
                s = doa.sim.stream_slab_torch(s)
                src = doa.sim_source(N, d, [30.0, 123.0][:M], [0.03125, 0.0625][:M], None, None, 0.1, seed=600 + b)
                src.work_dev(span, [t.data_ptr() for t in s], st)
                del src                               # (the generator owns a stream; nothing of the set-up stays alive in the timed part)
            bufs.append(s)
        ptrs = [[t.data_ptr() for t in s] for s in bufs]
        cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
        spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, B)
        pipe.set_lanes(lanes)
        serial = lambda n: [pipe.work_dev(B, ptrs[i % nbuf], cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(),
                                          am[i % nbuf].data_ptr(), st) for i in range(n)]
        calls = {}
        def lanes_fn(n):
            if n not in calls:                        # pointer arrays marshalled once, as a C caller has them
                idx = [i % nbuf for i in range(n)]
                calls[n] = pipe.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [spec[b].data_ptr() for b in idx],
                                                [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], doa.DETACHED)
            calls[n]()
        serial(1)
        torch.cuda.synchronize()
        chk = spot(name, bufs[0], N, K, ovl, fb, d, M, P, cov[0], am[0], None)
        us_serial = timed(serial, reps)
        ok_lanes = nbuf % lanes == 0 or nbuf >= lanes
        us_lanes = timed(lanes_fn, reps) if ok_lanes else None
        us_steady = timed(lanes_fn, steady) if ok_lanes else None       # a longer call: fill, drain and host latency amortised
        out[name] = {"config": desc, "batch": B, "us_per_step_serial": us_serial, "us_per_step_overlapped": us_lanes,
                     "overlapped_steps_per_call": reps, "us_per_step_overlapped_steady": us_steady, "steady_steps_per_call": steady,
                     "items_per_s_serial": B / us_serial * 1e6, "items_per_s_overlapped": (B / us_lanes * 1e6) if us_lanes else None,
                     "items_per_s_overlapped_steady": (B / us_steady * 1e6) if us_steady else None,
                     "overlap": f"doa_music_pipeline_work_dev_batches (detached), {lanes} lanes", "spot_check": chk}

    # configs[2]: covariance + Root-MUSIC through ONE handle (doa_root_pipeline, round 4): serial = one work_dev per step on one
    # stream, overlapped = all steps as one doa_root_pipeline_work_dev_batches call over the handle's lanes (detached form)
    try:
        N, K, d, M = 4, 1024, 0.44, 2
        nbuf = 8                                      # rotating input sets, as the headline's (more than the 256 MiB L3 holds)
        bufs = [doa.sim.make_batch_streams_torch(N, K, B, d, M, SNR_DB, seed=400 + b, device="cuda")[0] for b in range(nbuf)]
        ptrs = [[t.data_ptr() for t in s] for s in bufs]
        cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
        ang = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        rp = doa.root_pipeline(N, K, 0, 0, d, M, B)
        rp.set_lanes(lanes)
        def serial(n):
            for i in range(n):
                b = i % nbuf
                rp.work_dev(B, ptrs[b], cov[b].data_ptr(), ang[b].data_ptr(), None, st)
        rcalls = {}
        def overl(n):
            if n not in rcalls:                       # pointer arrays marshalled once, as a C caller has them
                idx = [i % nbuf for i in range(n)]
                rcalls[n] = rp.prepare_batches(B, [ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx], [ang[b].data_ptr() for b in idx],
                                               None, doa.DETACHED)
            rcalls[n]()                               # (timed() joins with a device synchronise, as for the MUSIC handle)
        serial(1)
        torch.cuda.synchronize()
        chk = spot("cfg3", bufs[0], N, K, 0, 0, d, M, 0, cov[0], None, ang[0])
        us_s, us_o, us_st = timed(serial, 20), timed(overl, 20), timed(overl, 100)
        out["cfg3_root_music"] = {"config": "BASELINE.json configs[2]: N=4, 2 sources, d=0.44, K=1024, covariance + Root-MUSIC", "batch": B,
                                  "us_per_step_serial": us_s, "us_per_step_overlapped": us_o, "overlapped_steps_per_call": 20,
                                  "us_per_step_overlapped_steady": us_st, "steady_steps_per_call": 100, "items_per_s_serial": B / us_s * 1e6,
                                  "items_per_s_overlapped": B / us_o * 1e6, "items_per_s_overlapped_steady": B / us_st * 1e6,
                                  "overlap": f"doa_root_pipeline_work_dev_batches (detached), {lanes} lanes, one call for all steps",
                                  "spot_check": chk}
        del bufs, cov, ang, rp
    except Exception as e:
        out["cfg3_root_music"] = {"error": repr(e)}
    for args_ in (("flowgraph_shape", 4, 2048, 512, 1, 0.4, 2, 1024, 8, 20, 100,
                   "run_MUSIC_lin_array_simulation.grc's shape: N=4, 2 sources, d=0.4, K=2048, overlap 512, forward-backward, P=1024"),
                  ("cfg4_n16", 16, 1024, 0, 0, 0.5, 3, 4096, 8, 8, 40,
                   "BASELINE.json configs[3]: N=16, 3 sources, K=1024, P=4096 (MFMA covariance, subspace EVD, LDS-row scan)")):
        try:
            pipeline_config(*args_)
        except Exception as e:
            out[args_[0]] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def load_launcher():
    """doa/launch.py by path: importing the `doa` package would load the HIP library into this process."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("doa_launch", os.path.join(ROOT, "gr-doa_amd", "python", "doa", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64),
                    help="internal precision of EVD + scan (items are fp32 either way)")
    ap.add_argument("--nbuf", type=int, default=6, help="distinct batches rotated through (defeats L3 residency)")
    ap.add_argument("--streams", type=int, default=4, help="lanes of the pipeline handle (--mode batches) / caller streams (--mode streams)")
    ap.add_argument("--mode", choices=("batches", "streams"), default="batches",
                    help="batches: the K steps as ONE doa_music_pipeline_work_dev_batches call on one handle (the library owns the "
                         "overlap); streams: one work_dev call per step on caller-created streams, one handle each (rounds 1-2)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the driver-timed secondary configs (cfg3, cfg4, flowgraph shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scan-roofline", action="store_true",
                    help="skip the isolated scan-kernel measurements (keeps rocprof kernel averages clean)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help="with --gpus N > 1: print the N rank commands and exit")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start one rank per GPU ourselves, BEFORE anything here touches torch or HIP (a
        # process that has initialised the GPU must never be replaced, and this parent never is: it only waits)
        launch = load_launcher()
        argv = [a for a in sys.argv[1:] if a != "--dry-run"]
        if args.dry_run:
            for cmd, env in launch.rank_commands(os.path.abspath(__file__), argv, args.gpus):
                print(" ".join(f"{k}={v}" for k, v in sorted(env.items())), " ".join(cmd))
            return
        sys.exit(launch.launch_ranks(os.path.abspath(__file__), argv, args.gpus, timeout=1500))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if os.environ.get("DOA_BENCH_SHARE_GPU"):        # rehearsal only: several ranks on one card (with DOA_BENCH_BACKEND=gloo)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import doa
    dist = None
    if world > 1 or os.environ.get("DOA_BENCH_FORCE_DIST"):      # the switch rehearses the RCCL code path with one rank
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(doa.launch.free_port())
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        dist = doa.distributed.init_process_group(os.environ.get("DOA_BENCH_BACKEND", "nccl"),
                                                  device=torch.device("cuda", local_rank))
        if world != args.gpus and rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting {world}",
                  file=sys.stderr)
    doa.set_internal_precision(args.precision)
    n_streams = max(1, args.streams)
    batched = (args.mode == "batches")
    # --mode batches: ONE handle, its lanes are the streams; --mode streams: one handle per caller stream
    pipes = [doa.music_pipeline(N_ANT, K_SNAP, 0, 0, NORM_SPACING, M_SRC, P_SPEC, BATCH) for _ in range(1 if batched else n_streams)]
    if batched:
        pipes[0].set_lanes(n_streams)
    hip_streams = [torch.cuda.Stream() for _ in range(n_streams)]
    cov_blk = doa.autocorrelate(N_ANT, K_SNAP, 0, 0)
    music_blk = doa.MUSIC_lin_array(NORM_SPACING, M_SRC, N_ANT, P_SPEC)
    peak_blk = doa.find_local_max(M_SRC, P_SPEC, 0.0, 180.0)

    # ---- synthetic, device-resident inputs (setup, not timed) -------------------------------------
    # a multiple of the lane count: step i and step i + nbuf (same buffers) then run on the SAME lane / stream, i.e. in
    # order -- no two steps ever write one buffer set concurrently
    nbuf = max(n_streams + 1, args.nbuf)
    nbuf = ((nbuf + n_streams - 1) // n_streams) * n_streams
    streams, thetas = [], []
    for b in range(nbuf):
        s, th = doa.sim.make_batch_streams_torch(N_ANT, K_SNAP, BATCH, NORM_SPACING, M_SRC, SNR_DB,
                                                 seed=1000 * rank + b, device=f"cuda:{local_rank}")
        streams.append(s)
        thetas.append(th)
    in_ptrs = [[t.data_ptr() for t in s] for s in streams]
    spec = [torch.empty((BATCH, P_SPEC), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    cov = [torch.empty((BATCH, N_ANT * N_ANT), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
    mx = [torch.empty((BATCH, M_SRC), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    am = [torch.empty((BATCH, M_SRC), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    st = torch.cuda.current_stream()

    def step(i):                                      # --mode streams: one work_dev call per step on a caller stream
        b, k = i % nbuf, i % n_streams
        pipes[k].work_dev(BATCH, in_ptrs[b], cov[b].data_ptr(), spec[b].data_ptr(), mx[b].data_ptr(),
                          am[b].data_ptr(), hip_streams[k])

    prepared = {}

    def prepare(first, count, with_spectrum=True):
        """marshal the pointer arrays of one library call (a C caller has them at hand) -- host work, done for EVERY call of
        this run before the first step is launched, so that nothing but the barrier sits between warm-up and timed region"""
        key = (first, count, with_spectrum)
        if batched and count > 0 and key not in prepared:
            idx = [i % nbuf for i in range(first, first + count)]
            prepared[key] = pipes[0].prepare_batches(BATCH, [in_ptrs[b] for b in idx], [cov[b].data_ptr() for b in idx],
                                                     [spec[b].data_ptr() for b in idx] if with_spectrum else None,
                                                     [mx[b].data_ptr() for b in idx], [am[b].data_ptr() for b in idx], doa.DETACHED)

    def run_steps(first, count, with_spectrum=True):
        """`count` steps starting at step index `first` (buffer set = step index mod nbuf)."""
        if not batched:
            for i in range(first, first + count):
                if with_spectrum:
                    step(i)
                else:
                    b, k = i % nbuf, i % n_streams
                    pipes[k].work_dev(BATCH, in_ptrs[b], cov[b].data_ptr(), 0, mx[b].data_ptr(), am[b].data_ptr(), hip_streams[k])
            return
        # detached: the inputs are resident and complete (synchronize before the region), the join is the contract's own
        # torch.cuda.synchronize() after the K steps -- no cross-stream event inside the timed region (they cost ~150 us per fork + join on
        # this runtime).
        prepare(first, count, with_spectrum)
        prepared[(first, count, with_spectrum)]()   # (every caller below follows with torch.cuda.synchronize(): the device-wide join)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # host-side marshalling of every call below, before anything is launched
    n_setup = max(n_streams, nbuf)
    for call in ((0, n_setup, True), (0, args.warmup, True), (0, args.steps, True), (0, 8, False), (0, args.steps, False),
                 (max(0, args.steps - n_streams), min(n_streams, args.steps), True)):
        prepare(*call)
    # setup, not warm-up: touch every lane / stream / buffer set once, so that first-launch code loading and lazy
    # workspace allocation never fall into a short timed region whatever --warmup says -- and the process group's first
    # collectives (kernel loading, channel set-up) likewise: the barrier in front of the region is then a warm one
    run_steps(0, n_setup)
    torch.cuda.synchronize()
    for _ in range(3):
        barrier()
    if args.warmup > 0:
        run_steps(0, args.warmup)
    # Timed region: barrier + synchronize on both sides.  Each rank stops its clock when ITS K steps have completed
    # (synchronize), the closing barrier follows, and the job's time is the MAX over ranks -- the time from the common
    # start to the slowest rank's completion, without the closing collective's own latency (tens of microseconds of
    # RCCL launch + ring on 8 GPUs would otherwise be charged to a 20-step region of ~0.6 ms).
    barrier()
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        elapsed = doa.distributed.max_over_ranks(elapsed, device="cuda", dist=dist)

    # Secondary figure, never `value`: the same steps with no spectrum pointer (the fused block with only its angle port
    # connected: the 4 KiB row per snapshot is neither converted to dB nor written).
    run_steps(0, 8, with_spectrum=False)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    run_steps(0, args.steps, with_spectrum=False)
    torch.cuda.synchronize()
    angles_elapsed = time.perf_counter() - ta
    run_steps(max(0, args.steps - n_streams), min(n_streams, args.steps))    # leave full results (spectra included) behind
    torch.cuda.synchronize()

    # Not part of the timed region (snapshots are independent: the path has no data-path collective): the sharded
    # run itself, product code (doa.distributed.run_sharded), as a measurement of its own (sharded_run's docstring).
    sharded = None
    if dist is not None:
        torch.cuda.synchronize()
        try:
            sharded = sharded_run(doa, torch, dist, world, rank, local_rank)
        except Exception as e:                      # never lose the throughput number to the check
            sharded = {"error": repr(e)}

    # sanity: the estimates of the last batch must sit on the directions they were generated with
    import numpy as np
    last = (args.steps - 1) % nbuf
    est = am[last].cpu().numpy()[:, 0]
    ang_err = float(np.abs(est - thetas[last][:, 0]).max())
    if not ang_err <= 1.0:
        raise SystemExit(f"bench sanity check failed: max angle error {ang_err} deg")

    # ---- per-kernel timing with HIP events on the launch stream (after the timed region) -----------
    def time_stage(fn, reps):
        for i in range(5):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        for i in range(reps):
            fn(i)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps           # us per launch (includes the ~1.5 us launch gap)

    reps = max(20, min(args.steps, 200))
    t_cov = time_stage(lambda i: cov_blk.work_dev(BATCH, in_ptrs[i % nbuf], cov[i % nbuf].data_ptr(), st), reps)
    t_music = time_stage(lambda i: music_blk.work_dev(BATCH, cov[i % nbuf].data_ptr(), spec[i % nbuf].data_ptr(), st), reps)
    t_peak = time_stage(lambda i: peak_blk.work_dev(BATCH, spec[i % nbuf].data_ptr(), mx[i % nbuf].data_ptr(),
                                                    am[i % nbuf].data_ptr(), st), reps)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ab = algorithmic_bytes()
    total_snap = args.steps * BATCH * world
    value = total_snap / elapsed
    cov_gbs = ab["cov"] * BATCH / (t_cov * 1e-6) / 1e9
    traffic = pmc_traffic(COV_KERNEL, BATCH)
    out = {
        "metric": "DoA snapshots/sec (autocorr+MUSIC+peak) @ N=4, 1024 samp, 1024 angles",
        "value": value,
        "unit": "snapshots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "angles_only_mode": {"value": args.steps * BATCH / angles_elapsed, "unit": "snapshots/s (this rank)",
                             "ms_per_step": angles_elapsed / args.steps * 1e3,
                             "note": "secondary: same steps with no spectrum output requested (spectrum neither converted to dB nor written); not the headline"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: 4-ch ULA, 1 source, K=1024 (overlap 0), P=1024, "
                               "batch=4096 snapshots/step, complex fp32, SNR 20 dB",
                   "batch": BATCH, "inputs": N_ANT, "snapshot_size": K_SNAP, "pspectrum_len": P_SPEC,
                   "num_targets": M_SRC, "internal_precision": args.precision, "rotating_batches": nbuf, "hip_streams": n_streams,
                   "step_entry": ("doa_music_pipeline_work_dev_batches (detached form): the K steps in one call on one handle, %d library-owned lanes; joined by the device synchronize that closes the timed region" % n_streams)
                                 if batched else "doa_music_pipeline_work_dev per step on %d caller streams, one handle each" % n_streams,
                   "input_layout": "N separate stream buffers in HBM, doa_stream_stride_bytes apart (4.5 KiB modulo 8 KiB: no HBM-channel aliasing between streams)",
                   "parallelism": f"snapshot-sharded x{world}, no data-path collective"},
        "pipeline_gbs": ab["fused_total"] * value / world / 1e9,      # fused algorithmic bytes x rate, per GPU
        "pipeline_gbs_unfused_accounting": ab["total"] * value / world / 1e9,   # SURVEY 8(d)'s 41 KB/snapshot
        # the dominant kernel (88 % of the step's bytes): K1, stand-alone launches on one stream
        "roofline": {"bound": "hbm", "kernel": COV_KERNEL + " (K1 covariance)",
                     "achieved": cov_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cov_gbs / HBM_PEAK_GBS,
                     "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                     "traffic_source": traffic["source"] if traffic else None,
                     "traffic_command": traffic["command"] if traffic else None,
                     "algorithmic_bytes_per_launch": ab["cov"] * BATCH, "avg_launch_us": t_cov},
        "kernels": {
            "K1_cov": {"us": t_cov, "GBs": cov_gbs},
            "K2-K4_music(evd+scan)": {"us": t_music, "GBs": (ab["evd"] + ab["scan"]) * BATCH / (t_music * 1e-6) / 1e9},
            "K5_peak": {"us": t_peak, "GBs": ab["peak"] * BATCH / (t_peak * 1e-6) / 1e9},
        },
        "kernels_note": "stand-alone block launches on one stream, HIP events; the timed pipeline fuses K5 into K4",
        "max_angle_error_deg": ang_err,
    }
    if dist is not None:
        out["world_size_seen_by_rccl"] = int(dist.get_world_size())
        out["sharded_run"] = sharded
    if world == 1 and args.precision == 64 and not args.no_scan_roofline:
        # the kernel the north star grades (>= 70 % of HBM on the spectrum scan): at the benchmark step's own batch and
        # at a batch where launch ramp / tail no longer matter
        rs = {"bound": "hbm", "kernel": SCAN_KERNEL + " (K4 scan + fused K5 peak pick)"}
        for key, b in (("at_benchmark_batch", BATCH), ("at_large_batch", 262144)):
            try:
                rs[key] = scan_kernel_roofline(doa, torch, st, b)
            except Exception as e:                      # secondary figure: never lose the headline to it
                rs[key] = {"error": repr(e)}
        out["roofline_scan"] = rs
    if world == 1 and args.precision == 64 and not args.no_other_configs:
        try:
            out["other_configs"] = other_configs(doa, torch, st, lanes=n_streams, check=not args.no_cpu_baseline)
        except Exception as e:
            out["other_configs"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:
        # in a child process: the CPU leg must never be able to take the GPU number down with it
        import subprocess
        try:
            env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_DYNAMIC="FALSE")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], env=env,
                               capture_output=True, text=True, timeout=240)
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
