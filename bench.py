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

Steps alternate over a few HIP streams (default 4, one pipeline handle = one workspace per
stream): each step's three launches stay in order on its own stream, while the HBM-bound covariance
of one batch overlaps the latency-bound EVD / scan of the batches before it (the covariance kernel fills
the register file of every SIMD, so the EVD / scan of step i really run beside the covariance of step
i + 2: three or four streams keep that from stalling the covariance of step i + 2; same-box A/B runs put four ahead
of three by 1-3 us per step more often than not).

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


SCAN_KERNEL = "music_scan_peak1_kernel<4, 4, double, false, true>"
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


def sharded_check(doa, torch, dist, world, local_rank, per_rank=512, K=1024, ovl=256, theta=(41.0, 117.0)):
    """doa.distributed.run_sharded over one simulated stream (two fixed sources, overlapping windows, FB averaging):
    returns what rank 0 reports.  Correctness = every gathered angle pair sits on the two source directions."""
    n_total, S = world * per_rank, K - ovl
    dev = torch.device("cuda", local_rank)

    def my_samples(begin, end):                     # this rank's shard only, generated in place
        bufs = [torch.zeros(end - begin, dtype=torch.complex64, device=dev) for _ in range(N_ANT)]
        try:
            src = doa.sim_source(N_ANT, 0.45, list(theta), [0.031, 0.047], None, None, 0.1, seed=99)
            src.seek(begin)
            src.work_dev(end - begin, [b.data_ptr() for b in bufs], torch.cuda.current_stream())
            torch.cuda.synchronize()
        except Exception as e:
            local_error.append(repr(e))
        return bufs

    local_error = []

    def compute(bufs, n_local):
        # never raises: a rank that failed locally must still enter the all_gather its peers are waiting in
        am = torch.full((n_local, 2), float("nan"), dtype=torch.float32, device=dev)
        try:
            pipe = doa.music_pipeline(N_ANT, K, ovl, 1, 0.45, 2, P_SPEC, max(n_local, 1))
            mx = torch.empty((n_local, 2), dtype=torch.float32, device=dev)
            pipe.work_dev(n_local, [b.data_ptr() for b in bufs], 0, 0, mx.data_ptr(), am.data_ptr(), torch.cuda.current_stream())
            torch.cuda.synchronize()
        except Exception as e:
            local_error.append(repr(e))
        return am

    t0 = time.perf_counter()
    angles, shard = doa.distributed.run_sharded(my_samples, n_total, K, ovl, compute, dist=dist)
    dt = time.perf_counter() - t0
    a = angles.cpu().numpy()
    err = float(max(abs(a[:, 0] - max(theta)).max(), abs(a[:, 1] - min(theta)).max()))     # port 1 is sorted descending
    return {"snapshots": n_total, "ranks": world, "halo_samples": ovl, "shard_samples_rank0": shard.n_samples,
            "gathered_rows": int(a.shape[0]), "max_angle_error_deg": err, "ok": bool(a.shape[0] == n_total and err <= 1.0),
            **({"rank0_error": local_error[0]} if local_error else {}),
            "seconds_incl_generation": dt}


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
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the steps alternate over")
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
    pipes = [doa.music_pipeline(N_ANT, K_SNAP, 0, 0, NORM_SPACING, M_SRC, P_SPEC, BATCH) for _ in range(n_streams)]
    hip_streams = [torch.cuda.Stream() for _ in range(n_streams)]
    cov_blk = doa.autocorrelate(N_ANT, K_SNAP, 0, 0)
    music_blk = doa.MUSIC_lin_array(NORM_SPACING, M_SRC, N_ANT, P_SPEC)
    peak_blk = doa.find_local_max(M_SRC, P_SPEC, 0.0, 180.0)

    # ---- synthetic, device-resident inputs (setup, not timed) -------------------------------------
    nbuf = max(n_streams + 1, args.nbuf)       # a buffer set is never reused while its step may be in flight
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

    def step(i):
        b, k = i % nbuf, i % n_streams
        pipes[k].work_dev(BATCH, in_ptrs[b], cov[b].data_ptr(), spec[b].data_ptr(), mx[b].data_ptr(),
                          am[b].data_ptr(), hip_streams[k])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not warm-up: touch every pipeline handle / stream / buffer set once, so that first-launch code
    # loading and lazy workspace allocation never fall into a short timed region whatever --warmup says
    for i in range(max(n_streams, nbuf)):
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    # Timed region: barrier + synchronize on both sides.  Each rank stops its clock when ITS K steps have completed
    # (synchronize), the closing barrier follows, and the job's time is the MAX over ranks -- the time from the common
    # start to the slowest rank's completion, without the closing collective's own latency (tens of microseconds of
    # RCCL launch + ring on 8 GPUs would otherwise be charged to a 20-step region of ~0.6 ms).
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        elapsed = doa.distributed.max_over_ranks(elapsed, device="cuda", dist=dist)

    # Secondary figure, never `value`: the same steps with no spectrum pointer (the fused block with only its angle port
    # connected: the 4 KiB row per snapshot is neither converted to dB nor written).
    def step_angles(i):
        b, k = i % nbuf, i % n_streams
        pipes[k].work_dev(BATCH, in_ptrs[b], cov[b].data_ptr(), 0, mx[b].data_ptr(), am[b].data_ptr(), hip_streams[k])
    for i in range(8):
        step_angles(i)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for i in range(args.steps):
        step_angles(i)
    torch.cuda.synchronize()
    angles_elapsed = time.perf_counter() - ta
    for i in range(n_streams):                        # leave full results (spectra included) in the buffers the checks read
        step(args.steps - 1 - i)
    torch.cuda.synchronize()

    # Not part of the timed region (snapshots are independent: the path has no data-path collective): the sharded
    # run itself, product code (doa.distributed.run_sharded) -- ONE stream of world x 512 overlapping windows is cut
    # into contiguous per-rank shards with their overlap halo, every rank generates only its own samples (seekable
    # device generator), runs its pipeline, and the per-snapshot angles meet in one RCCL all_gather.
    sharded = None
    if dist is not None:
        torch.cuda.synchronize()
        try:
            sharded = sharded_check(doa, torch, dist, world, local_rank)
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
