"""GPU micro-benchmark of the individual stages (not part of the graded bench): HIP-event timing of
back-to-back launches over rotating buffers, for A/B-ing kernel variants in one process."""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch
import doa

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--nbuf", type=int, default=4)
ap.add_argument("--precision", type=int, default=64)
ap.add_argument("--stages", default="cov,music,peak,pipe")
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--K", type=int, default=1024)
ap.add_argument("--N", type=int, default=4)
ap.add_argument("--P", type=int, default=1024)
ap.add_argument("--M", type=int, default=1)
ap.add_argument("--ovl", type=int, default=0, help="overlap_size (windows advance by K-ovl samples)")
ap.add_argument("--fb", type=int, default=0, help="avg_method (1 = forward-backward)")
ap.add_argument("--snr", type=float, default=20.0, help="per-source SNR in dB of the random-direction data")
ap.add_argument("--ablate", default="", help="mcov: multi-stream covariance only; mmusic: multi-stream MUSIC only")
args = ap.parse_args()
N, K, P, M, B = args.N, args.K, args.P, args.M, args.batch
OVL, FB = args.ovl, args.fb
STEP = K - OVL          # new samples per snapshot; a batch of B windows spans (B-1)*STEP + K samples
doa.set_internal_precision(args.precision)
st = torch.cuda.current_stream()
streams = []
for b in range(args.nbuf):
    if os.environ.get("BENCH_NOISE_ONLY"):       # pure noise: the worst case for the iterative kernels
        x = torch.randn((N, B * K, 2), device="cuda", dtype=torch.float32)
        streams.append([torch.view_as_complex(x[n].contiguous()) for n in range(N)])
    elif OVL > 0:
        # overlapping windows: ONE continuous stream with fixed directions, the simulation flowgraph's own generator (as bench.py's
        # flowgraph_shape leg).  Until round 4 this script cut overlapping windows out of per-snapshot random-direction blocks: a
        # window then straddled two direction sets -- twice the sources the estimator is told about, a covariance without a noise
        # subspace -- which says nothing about the kernels on array data
        span = (B - 1) * STEP + K
        s_ = doa.sim.stream_slab_torch([torch.empty(span, dtype=torch.complex64, device="cuda") for _ in range(N)])
        th = [30.0, 123.0, 75.0, 150.0][:M]
        doa.sim_source(N, 0.5, th, [0.03125, 0.0625, 0.11, 0.2][:M], None, None, 0.1, seed=600 + b).work_dev(span, [t.data_ptr() for t in s_], st)
        streams.append(s_)
    else:                                        # M sources at SNR 20 dB, a random direction set per snapshot
        s_, _ = doa.sim.make_batch_streams_torch(N, K, B, 0.5, M, args.snr, seed=b)     # B*K >= (B-1)*STEP + K samples
        streams.append(s_)
PAD = int(os.environ.get("BENCH_CH_PAD", "-1"))        # >= 0: the N streams of a buffer set in one slab, PAD bytes between them
if PAD >= 0:
    assert PAD % 16 == 0
    padded = []
    for s_ in streams:
        nbytes = s_[0].numel() * 8
        slab = torch.empty(N * (nbytes + PAD) + 4096, dtype=torch.uint8, device="cuda")
        off0 = (-slab.data_ptr()) % 4096
        chans = []
        for n in range(N):
            v = slab[off0 + n * (nbytes + PAD): off0 + n * (nbytes + PAD) + nbytes].view(torch.complex64)
            v.copy_(s_[n])
            chans.append(v)
        padded.append(chans)
    streams = padded
ptrs = [[t.data_ptr() for t in s] for s in streams]
cov = [torch.empty((B, N * N), dtype=torch.complex64, device="cuda") for _ in range(args.nbuf)]
spec = [torch.empty((B, P), dtype=torch.float32, device="cuda") for _ in range(args.nbuf)]
mx = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(args.nbuf)]
am = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(args.nbuf)]
cov_blk = doa.autocorrelate(N, K, OVL, FB)
music_blk = doa.MUSIC_lin_array(0.5, M, N, P)
peak_blk = doa.find_local_max(M, P, 0.0, 180.0)
pipe = doa.music_pipeline(N, K, OVL, FB, 0.5, M, P, B)

def timeit(fn):
    for i in range(10): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for r in range(5):
        torch.cuda.synchronize(); e0.record(st)
        for i in range(args.reps): fn(i)
        e1.record(st); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / args.reps)
    return min(best), sorted(best)[len(best) // 2]

nb = args.nbuf
res = {}
for i in range(nb):   # valid covariances in every buffer for the music stage
    cov_blk.work_dev(B, ptrs[i], cov[i].data_ptr(), st)
if "cov" in args.stages:
    res["cov_us"] = timeit(lambda i: cov_blk.work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), st))
    res["cov_GBs"] = (N * STEP * 8 + N * N * 8) * B / res["cov_us"][0] / 1e3      # new samples only: the halo is a cache re-read
if "music" in args.stages:
    res["music_us"] = timeit(lambda i: music_blk.work_dev(B, cov[i % nb].data_ptr(), spec[i % nb].data_ptr(), st))
if "peak" in args.stages:
    res["peak_us"] = timeit(lambda i: peak_blk.work_dev(B, spec[i % nb].data_ptr(), mx[i % nb].data_ptr(), am[i % nb].data_ptr(), st))
if "root" in args.stages:
    root_blk = doa.rootMUSIC_linear_array(0.5, M, N)
    ang = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nb)]
    res["root_us"] = timeit(lambda i: root_blk.work_dev(B, cov[i % nb].data_ptr(), ang[i % nb].data_ptr(), st))
if "rootpipe" in args.stages:
    # configs[2] through ONE handle (doa_root_pipeline): serial steps on one stream
    rp = doa.root_pipeline(N, K, OVL, FB, 0.5, M, B)
    ang_rp = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nb)]
    res["rootpipe_us"] = timeit(lambda i: rp.work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), ang_rp[i % nb].data_ptr(), None, st))
skip_later = os.environ.pop("DOA_PIPE_SKIP", None)      # "cov", "evd", "scan" (comma separated): stages to drop AFTER every intermediate holds real data
def apply_skip(p, on):
    sk = skip_later or ""
    p.set_stages(cov=not (on and "cov" in sk), evd=not (on and "evd" in sk), scan=not (on and "scan" in sk))
if "pipe" in args.stages:
    for i in range(nb):
        pipe.work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), spec[i % nb].data_ptr(), mx[i % nb].data_ptr(), am[i % nb].data_ptr(), st)
    torch.cuda.synchronize()
    apply_skip(pipe, True)
    res["pipe_us"] = timeit(lambda i: pipe.work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), spec[i % nb].data_ptr(),
                                                    mx[i % nb].data_ptr(), am[i % nb].data_ptr(), st))
    res["snapshots_per_s"] = B / res["pipe_us"][0] * 1e6
if "mpipe" in args.stages:
    # alternate steps over several streams, one pipeline handle (= workspace) per stream
    S = args.streams
    sts = [torch.cuda.Stream() for _ in range(S)]
    pipes = [doa.music_pipeline(N, K, OVL, FB, 0.5, M, P, B) for _ in range(S)]
    def run(n):
        for i in range(n):
            k = i % S
            pipes[k].work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), spec[i % nb].data_ptr(), mx[i % nb].data_ptr(),
                              am[i % nb].data_ptr(), sts[k])
    import time
    run(2 * S * nb); torch.cuda.synchronize()          # real data in every workspace
    for p_ in pipes: apply_skip(p_, True)
    run(20); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(args.reps); th = time.perf_counter(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / args.reps * 1e6)
        res["mpipe_host_enqueue_us"] = (th - t0) / args.reps * 1e6
    res["mpipe_streams"] = S
    res["mpipe_us"] = (min(ts), sorted(ts)[2])
    res["mpipe_snapshots_per_s"] = B / min(ts) * 1e6
if "mroot" in args.stages:
    # configs[2]: autocorrelate -> rootMUSIC_linear_array, steps alternating over several streams
    import time
    S = args.streams
    sts = [torch.cuda.Stream() for _ in range(S)]
    covs = [doa.autocorrelate(N, K, OVL, FB) for _ in range(S)]
    roots = [doa.rootMUSIC_linear_array(0.5, M, N) for _ in range(S)]
    angs = [torch.empty((B, M), dtype=torch.float32, device="cuda") for _ in range(nb)]
    def run_root(n):
        for i in range(n):
            k = i % S
            covs[k].work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), sts[k])
            roots[k].work_dev(B, cov[i % nb].data_ptr(), angs[i % nb].data_ptr(), sts[k])
    run_root(2 * S * nb); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run_root(args.reps); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / args.reps * 1e6)
    res["mroot_streams"] = S
    res["mroot_us"] = (min(ts), sorted(ts)[2])
    res["mroot_snapshots_per_s"] = B / min(ts) * 1e6
if "host" in args.stages:
    # the host-pointer entry point (PCIe-inclusive): pageable vs page-locked caller buffers,
    # angles only vs spectrum returned as well; also the GNU Radio-sized call (8 items)
    import time
    import numpy as np
    for pinned in (False, True):
        hx = [torch.view_as_complex(torch.randn((B * K, 2), dtype=torch.float32)) for _ in range(N)]
        h_mx, h_am = torch.empty((B, M)), torch.empty((B, M))
        h_spec = torch.empty((B, P))
        if pinned:
            hx = [t.pin_memory() for t in hx]; h_mx, h_am, h_spec = h_mx.pin_memory(), h_am.pin_memory(), h_spec.pin_memory()
        xs = [t.numpy() for t in hx]
        for what, sp in (("angles", None), ("spectrum", h_spec.numpy())):
            for nit in (B, 1024, 256, 64, 8):
                pipe.work(nit, xs, h_mx.numpy(), h_am.numpy(), spectrum_out=sp)
                reps = 5 if nit == B else (20 if nit >= 1024 else 200)
                t0 = time.perf_counter()
                for _ in range(reps):
                    pipe.work(nit, xs, h_mx.numpy(), h_am.numpy(), spectrum_out=sp)
                dt = (time.perf_counter() - t0) / reps
                res[f"host_{'pinned' if pinned else 'pageable'}_{what}_n{nit}"] = {"us_per_call": dt * 1e6, "snapshots_per_s": nit / dt,
                                                                              "GBs_in": nit * N * K * 8 / dt / 1e9}
if args.ablate:
    import time
    S = args.streams
    sts = [torch.cuda.Stream() for _ in range(S)]
    covs = [doa.autocorrelate(N, K, 0, 0) for _ in range(S)]
    mus = [doa.music_pipeline(N, K, 0, 0, 0.5, M, P, B) for _ in range(S)]
    mbl = [doa.MUSIC_lin_array(0.5, M, N, P) for _ in range(S)]
    def run(n):
        for i in range(n):
            k = i % S
            if "mcov" in args.ablate:
                covs[k].work_dev(B, ptrs[i % nb], cov[i % nb].data_ptr(), sts[k])
            if "mmusic" in args.ablate:
                mbl[k].work_dev(B, cov[i % nb].data_ptr(), spec[i % nb].data_ptr(), sts[k])
    run(20); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(args.reps); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / args.reps * 1e6)
    res["ablate"] = args.ablate; res["ablate_streams"] = S; res["ablate_us"] = (min(ts), sorted(ts)[2])
print(json.dumps({"env": {**{k: v for k, v in os.environ.items() if k.startswith("DOA_")}, **({"DOA_PIPE_SKIP": skip_later} if skip_later else {})}, "batch": B, **res}))
