"""One-off fuzz of the fused pipeline over array sizes, source counts and spectrum lengths (every scan kernel family: lean,
lean multi-peak, general, long-spectrum LDS-row, two-pass): the peak ports must be the reference's find_local_max on the
pipeline's OWN spectrum bit for bit, the spectrum must sit on the fp64 evaluation of the reference's formulas, and the
stand-alone blocks chained by hand must give the pipeline's bits.
usage: python tests/fuzz_pipeline_shapes.py [n_cases] [seed]
(lives under tests/ because it checks against the oracle, which only test code may import)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle")]
import numpy as np
import doa
import doa_oracle as oracle


# shapes every run starts with (N, M, P, K, overlap, avg_method): one of each scan-kernel family -- lean (256 / 512 / 1024, one
# peak and several), general register-resident (not a multiple of 256), long-spectrum LDS-row (multiples of 64 up to 4096), the
# two-pass stream kernel (2048 < P <= 4096 with P % 64 != 0), the generic one (P % 4 != 0) -- with overlap, forward-backward
# averaging and every eigen-stage path (one-lane, four-lane, wave subspace iteration at G = 8 / 16, Jacobi for 2M > N)
FORCED = [(4, 1, 1024, 64, 0, 0), (4, 2, 1024, 64, 16, 1), (3, 2, 512, 64, 0, 1), (2, 1, 256, 16, 4, 0), (4, 3, 256, 64, 0, 0),
          (4, 2, 1000, 64, 0, 0), (5, 2, 768, 64, 16, 1), (8, 2, 2048, 64, 0, 0), (8, 3, 1536, 256, 64, 1), (6, 4, 1024, 64, 0, 0),
          (16, 3, 4096, 64, 0, 0), (12, 4, 2112, 64, 16, 1), (16, 1, 3008, 16, 0, 0), (9, 3, 2560, 64, 0, 1),
          (16, 3, 2500, 64, 0, 0), (8, 2, 4000, 64, 16, 1), (4, 2, 3100, 64, 0, 0), (13, 4, 2052, 16, 4, 1),
          (4, 1, 1023, 64, 0, 0), (7, 2, 2501, 64, 0, 1), (4, 2, 20, 64, 0, 0), (16, 4, 64, 64, 0, 0)]


def run(n_cases=60, seed=0, verbose=True, forced=()):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        N = int(rng.integers(2, 17))
        M = int(rng.integers(1, min(N, 5)))
        P = int(rng.choice([256, 512, 1024, 1000, 768, 2048, 1536, 2112, 3008, 4096, 2500, 4000, 64 * int(rng.integers(33, 65))]))
        K = int(rng.choice([16, 64, 256]))
        ovl = int(rng.choice([0, 0, K // 4]))
        fb = int(rng.integers(0, 2))
        n = int(rng.integers(1, 40))
        if case < len(forced):
            N, M, P, K, ovl, fb = forced[case]
        d = float(rng.choice([0.5, 0.4, 0.44]))
        th = np.sort(rng.uniform(25.0, 155.0, M)) + 4.0 * np.arange(M)
        span = (n - 1) * (K - ovl) + K
        x = doa.sim.make_streams(N, span, list(th), d, snr_db=float(rng.choice([5.0, 20.0])), seed=int(rng.integers(1 << 30)))
        pipe = doa.music_pipeline(N, K, ovl, fb, d, M, P, max_batch=n)
        mx, am = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
        cov, spec = np.empty((n, N * N), np.complex64), np.empty((n, P), np.float32)
        pipe.work(n, [x[k] for k in range(N)], mx, am, cov_out=cov, spectrum_out=spec)
        o0, o1 = oracle.find_local_max(spec, M, P, 0.0, 180.0)
        s64 = oracle.music_lin_array(cov, d, M, N, P, "f64")
        # the separate blocks on the same covariance items
        blk, pk = doa.MUSIC_lin_array(d, M, N, P), doa.find_local_max(M, P, 0.0, 180.0)
        spec2 = np.empty((n, P), np.float32)
        blk.work(n, [cov], [spec2])
        v0, v1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
        pk.work(n, [spec2], [v0, v1])
        q0, q1 = oracle.find_local_max(spec2, M, P, 0.0, 180.0)
        # Root-MUSIC on the same covariance items against the fp64 oracle (angles to 1e-3 degrees; rows where the oracle itself
        # reports no angle -- NaN -- must be NaN here too)
        root = doa.rootMUSIC_linear_array(d, M, N)
        ang = np.empty((n, M), np.float32)
        try:
            root.work(n, [cov], [ang])
        except doa.DoaError:
            pass                                                        # DOA_ERR_NUMERIC rows: the others are still written
        a64 = oracle.root_music(cov, d, M, N, "f64")
        both = np.isfinite(a64) & np.isfinite(ang)
        root_ok = np.array_equal(np.isfinite(a64), np.isfinite(ang)) and (not both.any() or np.abs(ang[both] - a64[both]).max() <= 1e-3)
        # ... and the same chain as one handle (doa.root_pipeline): the block's bits, from the streams
        rp = doa.root_pipeline(N, K, ovl, fb, d, M, max_batch=n)
        ang2, cov2 = np.full((n, M), np.nan, np.float32), np.empty((n, N * N), np.complex64)
        try:
            rp.work(n, [x[k] for k in range(N)], ang2, cov_out=cov2)
        except doa.DoaError:
            pass
        root_ok = root_ok and np.array_equal(ang2, ang, equal_nan=True) and np.array_equal(cov2.view(np.float32), cov.view(np.float32))
        ok = (root_ok and np.array_equal(mx, o0) and np.array_equal(am, o1) and np.array_equal(v0, q0) and np.array_equal(v1, q1)
              and np.abs(spec - s64).max() <= 2e-4 and np.abs(spec2 - s64).max() <= 2e-4
              and np.all(spec.max(axis=1) == 0.0) and np.all(spec2.max(axis=1) == 0.0))
        if not ok:
            bad += 1
            if verbose:
                print(f"case {case}: MISMATCH N={N} M={M} P={P} K={K} ovl={ovl} fb={fb} n={n}: peaks {np.array_equal(mx, o0)}/{np.array_equal(am, o1)} "
                      f"root {root_ok} blocks {np.array_equal(v0, q0)}/{np.array_equal(v1, q1)} spec err {np.abs(spec - s64).max():.2e} / {np.abs(spec2 - s64).max():.2e}")
    if verbose:
        print(f"{n_cases} cases, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
