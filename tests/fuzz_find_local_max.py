"""One-off fuzz of the two find_local_max kernels (wave kernel and one-thread-per-vector kernel) against
the oracle, bit for bit: random lengths (multiples of 4 and not, up to 5000), num_max_vals 1..16, values
drawn from smooth / quantised (flat-heavy) / constant / spiky families with NaN and +-inf sprinkled in.
usage: python tests/fuzz_find_local_max.py [n_cases] [seed]
(lives under tests/ because it checks against the oracle, which only test code may import)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python"), os.path.join(ROOT, "oracle")]
import numpy as np
import doa
import doa_oracle as oracle

def run(n_cases=300, seed=0, verbose=True):
    """Returns the number of cases in which a kernel and the oracle differ (NaNs compare equal)."""
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        L = int(rng.choice([rng.integers(3, 64), 4 * rng.integers(1, 1250), rng.integers(64, 5000), 256, 1024, 4096]))
        M = int(rng.integers(1, 17))
        n = int(rng.integers(1, 12))
        fam = rng.integers(0, 5)
        t = np.linspace(0, 1, L)[None, :]
        if fam == 0:
            v = rng.standard_normal((n, L))
        elif fam == 1:
            v = np.round(3 * np.sin(2 * np.pi * rng.uniform(1, 9, (n, 1)) * t + rng.uniform(0, 6, (n, 1))) + rng.standard_normal((n, L)) * 0.3)
        elif fam == 2:
            v = np.full((n, L), rng.standard_normal())
        elif fam == 3:
            v = -np.abs(rng.standard_normal((n, L))) * 40
            v[:, rng.integers(0, L, size=max(1, L // 50))] = 0.0
        else:
            v = np.round(rng.standard_normal((n, L)) * 2) / 2
        v = v.astype(np.float32)
        if rng.random() < 0.3:
            k = rng.integers(0, L, size=3)
            v[rng.integers(0, n), k] = rng.choice([np.nan, np.inf, -np.inf], size=3)
        x_min, x_max = (0.0, 180.0) if rng.random() < 0.7 else (float(rng.uniform(-10, 0)), float(rng.uniform(1, 400)))
        blk = doa.find_local_max(M, L, x_min, x_max)
        o0, o1 = np.empty((n, M), np.float32), np.empty((n, M), np.float32)
        blk.work(n, [v], [o0, o1])
        with np.errstate(all="ignore"):
            r0, r1 = oracle.find_local_max(v, M, L, x_min, x_max)
        if not (np.array_equal(o0, r0, equal_nan=True) and np.array_equal(o1, r1, equal_nan=True)):
            bad += 1
            if verbose:
                print("MISMATCH case", case, "L", L, "M", M, "family", int(fam))
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
