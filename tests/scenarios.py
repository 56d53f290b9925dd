"""Seeded scenarios shared by the parity tests and the golden-fixture generator.

Each scenario restates a configuration the reference itself exercises (its QA tests, its
simulation flowgraphs, BASELINE.json's configs) on deterministic inputs."""
import numpy as np

from doa import sim

# name -> dict(N, M, d, thetas, K, ovl, fb, P, snr_db, per_source_noise, n)
SCENARIOS = {
    # python/qa_MUSIC_lin_array.py:46-62 (music_test_input_gen default SNR = 1000 dB -> noise-free)
    "qa_music_aoa23": dict(N=8, M=1, d=0.4, thetas=[23.0], K=256, ovl=32, fb=1, P=1024, snr_db=None, n=12),
    # python/qa_MUSIC_lin_array.py:102-118
    "qa_music_aoa121": dict(N=16, M=1, d=0.5, thetas=[121.0], K=256, ovl=32, fb=1, P=1024, snr_db=None, n=8),
    # python/qa_rootMUSIC_linear_array.py:41-57 / :94-110
    "qa_root_aoa23": dict(N=8, M=1, d=0.5, thetas=[23.0], K=256, ovl=32, fb=1, P=1024, snr_db=None, n=12),
    "qa_root_aoa52": dict(N=4, M=1, d=0.5, thetas=[52.0], K=1024, ovl=64, fb=1, P=1024, snr_db=None, n=12),
    # apps/run_MUSIC_lin_array_simulation.grc: 4 elements, 2 sources 30/123 deg, d=0.4, K=2048,
    # ovl=512, FB, noise added per source before the manifold (rank-2 covariance)
    "grc_music_sim": dict(N=4, M=2, d=0.4, thetas=[30.0, 123.0], K=2048, ovl=512, fb=1, P=1024, snr_db=None,
                          per_source_noise=[5e-5, 5e-3], n=8),
    # apps/run_RootMUSIC_lin_array_simulation.grc: d=0.44, noise amplitudes 5e-4 / 0.5
    "grc_root_sim": dict(N=4, M=2, d=0.44, thetas=[30.0, 123.0], K=2048, ovl=512, fb=1, P=1024, snr_db=None,
                         per_source_noise=[5e-4, 0.5], n=8),
    # BASELINE.json configs[1]: N=4, 1 source, K=1024, P=1024, per-antenna noise, SNR 20 dB
    "bench_cfg2": dict(N=4, M=1, d=0.5, thetas=[57.3], K=1024, ovl=0, fb=0, P=1024, snr_db=20.0, n=16),
    # BASELINE.json configs[2]: Root-MUSIC, N=4, 2 sources
    "bench_cfg3": dict(N=4, M=2, d=0.44, thetas=[30.0, 123.0], K=1024, ovl=0, fb=0, P=1024, snr_db=20.0, n=16),
    # BASELINE.json configs[3]: N=16, 3 sources, P=4096
    "bench_cfg4": dict(N=16, M=3, d=0.5, thetas=[40.0, 90.0, 121.0], K=1024, ovl=0, fb=0, P=4096, snr_db=10.0, n=4),
    # well-conditioned / awkward shapes
    "low_snr": dict(N=4, M=1, d=0.5, thetas=[101.7], K=1024, ovl=0, fb=0, P=1024, snr_db=0.0, n=12),
    "two_ant": dict(N=2, M=1, d=0.5, thetas=[70.0], K=512, ovl=0, fb=0, P=256, snr_db=15.0, n=8),
    "three_ant_fb": dict(N=3, M=2, d=0.45, thetas=[50.0, 110.0], K=600, ovl=100, fb=1, P=512, snr_db=25.0, n=8),
    "five_ant": dict(N=5, M=2, d=0.5, thetas=[35.0, 140.0], K=400, ovl=0, fb=0, P=1000, snr_db=12.0, n=8),
    "xml_default_p20": dict(N=4, M=1, d=0.5, thetas=[80.0], K=2048, ovl=512, fb=0, P=20, snr_db=20.0, n=5),
    "twelve_ant": dict(N=12, M=4, d=0.5, thetas=[25.0, 60.0, 100.0, 150.0], K=512, ovl=0, fb=1, P=2048, snr_db=15.0, n=4),
}


def make_input(name):
    """Returns (cfg, x) with x = [N, (n-1)*(K-ovl)+K] complex64 including history samples."""
    c = SCENARIOS[name]
    S = c["K"] - c["ovl"]
    T = (c["n"] - 1) * S + c["K"]
    seed = sum(ord(ch) for ch in name)
    x = sim.make_streams(c["N"], T, c["thetas"], c["d"], snr_db=c["snr_db"], seed=seed,
                         per_source_noise=c.get("per_source_noise"))
    return c, x


def is_rank_deficient(c):
    """Noise-free (or noise added before the manifold): the noise eigenvalues are rounding noise and
    Q at a source direction is cancellation-dominated."""
    return c["snr_db"] is None
