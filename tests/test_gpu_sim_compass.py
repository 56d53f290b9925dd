"""GPU tests of the two neighbours of the hot path: the device-side signal
front end of the simulation flowgraphs (doa.sim_source) and the vector_to_streams + compass averaging
(doa.compass_mean), each against its numpy restatement, then the whole simulation
(apps/run_MUSIC_lin_array_simulation.py) device-resident from generator to compass."""
import numpy as np
import pytest

import doa
import doa_oracle as oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FLOWGRAPH = dict(N=4, d=0.4, thetas=[30.0, 123.0], freqs=[10e3 / 320e3, 20e3 / 320e3], src_noise=[5e-5, 5e-3])


@pytest.mark.parametrize("N,d,thetas,freqs,ampl,src_noise,sigma,seed", [
    (4, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, [5e-5, 5e-3], 0.0, 1),           # the flowgraph
    (4, 0.5, [57.3], [0.0419], None, None, 0.1, 0xDEADBEEFCAFE),                       # benchmark SNR model
    (5, 0.45, [20.0, 90.0, 160.0], [0.011, 0.2503, 0.4999], [1.0, 0.5, 2.0], [0.1, 0.0, 0.3], 0.25, 77),
    (16, 0.5, [40.0], [0.3], [0.7], None, 0.0, 3),                                     # noise-free
])
def test_sim_source_matches_oracle(N, d, thetas, freqs, ampl, src_noise, sigma, seed):
    n = 4099                                     # odd: the last sample of a run is a half pair
    src = doa.sim_source(N, d, thetas, freqs, ampl, src_noise, sigma, seed)
    out = [np.empty(n, np.complex64) for _ in range(N)]
    assert src.work(n, out) == n
    ref = oracle.sim_source(N, d, thetas, freqs, n, ampl, src_noise, sigma, seed)
    scale = max(1.0, float(np.abs(ref).max()))
    # float32 log/sqrt/sincos on the device against float64 in the oracle; the Philox integers are exact
    assert np.abs(np.stack(out) - ref).max() <= 2e-5 * scale


def test_sim_source_is_independent_of_call_boundaries_and_seekable():
    N, n = 4, 6000
    mk = lambda: doa.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], None, [5e-5, 5e-3], 0.05, 9)
    whole = [np.empty(n, np.complex64) for _ in range(N)]
    mk().work(n, whole)
    s = mk()
    parts = []
    for m in (2, 1000, 998, 4000):               # even-sized calls, then the rest
        o = [np.empty(m, np.complex64) for _ in range(N)]
        assert s.work(m, o) == m
        parts.append(np.stack(o))
    assert s.tell() == n
    assert np.array_equal(np.concatenate(parts, axis=1), np.stack(whole))
    s.seek(1000)
    o = [np.empty(500, np.complex64) for _ in range(N)]
    s.work(500, o)
    assert np.array_equal(np.stack(o), np.stack(whole)[:, 1000:1500])
    ref = oracle.sim_source(N, 0.4, [30.0, 123.0], [0.03125, 0.0625], 500, None, [5e-5, 5e-3], 0.05, 9, first_sample=1000)
    assert np.abs(np.stack(o) - ref).max() <= 2e-5 * 2.0
    # positions stay even: an odd-sized call must be followed by a seek
    s.work(3, [np.empty(3, np.complex64) for _ in range(N)])
    with pytest.raises(doa.DoaError):
        s.work(2, [np.empty(2, np.complex64) for _ in range(N)])
    with pytest.raises(doa.DoaError):
        s.seek(7)
    s.seek(8)
    assert s.work(2, [np.empty(2, np.complex64) for _ in range(N)]) == 2


def test_sim_source_large_positions():
    # sample indices beyond 2^32 (the Philox counter's high word, the double phase of the tone)
    N, first, n = 2, (1 << 33) + 4096, 256
    s = doa.sim_source(N, 0.5, [70.0], [0.123456789], None, [0.2], 0.1, 5)
    s.seek(first)
    o = [np.empty(n, np.complex64) for _ in range(N)]
    s.work(n, o)
    ref = oracle.sim_source(N, 0.5, [70.0], [0.123456789], n, None, [0.2], 0.1, 5, first_sample=first)
    assert np.abs(np.stack(o) - ref).max() <= 5e-5


def test_sim_source_rejects_bad_arguments():
    with pytest.raises(doa.DoaError):
        doa.sim_source(0, 0.5, [10.0], [0.1])
    with pytest.raises(doa.DoaError):
        doa.sim_source(17, 0.5, [10.0], [0.1])
    with pytest.raises(doa.DoaError):
        doa.sim_source(4, 0.0, [10.0], [0.1])
    with pytest.raises(ValueError):
        doa.sim_source(4, 0.5, [10.0, 20.0], [0.1])
    with pytest.raises(doa.DoaError):
        doa.sim_source(4, 0.5, [10.0], [0.1], antenna_noise_sigma=-1.0)


@pytest.mark.parametrize("n,M", [(1, 1), (7, 2), (256, 3), (4096, 1), (5000, 16)])
def test_compass_mean_matches_numpy_mean(n, M):
    rng = np.random.default_rng(n + M)
    a = rng.uniform(0.0, 180.0, size=(n, M)).astype(np.float32)
    blk = doa.compass_mean(M)
    assert blk.work(n, [a]) == n                                   # "consume all inputs"
    ref = oracle.compass_mean(a, M)
    exact = a.astype(np.float64).mean(axis=0)
    assert np.abs(blk.next_angle - exact).max() <= 8e-6            # correctly rounded mean (ulp(180) = 1.5e-5)
    assert np.abs(blk.next_angle - ref).max() <= 1e-4              # numpy's float32 pairwise sum is a few ulp off that


def test_compass_mean_empty_and_bad_arguments():
    blk = doa.compass_mean(2)
    assert blk.work(0, [np.empty((0, 2), np.float32)]) == 0
    assert np.isnan(blk.next_angle).all()                           # numpy.mean([]) is nan
    with pytest.raises(doa.DoaError):
        doa.compass_mean(0)
    with pytest.raises(doa.DoaError):
        doa.compass_mean(17)


def test_simulation_flowgraph_device_resident_end_to_end():
    """apps/run_MUSIC_lin_array_simulation.py headless: generator -> autocorrelate(4, 2048, 512, FB) ->
    MUSIC_lin_array(0.4, 2, 4, 1024) -> find_local_max(2, 1024, 0, 180) -> vector_to_streams -> compass x2,
    nothing but the two averaged angles leaving the device."""
    f = FLOWGRAPH
    N, K, ovl, M, P, n = f["N"], 2048, 512, 2, 1024, 64
    T = (n - 1) * (K - ovl) + K
    gen = doa.sim_source(N, f["d"], f["thetas"], f["freqs"], None, f["src_noise"], 0.0, seed=2024)
    streams = [torch.empty(T, dtype=torch.complex64, device="cuda") for _ in range(N)]
    st = torch.cuda.current_stream()
    assert gen.work_dev(T, [s.data_ptr() for s in streams], st) == T
    pipe = doa.music_pipeline(N, K, ovl, 1, f["d"], M, P, max_batch=n)
    mx = torch.empty((n, M), dtype=torch.float32, device="cuda")
    am = torch.empty((n, M), dtype=torch.float32, device="cuda")
    pipe.work_dev(n, [s.data_ptr() for s in streams], 0, 0, mx.data_ptr(), am.data_ptr(), st)
    comp = doa.compass_mean(M)
    ang = torch.empty(M, dtype=torch.float32, device="cuda")
    assert comp.work_dev(n, am.data_ptr(), ang.data_ptr(), st) == n
    torch.cuda.synchronize()
    got = ang.cpu().numpy()
    # find_local_max's location port is sorted descending (reference find_local_max_impl.cc:187-190)
    assert np.abs(got - np.array([123.0, 30.0])).max() <= 1.0
    # the same chain through the oracle on the generated samples
    x = np.stack([s.cpu().numpy() for s in streams])
    _, _, _, loc = oracle.music_pipeline(x, K, ovl, 1, f["d"], M, P, n)
    assert np.abs(got - oracle.compass_mean(loc, M)).max() <= 180.0 / P + 1e-3
    # and the generated streams are the oracle's
    ref = oracle.sim_source(N, f["d"], f["thetas"], f["freqs"], T, None, f["src_noise"], 0.0, 2024)
    assert np.abs(x - ref).max() <= 4e-5
