"""GPU test of the lane streams of the pipeline handles (gr-doa_amd/csrc/lane_streams.hip; the reference has no counterpart:
GNU Radio gets its overlap from one thread per block).  The HIP runtime may put two streams on one hardware queue, where
their kernels run one after the other; the handles therefore probe the streams they create and keep only those seen to run
side by side.  Checked here in the situation that exposed it (profiles/r04_lab_lane_queue_sharing.txt): other handles that
own streams are alive while the lanes are created -- and the results of such a handle equal the single-lane ones, bit for bit."""
import numpy as np
import pytest
import torch

import doa
from doa._lib import lib

pytestmark = pytest.mark.gpu


def _flow_inputs(n_items, seed):
    N, K, ovl = 4, 2048, 512
    span = (n_items - 1) * (K - ovl) + K
    x = doa.sim.make_streams(N, span, [30.0, 123.0], 0.4, snr_db=20.0, seed=seed)
    return [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in x]


@pytest.mark.parametrize("n_foreign", [0, 1, 2, 5])
def test_lanes_run_side_by_side_whatever_streams_are_alive(n_foreign):
    n, steps = 256, 8
    st = torch.cuda.current_stream()
    # foreign streams created in the overlapping pattern that left two lanes on one queue: each generator handle owns a stream
    gen = lambda k: doa.sim_source(4, 0.4, [30.0], [0.03125], None, None, 0.1, seed=k)
    prev = None
    for k in range(3):
        prev = gen(k)                                # the previous handle is released after this one exists
    foreign = [prev][:n_foreign] + [gen(10 + k) for k in range(max(0, n_foreign - 1))]
    del prev
    assert len(foreign) == n_foreign
    ins = [_flow_inputs(n, seed=10 + b) for b in range(2)]
    out = {}
    for lanes in (1, 4):
        p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, n)
        p.set_lanes(lanes)
        cov = [torch.empty((n, 16), dtype=torch.complex64, device="cuda") for _ in range(steps)]
        spec = [torch.empty((n, 1024), dtype=torch.float32, device="cuda") for _ in range(steps)]
        mx = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        am = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        p.work_dev_batches(n, [[t.data_ptr() for t in ins[b % 2]] for b in range(steps)], [t.data_ptr() for t in cov],
                           [t.data_ptr() for t in spec], [t.data_ptr() for t in mx], [t.data_ptr() for t in am], st)
        torch.cuda.synchronize()
        if lanes == 4:
            assert lib.doa_hip_lane_streams_verified_debug() == 4, "the four lanes were not all seen to run side by side"
            assert 0 <= lib.doa_hip_lane_streams_set_aside_debug() <= 18
        out[lanes] = [torch.stack(v).cpu().numpy() for v in (cov, spec, mx, am)]
    for a, b in zip(out[1], out[4]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    del foreign


def test_more_lanes_than_hardware_queues_still_work():
    """eight lanes cannot all be on queues of their own (the runtime has four): the probe gives up after its tries and the
    handle works as before, results unchanged"""
    n, steps = 128, 8
    st = torch.cuda.current_stream()
    ins = _flow_inputs(n, seed=3)
    res = []
    for lanes in (1, 8):
        p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, n)
        p.set_lanes(lanes)
        am = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        spec = [torch.empty((n, 1024), dtype=torch.float32, device="cuda") for _ in range(steps)]
        mx = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        cov = [torch.empty((n, 16), dtype=torch.complex64, device="cuda") for _ in range(steps)]
        p.work_dev_batches(n, [[t.data_ptr() for t in ins]] * steps, [t.data_ptr() for t in cov], [t.data_ptr() for t in spec],
                           [t.data_ptr() for t in mx], [t.data_ptr() for t in am], st)
        torch.cuda.synchronize()
        if lanes == 8:
            assert 4 <= lib.doa_hip_lane_streams_verified_debug() <= 8
        res.append(torch.stack(spec).cpu().numpy())
    assert np.array_equal(res[0].view(np.uint8), res[1].view(np.uint8))


def test_handles_created_and_probed_from_several_threads_at_once():
    """GNU Radio runs one thread per block: several fused handles may create and probe their lanes at the same time.  The probes
    then disturb each other's timing (a probe kernel delayed by a neighbour's reads as 'not side by side' and costs a retry);
    whatever they decide, every handle works and gives the single-lane results."""
    import threading
    n, steps, n_threads = 128, 8, 4
    ins = [_flow_inputs(n, seed=60 + t) for t in range(n_threads)]
    want = []
    for t in range(n_threads):                       # single-lane references, one after the other
        p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, n)
        p.set_lanes(1)
        spec = [torch.empty((n, 1024), dtype=torch.float32, device="cuda") for _ in range(steps)]
        mx = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        am = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
        p.work_dev_batches(n, [[x.data_ptr() for x in ins[t]]] * steps, None, [x.data_ptr() for x in spec], [x.data_ptr() for x in mx],
                           [x.data_ptr() for x in am], torch.cuda.current_stream())
        torch.cuda.synchronize()
        want.append(torch.stack(spec).cpu().numpy())
    got, err = [None] * n_threads, []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            p = doa.music_pipeline(4, 2048, 512, 1, 0.4, 2, 1024, n)
            p.set_lanes(4)
            spec = [torch.empty((n, 1024), dtype=torch.float32, device="cuda") for _ in range(steps)]
            mx = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
            am = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(steps)]
            call = p.prepare_batches(n, [[x.data_ptr() for x in ins[t]]] * steps, None, [x.data_ptr() for x in spec],
                                     [x.data_ptr() for x in mx], [x.data_ptr() for x in am], doa.DETACHED)
            start.wait()
            call()                                   # creates and probes this handle's four lanes
            p.synchronize()
            call()
            p.synchronize()
            got[t] = torch.stack(spec).cpu().numpy()
        except Exception as e:                       # pragma: no cover - reported by the main thread
            err.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not err, err
    for t in range(n_threads):
        assert np.array_equal(got[t].view(np.uint8), want[t].view(np.uint8))
    assert 1 <= lib.doa_hip_lane_streams_verified_debug() <= 4
