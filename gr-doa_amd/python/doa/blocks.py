"""Host-side mirror of the reference's `doa` Python namespace for the hot-path blocks.

The reference exposes its C++ blocks to Python through SWIG (`swig/doa_swig.i:22-34`) as
`doa.autocorrelate(inputs, snapshot_size, overlap_size, avg_method)`,
`doa.MUSIC_lin_array(norm_spacing, num_targets, inputs, pspectrum_len)`,
`doa.find_local_max(num_max_vals, vector_len, x_min, x_max)` and
`doa.rootMUSIC_linear_array(norm_spacing, num_targets, inputs)` — the make strings of the GRC
descriptors (`grc/doa_*.xml`).  The classes below keep those names and argument orders and forward
every call to the HIP library through its C ABI (include/doa_hip.h); they hold no arithmetic.

Each block offers
  * `general_work` / `work(noutput_items, input_items, output_items)` with the item layouts of the
    GNU Radio buffers (numpy arrays standing in for the scheduler's buffers), used by
    `doa.runtime`'s mini scheduler, and
  * `work_dev(...)` on raw device pointers (ints, e.g. `torch.Tensor.data_ptr()`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check, check_handle, ptr_array

_C64 = np.complex64
_F32 = np.float32


def _vp(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


DETACHED = "detached"      # work_dev_batches(..., stream=DETACHED): no ordering on any caller stream (DOA_STREAM_DETACHED)


def _stream_ptr(stream) -> C.c_void_p:
    """Accept None, an int (hipStream_t), a torch.cuda.Stream or DETACHED."""
    if stream is None:
        return C.c_void_p(0)
    if isinstance(stream, str) and stream == DETACHED:
        return C.c_void_p(2 ** 64 - 1)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(int(stream.cuda_stream))
    return C.c_void_p(int(stream))


class _Block:
    _destroy = None

    def __init__(self):
        self._h = None

    def close(self):
        if getattr(self, "_h", None):
            type(self)._destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    _set_precision = None

    def set_internal_precision(self, bits: int) -> None:
        """Per-handle internal precision of EVD / scan (32 or 64; include/doa_hip.h); blocks without an EVD have none."""
        if type(self)._set_precision is None:
            raise AttributeError(f"{type(self).__name__} has no internal precision")
        check(type(self)._set_precision(self._h, int(bits)))


class autocorrelate(_Block):
    """doa.autocorrelate(inputs, snapshot_size, overlap_size, avg_method) — gr::block with
    history overlap_size+1 (reference lib/autocorrelate_impl.cc:47-65)."""

    _destroy = staticmethod(lib.doa_autocorrelate_destroy)

    def __init__(self, inputs, snapshot_size, overlap_size, avg_method):
        super().__init__()
        self.inputs, self.snapshot_size = int(inputs), int(snapshot_size)
        self.overlap_size, self.avg_method = int(overlap_size), int(avg_method)
        self._h = check_handle(lib.doa_autocorrelate_create(self.inputs, self.snapshot_size,
                                                            self.overlap_size, self.avg_method),
                               "autocorrelate")
        # io signature of the reference block (:49-50)
        self.in_sig = [(_C64, 1)] * self.inputs
        self.out_sig = [(_C64, self.inputs * self.inputs)]

    def history(self) -> int:
        return check(lib.doa_autocorrelate_history(self._h))

    def forecast(self, noutput_items: int) -> int:
        return check(lib.doa_autocorrelate_forecast(self._h, int(noutput_items)))

    def input_span(self, noutput_items: int) -> int:
        return int(lib.doa_autocorrelate_input_span(self._h, int(noutput_items)))

    def general_work(self, noutput_items, input_items, output_items):
        """input_items[k]: complex64 array starting at the first history sample of stream k
        (at least input_span(noutput_items) long); output_items[0]: [>=n, N*N] complex64.
        Returns (items produced, items consumed per input) — the caller applies consume_each."""
        n = int(noutput_items)
        span = self.input_span(n)
        arrs = []
        for k in range(self.inputs):
            a = np.ascontiguousarray(input_items[k], dtype=_C64)
            if a.shape[0] < span:
                raise ValueError(f"input {k}: {a.shape[0]} samples, need {span}")
            arrs.append(a)
        out = output_items[0]
        assert out.dtype == _C64 and out.flags.c_contiguous and out.size >= n * self.inputs ** 2
        produced = check(lib.doa_autocorrelate_work(self._h, n, ptr_array([a.ctypes.data for a in arrs]),
                                                    _vp(out)))
        return produced, self.forecast(produced)

    def work_dev(self, noutput_items, d_input_ptrs, d_out_ptr, stream=None) -> int:
        return check(lib.doa_autocorrelate_work_dev(self._h, int(noutput_items), ptr_array(d_input_ptrs),
                                                    C.c_void_p(int(d_out_ptr)), _stream_ptr(stream)))

    def fuse_antenna_correction(self, correction) -> None:
        """Fold a doa.antenna_correction block (or an array of complex gains, or None to undo) into
        this block: equivalent to wiring the correction block in front of it."""
        _fuse(lib.doa_autocorrelate_fuse_antenna_correction, self._h, correction, self.inputs)


def _fuse(fn, handle, correction, n):
    if correction is None:
        check(fn(handle, C.c_void_p(0)))
        return
    g = correction.gains() if hasattr(correction, "gains") else np.asarray(correction)
    g = np.ascontiguousarray(np.asarray(g, dtype=_C64).reshape(n))
    check(fn(handle, _vp(g)))


class antenna_correction(_Block):
    """doa.antenna_correction(num_inputs, config_filename) — gr::sync_block, N complex streams in
    and out (reference lib/antenna_correction_impl.cc:47-99).  Raises ValueError with the
    reference's std::invalid_argument text for a missing / too long / too short config file."""

    _destroy = staticmethod(lib.doa_antenna_correction_destroy)

    def __init__(self, num_inputs, config_filename):
        super().__init__()
        self.num_ant_ele = int(num_inputs)
        h = lib.doa_antenna_correction_create(self.num_ant_ele, str(config_filename).encode())
        if not h:
            msg = _lib.last_error()
            if msg.startswith(("Cannot find configuration", "Configuration file")):
                raise ValueError(msg)                   # std::invalid_argument in the reference
            raise _lib.DoaError(-1, msg or "antenna_correction: create failed")
        self._h = h
        self.in_sig = [(_C64, 1)] * self.num_ant_ele
        self.out_sig = [(_C64, 1)] * self.num_ant_ele

    def gains(self) -> np.ndarray:
        g = np.empty(self.num_ant_ele, dtype=_C64)
        check(lib.doa_antenna_correction_gains(self._h, _vp(g)))
        return g

    def work(self, noutput_items, input_items, output_items) -> int:
        n = int(noutput_items)
        ins = [np.ascontiguousarray(a, dtype=_C64) for a in input_items]
        for a, o in zip(ins, output_items):
            assert a.size >= n and o.dtype == _C64 and o.flags.c_contiguous and o.size >= n
        return check(lib.doa_antenna_correction_work(self._h, n, ptr_array([a.ctypes.data for a in ins]),
                                                     ptr_array([o.ctypes.data for o in output_items])))

    def work_dev(self, noutput_items, d_in_ptrs, d_out_ptrs, stream=None) -> int:
        return check(lib.doa_antenna_correction_work_dev(self._h, int(noutput_items), ptr_array(d_in_ptrs),
                                                         ptr_array(d_out_ptrs), _stream_ptr(stream)))


def read_phase_config(filename):
    """The phase file of phase_correct_hier, parsed the way python/phase_correct_hier.py:33-45 parses it: every line
    that float() accepts AND whose value is truthy is a phase (so a line reading 0 is dropped, like a comment line)."""
    def ok(text):
        try:
            return float(text)
        except ValueError:
            return False
    with open(filename, "r") as f:
        lines = [line.rstrip("\n") for line in f]
    return [float(t) for t in lines if ok(t)]


class phase_correct_hier(antenna_correction):
    """doa.phase_correct_hier(num_ports, config_filename) -- the reference's hier block (python/phase_correct_hier.py:
    52-104): stream 0 passes through, stream p+1 is multiplied by exp(1j*phase_p), phases read from a text file.  Here one
    launch of the per-stream complex-gain kernel (antenna_correction's), or no launch at all when the gains are folded
    into the covariance kernel (autocorrelate.fuse_antenna_correction(self.gains())).  The reference writes its message to
    stderr and exits; this raises ValueError with the same text."""

    def __init__(self, num_ports=2, config_filename=""):
        _Block.__init__(self)
        self.num_ports = self.num_ant_ele = int(num_ports)
        self.config_filename = config_filename
        try:
            open(config_filename, "r").close()
        except (OSError, IOError):
            raise ValueError("Configuration " + str(config_filename) + ", not valid")
        self.phases = read_phase_config(config_filename)
        if len(self.phases) != self.num_ports - 1:
            raise ValueError("Configuration " + str(config_filename) + ". Not valid number of phase estimates")
        # numpy.exp(1j*phase) in double, handed to multiply_const_vcc as gr_complex (:93-94)
        g = np.array([1.0 + 0.0j] + [np.exp(1j * ph) for ph in self.phases], dtype=np.complex128).astype(_C64)
        h = lib.doa_antenna_correction_create_gains(self.num_ports, _vp(np.ascontiguousarray(g)))
        if not h:
            raise _lib.DoaError(-1, _lib.last_error() or "phase_correct_hier: create failed")
        self._h = h
        self.in_sig = [(_C64, 1)] * self.num_ports
        self.out_sig = [(_C64, 1)] * self.num_ports


class MUSIC_lin_array(_Block):
    """doa.MUSIC_lin_array(norm_spacing, num_targets, inputs, pspectrum_len) — gr::sync_block
    (reference lib/MUSIC_lin_array_impl.cc:47-87)."""

    _destroy = staticmethod(lib.doa_MUSIC_lin_array_destroy)
    _set_precision = staticmethod(lib.doa_MUSIC_lin_array_set_internal_precision)

    def __init__(self, norm_spacing, num_targets, inputs, pspectrum_len):
        super().__init__()
        self.norm_spacing, self.num_targets = float(norm_spacing), int(num_targets)
        self.num_ant_ele, self.pspectrum_len = int(inputs), int(pspectrum_len)
        self._h = check_handle(lib.doa_MUSIC_lin_array_create(self.norm_spacing, self.num_targets,
                                                              self.num_ant_ele, self.pspectrum_len),
                               "MUSIC_lin_array")
        self.in_sig = [(_C64, self.num_ant_ele ** 2)]
        self.out_sig = [(_F32, self.pspectrum_len)]

    def work(self, noutput_items, input_items, output_items) -> int:
        n = int(noutput_items)
        a = np.ascontiguousarray(input_items[0], dtype=_C64)
        out = output_items[0]
        assert a.size >= n * self.num_ant_ele ** 2
        assert out.dtype == _F32 and out.flags.c_contiguous and out.size >= n * self.pspectrum_len
        return check(lib.doa_MUSIC_lin_array_work(self._h, n, _vp(a), _vp(out)))

    def work_dev(self, noutput_items, d_in_ptr, d_out_ptr, stream=None) -> int:
        return check(lib.doa_MUSIC_lin_array_work_dev(self._h, int(noutput_items), C.c_void_p(int(d_in_ptr)),
                                                      C.c_void_p(int(d_out_ptr)), _stream_ptr(stream)))

    def debug(self, R_items: np.ndarray):
        """(P_N [n, N*N] complex64 column-major items, Q [n, P] float32) for parity tests."""
        a = np.ascontiguousarray(R_items, dtype=_C64).reshape(-1, self.num_ant_ele ** 2)
        n = a.shape[0]
        pn = np.empty((n, self.num_ant_ele ** 2), dtype=_C64)
        q = np.empty((n, self.pspectrum_len), dtype=_F32)
        check(lib.doa_MUSIC_lin_array_debug(self._h, n, _vp(a), _vp(pn), _vp(q)))
        return pn, q

    def nout_items_total(self) -> int:
        return int(lib.doa_MUSIC_lin_array_items_total(self._h))


class find_local_max(_Block):
    """doa.find_local_max(num_max_vals, vector_len, x_min, x_max) — gr::sync_block with two
    outputs (reference lib/find_local_max_impl.cc:47-71)."""

    _destroy = staticmethod(lib.doa_find_local_max_destroy)

    def __init__(self, num_max_vals, vector_len, x_min, x_max):
        super().__init__()
        self.num_max_vals, self.vector_len = int(num_max_vals), int(vector_len)
        self.x_min, self.x_max = float(x_min), float(x_max)
        self._h = check_handle(lib.doa_find_local_max_create(self.num_max_vals, self.vector_len,
                                                             self.x_min, self.x_max), "find_local_max")
        self.in_sig = [(_F32, self.vector_len)]
        self.out_sig = [(_F32, self.num_max_vals), (_F32, self.num_max_vals)]

    def work(self, noutput_items, input_items, output_items) -> int:
        n = int(noutput_items)
        a = np.ascontiguousarray(input_items[0], dtype=_F32)
        o0, o1 = output_items
        assert a.size >= n * self.vector_len
        for o in (o0, o1):
            assert o.dtype == _F32 and o.flags.c_contiguous and o.size >= n * self.num_max_vals
        return check(lib.doa_find_local_max_work(self._h, n, _vp(a), _vp(o0), _vp(o1)))

    def work_dev(self, noutput_items, d_in_ptr, d_out0_ptr, d_out1_ptr, stream=None) -> int:
        return check(lib.doa_find_local_max_work_dev(self._h, int(noutput_items), C.c_void_p(int(d_in_ptr)),
                                                     C.c_void_p(int(d_out0_ptr)), C.c_void_p(int(d_out1_ptr)),
                                                     _stream_ptr(stream)))


class rootMUSIC_linear_array(_Block):
    """doa.rootMUSIC_linear_array(norm_spacing, num_targets, inputs) — gr::sync_block
    (reference lib/rootMUSIC_linear_array_impl.cc:46-59)."""

    _destroy = staticmethod(lib.doa_rootMUSIC_linear_array_destroy)
    _set_precision = staticmethod(lib.doa_rootMUSIC_linear_array_set_internal_precision)

    def __init__(self, norm_spacing, num_targets, inputs):
        super().__init__()
        self.norm_spacing, self.num_targets, self.num_ant_ele = float(norm_spacing), int(num_targets), int(inputs)
        self._h = check_handle(lib.doa_rootMUSIC_linear_array_create(self.norm_spacing, self.num_targets,
                                                                     self.num_ant_ele), "rootMUSIC_linear_array")
        self.in_sig = [(_C64, self.num_ant_ele ** 2)]
        self.out_sig = [(_F32, self.num_targets)]   # io_signature::make(1, num_targets, ...): port 0 only is written

    def work(self, noutput_items, input_items, output_items) -> int:
        n = int(noutput_items)
        a = np.ascontiguousarray(input_items[0], dtype=_C64)
        out = output_items[0]
        assert a.size >= n * self.num_ant_ele ** 2
        assert out.dtype == _F32 and out.flags.c_contiguous and out.size >= n * self.num_targets
        return check(lib.doa_rootMUSIC_linear_array_work(self._h, n, _vp(a), _vp(out)))

    def work_dev(self, noutput_items, d_in_ptr, d_out_ptr, stream=None) -> int:
        return check(lib.doa_rootMUSIC_linear_array_work_dev(self._h, int(noutput_items), C.c_void_p(int(d_in_ptr)),
                                                             C.c_void_p(int(d_out_ptr)), _stream_ptr(stream)))


    def debug(self, R_items: np.ndarray):
        """(angles [n, M] float32, roots [n, 2N-2] complex128, status [n] int32) for parity tests."""
        a = np.ascontiguousarray(R_items, dtype=_C64).reshape(-1, self.num_ant_ele ** 2)
        n = a.shape[0]
        ang = np.empty((n, self.num_targets), dtype=_F32)
        roots = np.empty((n, 2 * self.num_ant_ele - 2), dtype=np.complex128)
        status = np.empty(n, dtype=np.int32)
        check(lib.doa_rootMUSIC_linear_array_debug(self._h, n, _vp(a), _vp(ang), _vp(roots), _vp(status)))
        return ang, roots, status

    def select_debug(self, roots: np.ndarray):
        """The device's root-selection stage alone (reference lib/rootMUSIC_linear_array_impl.cc:122-145) on the given
        roots [n, 2N-2] complex128 -> (angles [n, M] float32, status [n] int32; 1 = no interior root)."""
        z = np.ascontiguousarray(roots, dtype=np.complex128).reshape(-1, 2 * self.num_ant_ele - 2)
        n = z.shape[0]
        ang = np.empty((n, self.num_targets), dtype=_F32)
        status = np.empty(n, dtype=np.int32)
        check(lib.doa_rootMUSIC_linear_array_select_debug(self._h, n, _vp(z), _vp(ang), _vp(status)))
        return ang, status


class calibrate_lin_array(_Block):
    """doa.calibrate_lin_array(norm_spacing, num_ant_ele, pilot_angle) — gr::sync_block, vlen N^2
    complex in, vlen N complex out (reference lib/calibrate_lin_array_impl.cc:46-75)."""

    _destroy = staticmethod(lib.doa_calibrate_lin_array_destroy)
    _set_precision = staticmethod(lib.doa_calibrate_lin_array_set_internal_precision)

    def __init__(self, norm_spacing, num_ant_ele, pilot_angle):
        super().__init__()
        self.norm_spacing, self.num_ant_ele, self.pilot_angle = float(norm_spacing), int(num_ant_ele), float(pilot_angle)
        self._h = check_handle(lib.doa_calibrate_lin_array_create(self.norm_spacing, self.num_ant_ele, self.pilot_angle),
                               "calibrate_lin_array")
        self.in_sig = [(_C64, self.num_ant_ele ** 2)]
        self.out_sig = [(_C64, self.num_ant_ele)]

    def work(self, noutput_items, input_items, output_items) -> int:
        n = int(noutput_items)
        a = np.ascontiguousarray(input_items[0], dtype=_C64)
        out = output_items[0]
        assert a.size >= n * self.num_ant_ele ** 2
        assert out.dtype == _C64 and out.flags.c_contiguous and out.size >= n * self.num_ant_ele
        return check(lib.doa_calibrate_lin_array_work(self._h, n, _vp(a), _vp(out)))

    def work_dev(self, noutput_items, d_in_ptr, d_out_ptr, stream=None) -> int:
        return check(lib.doa_calibrate_lin_array_work_dev(self._h, int(noutput_items), C.c_void_p(int(d_in_ptr)),
                                                          C.c_void_p(int(d_out_ptr)), _stream_ptr(stream)))


class music_pipeline(_Block):
    """autocorrelate -> MUSIC_lin_array -> find_local_max(num_targets, pspectrum_len, 0, 180) on
    device-resident streams (the wiring of apps/run_MUSIC_lin_array_simulation.grc); the batch
    entry point the benchmark drives.  Not a block of the reference."""

    _destroy = staticmethod(lib.doa_music_pipeline_destroy)
    _set_precision = staticmethod(lib.doa_music_pipeline_set_internal_precision)

    def __init__(self, inputs, snapshot_size, overlap_size, avg_method, norm_spacing, num_targets,
                 pspectrum_len, max_batch=4096):
        super().__init__()
        self.inputs, self.snapshot_size, self.overlap_size = int(inputs), int(snapshot_size), int(overlap_size)
        self.avg_method, self.norm_spacing = int(avg_method), float(norm_spacing)
        self.num_targets, self.pspectrum_len, self.max_batch = int(num_targets), int(pspectrum_len), int(max_batch)
        self._h = check_handle(lib.doa_music_pipeline_create(self.inputs, self.snapshot_size, self.overlap_size,
                                                             self.avg_method, self.norm_spacing, self.num_targets,
                                                             self.pspectrum_len, self.max_batch), "music_pipeline")
        # as a flowgraph block (gr::doa::music_pipeline of the C++ shells, grc/doa_music_pipeline.xml): N complex
        # streams in; out0 = peak locations, out1 = peak values, out2 = spectrum
        self.in_sig = [(_C64, 1)] * self.inputs
        self.out_sig = [(_F32, self.num_targets), (_F32, self.num_targets), (_F32, self.pspectrum_len)]

    def history(self) -> int:
        return self.overlap_size + 1                          # as doa.autocorrelate (autocorrelate_impl.cc:56-57)

    def forecast(self, noutput_items: int) -> int:
        return (self.snapshot_size - self.overlap_size) * int(noutput_items)

    def general_work(self, noutput_items, input_items, output_items):
        """output_items = [argmax [>=n, M], max [>=n, M], spectrum [>=n, P]] (trailing ports may be omitted).
        Returns (items produced, items consumed per input)."""
        n, done = int(noutput_items), 0
        S = self.snapshot_size - self.overlap_size
        while done < n:
            k = min(self.max_batch, n - done)
            am = output_items[0][done:done + k]
            mx = output_items[1][done:done + k] if len(output_items) > 1 else np.empty((k, self.num_targets), _F32)
            sp = output_items[2][done:done + k] if len(output_items) > 2 else None
            done += self.work(k, [a[done * S:] for a in input_items], mx, am, spectrum_out=sp)
        return done, self.forecast(done)

    def fuse_antenna_correction(self, correction) -> None:
        _fuse(lib.doa_music_pipeline_fuse_antenna_correction, self._h, correction, self.inputs)

    def set_stages(self, cov=True, evd=True, scan=True) -> None:
        """Profiling aid: drop stages from later work_dev calls (their outputs keep the previous call's values)."""
        check(lib.doa_music_pipeline_set_stages(self._h, (1 if cov else 0) | (2 if evd else 0) | (4 if scan else 0)))

    def inject_failure(self, chunk_index: int) -> None:
        """Test aid: the next work() call fails in chunk `chunk_index` as if a HIP call had (one-shot; -1 disarms)."""
        check(lib.doa_music_pipeline_inject_failure(self._h, int(chunk_index)))

    def lanes_idle(self) -> bool:
        """Test aid: True when neither copy/compute lane of the host-buffer entry has work pending."""
        return bool(check(lib.doa_music_pipeline_lanes_idle(self._h)))

    def work_dev(self, noutput_items, d_input_ptrs, d_cov_ptr, d_spec_ptr, d_max_ptr, d_argmax_ptr, stream=None) -> int:
        return check(lib.doa_music_pipeline_work_dev(
            self._h, int(noutput_items), ptr_array(d_input_ptrs), C.c_void_p(int(d_cov_ptr or 0)),
            C.c_void_p(int(d_spec_ptr or 0)), C.c_void_p(int(d_max_ptr)), C.c_void_p(int(d_argmax_ptr)),
            _stream_ptr(stream)))

    def set_lanes(self, n_lanes: int) -> None:
        check(lib.doa_music_pipeline_set_lanes(self._h, int(n_lanes)))

    def set_lane_streams(self, streams) -> None:
        """Lanes on streams the caller created (torch.cuda.Stream objects or hipStream_t values); the caller keeps them alive."""
        self._lane_streams = list(streams)
        check(lib.doa_music_pipeline_set_lane_streams(self._h, len(self._lane_streams),
                                                      ptr_array([_stream_ptr(s).value or 0 for s in self._lane_streams])))

    def synchronize(self) -> None:
        """Host-side join of the lanes (after work_dev_batches(..., stream=doa.DETACHED))."""
        check(lib.doa_music_pipeline_synchronize(self._h))

    def work_dev_batches(self, noutput_items, d_input_ptrs, d_cov_ptrs, d_spec_ptrs, d_max_ptrs, d_argmax_ptrs, stream=None) -> int:
        """n_batches = len(d_max_ptrs) batches in one call, overlapped over the handle's own lanes (doa_hip.h).
        d_input_ptrs: n_batches lists of `inputs` device pointers (or one flat list); d_cov_ptrs / d_spec_ptrs: lists of
        device pointers (0 = not wanted) or None."""
        return self.prepare_batches(noutput_items, d_input_ptrs, d_cov_ptrs, d_spec_ptrs, d_max_ptrs, d_argmax_ptrs, stream)()

    def prepare_batches(self, noutput_items, d_input_ptrs, d_cov_ptrs, d_spec_ptrs, d_max_ptrs, d_argmax_ptrs, stream=None):
        """The argument marshalling of work_dev_batches done once: returns a callable that makes the C call (a C or C++
        caller has its pointer arrays at hand; a Python caller that repeats a call should not rebuild them every time)."""
        nb = len(d_max_ptrs)
        flat = [p for b in d_input_ptrs for p in b] if nb and isinstance(d_input_ptrs[0], (list, tuple)) else list(d_input_ptrs)
        assert len(flat) == nb * self.inputs and len(d_argmax_ptrs) == nb
        opt = lambda ptrs: None if ptrs is None else ptr_array([int(p or 0) for p in ptrs])
        args = (self._h, nb, int(noutput_items), ptr_array(flat), opt(d_cov_ptrs), opt(d_spec_ptrs), ptr_array(d_max_ptrs),
                ptr_array(d_argmax_ptrs), _stream_ptr(stream))
        fn = lib.doa_music_pipeline_work_dev_batches
        return lambda: check(fn(*args))

    def input_span(self, noutput_items) -> int:
        n = int(noutput_items)
        return 0 if n <= 0 else (n - 1) * (self.snapshot_size - self.overlap_size) + self.snapshot_size

    def work(self, noutput_items, input_items, max_out, argmax_out, cov_out=None, spectrum_out=None) -> int:
        """Host buffers in, host buffers out (numpy): input_items[k] = complex64 stream k starting at
        its first history sample; max_out / argmax_out [>=n, M] float32; cov_out [>=n, N*N] complex64
        and spectrum_out [>=n, P] float32 are optional."""
        n = int(noutput_items)
        span = self.input_span(n)
        arrs = []
        for k in range(self.inputs):
            a = np.ascontiguousarray(input_items[k], dtype=_C64)
            if a.shape[0] < span:
                raise ValueError(f"input {k}: {a.shape[0]} samples, need {span}")
            arrs.append(a)
        for o, dt, per in ((max_out, _F32, self.num_targets), (argmax_out, _F32, self.num_targets),
                           (cov_out, _C64, self.inputs ** 2), (spectrum_out, _F32, self.pspectrum_len)):
            if o is not None:
                assert o.dtype == dt and o.flags.c_contiguous and o.size >= n * per
        none = C.c_void_p(0)
        return check(lib.doa_music_pipeline_work(
            self._h, n, ptr_array([a.ctypes.data for a in arrs]), none if cov_out is None else _vp(cov_out),
            none if spectrum_out is None else _vp(spectrum_out), _vp(max_out), _vp(argmax_out)))


class root_pipeline(_Block):
    """autocorrelate -> rootMUSIC_linear_array on device-resident streams (the wiring of
    apps/run_RootMUSIC_lin_array_simulation.grc) as ONE handle: the Root-MUSIC branch of the hot path with the same entry
    points as music_pipeline (work_dev, work_dev_batches over the handle's lanes, detached form, host-buffer work).
    Not a block of the reference."""

    _destroy = staticmethod(lib.doa_root_pipeline_destroy)
    _set_precision = staticmethod(lib.doa_root_pipeline_set_internal_precision)

    def __init__(self, inputs, snapshot_size, overlap_size, avg_method, norm_spacing, num_targets, max_batch=4096):
        super().__init__()
        self.inputs, self.snapshot_size, self.overlap_size = int(inputs), int(snapshot_size), int(overlap_size)
        self.avg_method, self.norm_spacing = int(avg_method), float(norm_spacing)
        self.num_targets, self.max_batch = int(num_targets), int(max_batch)
        self._h = check_handle(lib.doa_root_pipeline_create(self.inputs, self.snapshot_size, self.overlap_size, self.avg_method,
                                                            self.norm_spacing, self.num_targets, self.max_batch), "root_pipeline")
        # as a flowgraph block (gr::doa::root_music_pipeline of the C++ shells): N complex streams in; out0 = angles
        self.in_sig = [(_C64, 1)] * self.inputs
        self.out_sig = [(_F32, self.num_targets)]

    def history(self) -> int:
        return self.overlap_size + 1                          # as doa.autocorrelate (autocorrelate_impl.cc:56-57)

    def forecast(self, noutput_items: int) -> int:
        return (self.snapshot_size - self.overlap_size) * int(noutput_items)

    def input_span(self, noutput_items) -> int:
        n = int(noutput_items)
        return 0 if n <= 0 else (n - 1) * (self.snapshot_size - self.overlap_size) + self.snapshot_size

    def general_work(self, noutput_items, input_items, output_items):
        """output_items = [angles [>=n, M]].  Returns (items produced, items consumed per input)."""
        n, done = int(noutput_items), 0
        S = self.snapshot_size - self.overlap_size
        while done < n:
            k = min(self.max_batch, n - done)
            done += self.work(k, [a[done * S:] for a in input_items], output_items[0][done:done + k])
        return done, self.forecast(done)

    def fuse_antenna_correction(self, correction) -> None:
        _fuse(lib.doa_root_pipeline_fuse_antenna_correction, self._h, correction, self.inputs)

    def inject_failure(self, chunk_index: int) -> None:
        """Test aid: the next work() / work_dev_batches() call fails in chunk / batch `chunk_index` (one-shot; -1 disarms)."""
        check(lib.doa_root_pipeline_inject_failure(self._h, int(chunk_index)))

    def lanes_idle(self) -> bool:
        return bool(check(lib.doa_root_pipeline_lanes_idle(self._h)))

    def work_dev(self, noutput_items, d_input_ptrs, d_cov_ptr, d_angles_ptr, d_status_ptr=None, stream=None) -> int:
        return check(lib.doa_root_pipeline_work_dev(
            self._h, int(noutput_items), ptr_array(d_input_ptrs), C.c_void_p(int(d_cov_ptr or 0)), C.c_void_p(int(d_angles_ptr)),
            C.c_void_p(int(d_status_ptr or 0)), _stream_ptr(stream)))

    def set_lanes(self, n_lanes: int) -> None:
        check(lib.doa_root_pipeline_set_lanes(self._h, int(n_lanes)))

    def set_lane_streams(self, streams) -> None:
        self._lane_streams = list(streams)
        check(lib.doa_root_pipeline_set_lane_streams(self._h, len(self._lane_streams),
                                                     ptr_array([_stream_ptr(s).value or 0 for s in self._lane_streams])))

    def synchronize(self) -> None:
        check(lib.doa_root_pipeline_synchronize(self._h))

    def work_dev_batches(self, noutput_items, d_input_ptrs, d_cov_ptrs, d_angles_ptrs, d_status_ptrs=None, stream=None) -> int:
        return self.prepare_batches(noutput_items, d_input_ptrs, d_cov_ptrs, d_angles_ptrs, d_status_ptrs, stream)()

    def prepare_batches(self, noutput_items, d_input_ptrs, d_cov_ptrs, d_angles_ptrs, d_status_ptrs=None, stream=None):
        """The argument marshalling of work_dev_batches done once: returns a callable that makes the C call."""
        nb = len(d_angles_ptrs)
        flat = [p for b in d_input_ptrs for p in b] if nb and isinstance(d_input_ptrs[0], (list, tuple)) else list(d_input_ptrs)
        assert len(flat) == nb * self.inputs
        opt = lambda ptrs: None if ptrs is None else ptr_array([int(p or 0) for p in ptrs])
        args = (self._h, nb, int(noutput_items), ptr_array(flat), opt(d_cov_ptrs), ptr_array(d_angles_ptrs), opt(d_status_ptrs),
                _stream_ptr(stream))
        fn = lib.doa_root_pipeline_work_dev_batches
        return lambda: check(fn(*args))

    def work(self, noutput_items, input_items, angles_out, cov_out=None) -> int:
        """Host buffers in, host buffers out (numpy): input_items[k] = complex64 stream k starting at its first history
        sample; angles_out [>=n, M] float32; cov_out [>=n, N*N] complex64 is optional.  Raises DoaError(DOA_ERR_NUMERIC) when
        an item has no root inside the unit circle (the outputs of the other items are valid)."""
        n = int(noutput_items)
        span = self.input_span(n)
        arrs = []
        for k in range(self.inputs):
            a = np.ascontiguousarray(input_items[k], dtype=_C64)
            if a.shape[0] < span:
                raise ValueError(f"input {k}: {a.shape[0]} samples, need {span}")
            arrs.append(a)
        for o, dt, per in ((angles_out, _F32, self.num_targets), (cov_out, _C64, self.inputs ** 2)):
            if o is not None:
                assert o.dtype == dt and o.flags.c_contiguous and o.size >= n * per
        none = C.c_void_p(0)
        return check(lib.doa_root_pipeline_work(self._h, n, ptr_array([a.ctypes.data for a in arrs]),
                                                none if cov_out is None else _vp(cov_out), _vp(angles_out)))


root_music_pipeline = root_pipeline      # the name of the C++ shell (gr::doa::root_music_pipeline) and of grc/doa_root_music_pipeline.xml


class compass_mean(_Block):
    """blocks.vector_to_streams(float, num_streams) + the averaging step of doa.compass
    (reference python/compass.py:134-136: next_angle = numpy.mean(input_items[0]) per work call),
    on the device.  The dial/LCD GUI of the compass is not reproduced."""

    _destroy = staticmethod(lib.doa_compass_mean_destroy)

    def __init__(self, num_streams):
        super().__init__()
        self.num_streams = int(num_streams)
        self._h = check_handle(lib.doa_compass_mean_create(self.num_streams), "compass_mean")
        self.next_angle = np.full(self.num_streams, np.nan, _F32)
        self.in_sig = [(_F32, self.num_streams)]
        self.out_sig = []

    def work(self, ninput_items, input_items, output_items=None) -> int:
        n = int(ninput_items)
        a = np.ascontiguousarray(input_items[0], dtype=_F32)
        assert a.size >= n * self.num_streams
        return check(lib.doa_compass_mean_work(self._h, n, _vp(a), _vp(self.next_angle)))

    def work_dev(self, ninput_items, d_in_ptr, d_next_angle_ptr, stream=None) -> int:
        return check(lib.doa_compass_mean_work_dev(self._h, int(ninput_items), C.c_void_p(int(d_in_ptr or 0)),
                                                   C.c_void_p(int(d_next_angle_ptr)), _stream_ptr(stream)))


class sim_source(_Block):
    """The signal front end of apps/run_MUSIC_lin_array_simulation.py (:66-74, :204-210) as one
    device-side generator: tones + per-source Gaussian noise through the array manifold, plus
    optional per-antenna noise.  tone_freq in cycles/sample."""

    _destroy = staticmethod(lib.doa_sim_source_destroy)

    def __init__(self, num_ant_ele, norm_spacing, theta_deg, tone_freq, tone_ampl=None, source_noise_ampl=None,
                 antenna_noise_sigma=0.0, seed=0):
        super().__init__()
        th = np.ascontiguousarray(np.atleast_1d(theta_deg), dtype=_F32)
        fr = np.ascontiguousarray(np.atleast_1d(tone_freq), dtype=np.float64)
        M = th.shape[0]
        if fr.shape[0] != M:
            raise ValueError("tone_freq and theta_deg differ in length")
        opt = []
        for v in (tone_ampl, source_noise_ampl):
            if v is None:
                opt.append(None)
            else:
                v = np.ascontiguousarray(np.atleast_1d(v), dtype=_F32)
                if v.shape[0] != M:
                    raise ValueError("per-source arrays differ in length")
                opt.append(v)
        self.num_ant_ele, self.num_sources = int(num_ant_ele), int(M)
        none = C.c_void_p(0)
        self._h = check_handle(lib.doa_sim_source_create(
            self.num_ant_ele, self.num_sources, float(norm_spacing), _vp(th), _vp(fr),
            none if opt[0] is None else _vp(opt[0]), none if opt[1] is None else _vp(opt[1]),
            float(antenna_noise_sigma), int(seed) & 0xFFFFFFFFFFFFFFFF), "sim_source")

    def seek(self, sample_index) -> None:
        check(lib.doa_sim_source_seek(self._h, int(sample_index)))

    def tell(self) -> int:
        return int(lib.doa_sim_source_tell(self._h))

    def work(self, noutput_items, output_items) -> int:
        n = int(noutput_items)
        for o in output_items[:self.num_ant_ele]:
            assert o.dtype == _C64 and o.flags.c_contiguous and o.size >= n
        return check(lib.doa_sim_source_work(self._h, n, ptr_array([o.ctypes.data for o in output_items[:self.num_ant_ele]])))

    def work_dev(self, noutput_items, d_output_ptrs, stream=None) -> int:
        return check(lib.doa_sim_source_work_dev(self._h, int(noutput_items), ptr_array(d_output_ptrs), _stream_ptr(stream)))


def set_internal_precision(bits: int) -> None:
    check(lib.doa_set_internal_precision(int(bits)))


def get_internal_precision() -> int:
    return int(lib.doa_get_internal_precision())


def device_count() -> int:
    return int(lib.doa_hip_device_count())


def evd_fallback_count(reset: bool = False) -> int:
    """Diagnostics: items whose eigendecomposition left the signal-subspace fast path for the Jacobi fall-back."""
    return int(lib.doa_hip_evd_fallback_count(1 if reset else 0))
