"""ctypes binding of libdoa_hip.so (the C ABI declared in include/doa_hip.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails
loudly if the shared object has not been built (`make -C gr-doa_amd`), and every block
constructor raises if no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "DOA_HIP_LIB", os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libdoa_hip.so")))


class DoaError(RuntimeError):
    """A C-ABI call failed; `.status` is the doa_status code."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libdoa_hip: {message} (status {status})")
        self.status = status


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (make -C gr-doa_amd, or "
        f"python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")



def _preload_torch_hip_runtime() -> None:
    """PyTorch's ROCm wheel ships its own libamdhip64.so / libhsa-runtime64.so (same SONAMEs as
    /opt/rocm's).  Two HSA runtimes in one process cannot both open the GPU, so when torch is
    installed its runtime is loaded first and libdoa_hip.so's DT_NEEDED `libamdhip64.so.7` binds
    to it; without torch the library binds to /opt/rocm through its RUNPATH (the C++ deployment).
    Set DOA_HIP_SYSTEM_RUNTIME=1 to skip this (then do not import torch in the same process)."""
    if os.environ.get("DOA_HIP_SYSTEM_RUNTIME"):
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


_preload_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)

_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/doa_hip.h and include/doa_hip_test.h one to one
SIGNATURES = {
    "doa_last_error": (C.c_char_p, []),
    "doa_hip_abi_version": (C.c_int, []),
    "doa_hip_device_count": (C.c_int, []),
    "doa_stream_stride_bytes": (C.c_size_t, [C.c_size_t]),
    "doa_set_internal_precision": (C.c_int, [C.c_int]),
    "doa_get_internal_precision": (C.c_int, []),
    "doa_autocorrelate_create": (_vp, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "doa_autocorrelate_destroy": (None, [_vp]),
    "doa_autocorrelate_history": (C.c_int, [_vp]),
    "doa_autocorrelate_forecast": (C.c_int, [_vp, C.c_int]),
    "doa_autocorrelate_input_span": (C.c_longlong, [_vp, C.c_int]),
    "doa_autocorrelate_work": (C.c_int, [_vp, C.c_int, _vpp, _vp]),
    "doa_autocorrelate_work_dev": (C.c_int, [_vp, C.c_int, _vpp, _vp, _vp]),
    "doa_hip_evd_fallback_count": (C.c_longlong, [C.c_int]),
    "doa_MUSIC_lin_array_set_internal_precision": (C.c_int, [_vp, C.c_int]),
    "doa_rootMUSIC_linear_array_set_internal_precision": (C.c_int, [_vp, C.c_int]),
    "doa_calibrate_lin_array_set_internal_precision": (C.c_int, [_vp, C.c_int]),
    "doa_music_pipeline_set_internal_precision": (C.c_int, [_vp, C.c_int]),
    "doa_MUSIC_lin_array_create": (_vp, [C.c_float, C.c_int, C.c_int, C.c_int]),
    "doa_MUSIC_lin_array_destroy": (None, [_vp]),
    "doa_MUSIC_lin_array_work": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "doa_MUSIC_lin_array_work_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_MUSIC_lin_array_debug": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_MUSIC_lin_array_items_total": (C.c_longlong, [_vp]),
    "doa_find_local_max_create": (_vp, [C.c_int, C.c_int, C.c_float, C.c_float]),
    "doa_find_local_max_destroy": (None, [_vp]),
    "doa_find_local_max_work": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_find_local_max_work_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "doa_rootMUSIC_linear_array_create": (_vp, [C.c_float, C.c_int, C.c_int]),
    "doa_rootMUSIC_linear_array_destroy": (None, [_vp]),
    "doa_rootMUSIC_linear_array_work": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "doa_rootMUSIC_linear_array_debug": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "doa_rootMUSIC_linear_array_select_debug": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_rootMUSIC_linear_array_work_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_antenna_correction_create": (_vp, [C.c_int, C.c_char_p]),
    "doa_antenna_correction_create_gains": (_vp, [C.c_int, _vp]),
    "doa_antenna_correction_destroy": (None, [_vp]),
    "doa_antenna_correction_gains": (C.c_int, [_vp, _vp]),
    "doa_antenna_correction_work": (C.c_int, [_vp, C.c_int, _vpp, _vpp]),
    "doa_antenna_correction_work_dev": (C.c_int, [_vp, C.c_int, _vpp, _vpp, _vp]),
    "doa_autocorrelate_fuse_antenna_correction": (C.c_int, [_vp, _vp]),
    "doa_music_pipeline_fuse_antenna_correction": (C.c_int, [_vp, _vp]),
    "doa_calibrate_lin_array_create": (_vp, [C.c_float, C.c_int, C.c_float]),
    "doa_calibrate_lin_array_destroy": (None, [_vp]),
    "doa_calibrate_lin_array_work": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "doa_calibrate_lin_array_work_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_music_pipeline_create": (_vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]),
    "doa_music_pipeline_destroy": (None, [_vp]),
    "doa_music_pipeline_set_stages": (C.c_int, [_vp, C.c_int]),
    "doa_music_pipeline_work_dev": (C.c_int, [_vp, C.c_int, _vpp, _vp, _vp, _vp, _vp, _vp]),
    "doa_music_pipeline_work": (C.c_int, [_vp, C.c_int, _vpp, _vp, _vp, _vp, _vp]),
    "doa_music_pipeline_work_dev_batches": (C.c_int, [_vp, C.c_int, C.c_int, _vpp, _vpp, _vpp, _vpp, _vpp, _vp]),
    "doa_music_pipeline_set_lanes": (C.c_int, [_vp, C.c_int]),
    "doa_music_pipeline_set_lane_streams": (C.c_int, [_vp, C.c_int, _vpp]),
    "doa_music_pipeline_synchronize": (C.c_int, [_vp]),
    "doa_music_pipeline_inject_failure": (C.c_int, [_vp, C.c_int]),
    "doa_music_pipeline_lanes_idle": (C.c_int, [_vp]),
    "doa_compass_mean_create": (_vp, [C.c_int]),
    "doa_compass_mean_destroy": (None, [_vp]),
    "doa_compass_mean_work": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "doa_compass_mean_work_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "doa_sim_source_create": (_vp, [C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, _vp, C.c_float, C.c_ulonglong]),
    "doa_sim_source_destroy": (None, [_vp]),
    "doa_sim_source_seek": (C.c_int, [_vp, C.c_longlong]),
    "doa_sim_source_tell": (C.c_longlong, [_vp]),
    "doa_sim_source_work": (C.c_int, [_vp, C.c_int, _vpp]),
    "doa_sim_source_work_dev": (C.c_int, [_vp, C.c_int, _vpp, _vp]),
    "doa_root_pipeline_create": (_vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]),
    "doa_root_pipeline_destroy": (None, [_vp]),
    "doa_root_pipeline_fuse_antenna_correction": (C.c_int, [_vp, _vp]),
    "doa_root_pipeline_work_dev": (C.c_int, [_vp, C.c_int, _vpp, _vp, _vp, _vp, _vp]),
    "doa_root_pipeline_work_dev_batches": (C.c_int, [_vp, C.c_int, C.c_int, _vpp, _vpp, _vpp, _vpp, _vp]),
    "doa_root_pipeline_synchronize": (C.c_int, [_vp]),
    "doa_root_pipeline_set_lanes": (C.c_int, [_vp, C.c_int]),
    "doa_root_pipeline_set_lane_streams": (C.c_int, [_vp, C.c_int, _vpp]),
    "doa_root_pipeline_set_internal_precision": (C.c_int, [_vp, C.c_int]),
    "doa_root_pipeline_work": (C.c_int, [_vp, C.c_int, _vpp, _vp, _vp]),
    # include/doa_hip_test.h (diagnostics, profiling, fault injection: the test suite's entry points)
    "doa_root_pipeline_inject_failure": (C.c_int, [_vp, C.c_int]),
    "doa_root_pipeline_lanes_idle": (C.c_int, [_vp]),
    "doa_hip_evd_fallback_counter_device_debug": (C.c_int, []),
    "doa_hip_lane_streams_verified_debug": (C.c_int, []),
    "doa_hip_lane_streams_set_aside_debug": (C.c_int, []),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return (lib.doa_last_error() or b"").decode("utf-8", "replace")


def check(status: int) -> int:
    """Raise DoaError for a negative doa_status, pass everything else through."""
    if status < 0:
        raise DoaError(status, last_error() or "call failed")
    return status


def check_handle(handle, what: str):
    if not handle:
        msg = last_error() or f"{what}: create failed"
        raise DoaError(-1, msg)
    return handle


def ptr_array(ptrs) -> "C.Array":
    """Host array of (device or host) pointers, as `const void* const*`."""
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = int(p)
    return arr
