"""`doa` — the hot-path slice of the reference's Python namespace (python/__init__.py:26-41,
swig/doa_swig.i:22-34), backed by hand-written gfx950 HIP kernels behind a C ABI.

    import doa
    blk = doa.autocorrelate(inputs, snapshot_size, overlap_size, avg_method)
    blk = doa.MUSIC_lin_array(norm_spacing, num_targets, inputs, pspectrum_len)
    blk = doa.find_local_max(num_max_vals, vector_len, x_min, x_max)
    blk = doa.rootMUSIC_linear_array(norm_spacing, num_targets, inputs)

Importing fails if gr-doa_amd/lib/libdoa_hip.so has not been built; constructing a block fails if
no HIP device is usable.  There is no CPU fallback.
"""
from ._lib import DoaError, LIB_PATH, last_error  # noqa: F401
from .blocks import (autocorrelate, antenna_correction, phase_correct_hier, read_phase_config, calibrate_lin_array, MUSIC_lin_array, find_local_max, rootMUSIC_linear_array,  # noqa: F401
                     music_pipeline, root_pipeline, root_music_pipeline, compass_mean, sim_source, set_internal_precision, get_internal_precision, device_count,
                     evd_fallback_count, DETACHED)
from . import runtime, sim, sharding, distributed, launch  # noqa: F401
