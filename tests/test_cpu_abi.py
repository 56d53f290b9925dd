"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol include/doa_hip.h
declares, the Python binding covers them one to one, constructors validate their arguments and —
in this GPU-less container — everything that needs the device fails loudly (no CPU fallback).
No compute is attempted."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "doa_hip.h")
TEST_HEADER = os.path.join(ROOT, "include", "doa_hip_test.h")       # diagnostics / profiling / fault injection: not the boundary


def _declared_in(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"DOA_HIP_API\s+[\w\s\*]+?\b(doa_\w+)\s*\(", src)))


def _declared():
    return sorted(set(_declared_in(HEADER)) | set(_declared_in(TEST_HEADER)))


def test_the_boundary_header_carries_no_test_hooks():
    """doa_hip.h is what a gr-doa block shell binds; fault injection, stage masks and *_debug entries live in doa_hip_test.h."""
    boundary = _declared_in(HEADER)
    assert not [n for n in boundary if n.endswith("_debug") or "inject" in n or n.endswith("_lanes_idle") or n.endswith("_set_stages")]
    hooks = _declared_in(TEST_HEADER)
    assert "doa_music_pipeline_inject_failure" in hooks and "doa_root_pipeline_inject_failure" in hooks
    assert not set(boundary) & set(hooks)


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for blk in ("autocorrelate", "MUSIC_lin_array", "find_local_max", "rootMUSIC_linear_array"):
        for suffix in ("create", "work", "work_dev", "destroy"):
            assert f"doa_{blk}_{suffix}" in names
    assert "doa_music_pipeline_work_dev" in names and "doa_last_error" in names
    for suffix in ("create", "work", "work_dev", "work_dev_batches", "synchronize", "set_lanes", "destroy"):
        assert f"doa_root_pipeline_{suffix}" in names


def test_library_exports_every_declared_symbol():
    import doa
    from doa import _lib
    for name in _declared():
        assert hasattr(_lib.lib, name), f"{name} declared in doa_hip.h but not exported by {doa.LIB_PATH}"
    assert sorted(_lib.SIGNATURES) == _declared(), "python binding and header out of sync"
    assert _lib.lib.doa_hip_abi_version() == 1


def test_library_is_not_older_than_its_sources():
    lib = os.path.join(ROOT, "gr-doa_amd", "lib", "libdoa_hip.so")
    src_dir = os.path.join(ROOT, "gr-doa_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
    newest = max(newest, os.path.getmtime(HEADER), os.path.getmtime(TEST_HEADER))
    assert os.path.getmtime(lib) >= newest, "libdoa_hip.so is stale: run make -C gr-doa_amd"


def test_argument_validation_happens_in_create():
    import doa
    bad = [
        (doa.autocorrelate, (0, 16, 0, 0)), (doa.autocorrelate, (4, 16, 16, 0)), (doa.autocorrelate, (4, 0, 0, 0)),
        (doa.MUSIC_lin_array, (0.5, 4, 4, 64)), (doa.MUSIC_lin_array, (0.6, 1, 4, 64)), (doa.MUSIC_lin_array, (0.5, 1, 4, 0)),
        (doa.find_local_max, (0, 64, 0.0, 1.0)), (doa.find_local_max, (2, 64, 1.0, 1.0)),
        (doa.rootMUSIC_linear_array, (0.5, 4, 4)), (doa.rootMUSIC_linear_array, (0.7, 1, 4)),
    ]
    for cls, args in bad:
        with pytest.raises(doa.DoaError) as ei:
            cls(*args)
        assert "no HIP device" not in str(ei.value)      # rejected on the arguments, before the device
    with pytest.raises(doa.DoaError):
        doa.set_internal_precision(16)
    assert doa.get_internal_precision() == 64


def test_no_device_means_failure_not_fallback():
    import doa
    if doa.device_count() > 0:
        pytest.skip("a HIP device is present")
    for cls, args in [(doa.autocorrelate, (4, 1024, 0, 0)), (doa.MUSIC_lin_array, (0.5, 1, 4, 1024)),
                      (doa.find_local_max, (1, 1024, 0.0, 180.0)), (doa.rootMUSIC_linear_array, (0.5, 1, 4))]:
        with pytest.raises(doa.DoaError) as ei:
            cls(*args)
        assert "no CPU fallback" in str(ei.value)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "gr-doa_amd")
    for base, _dirs, files in os.walk(pkg):
        if os.sep + "build" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "doa_oracle" not in text and "liboracle" not in text, os.path.join(base, f)
    assert "oracle" not in open(HEADER).read()


def test_sharding_helper():
    from doa import sharding
    K, ovl, n = 2048, 512, 1001
    S = K - ovl
    shards = sharding.all_shards(n, 8, K, ovl)
    assert sum(s.n_snapshots for s in shards) == n
    pos = 0
    for s in shards:
        assert s.first_snapshot == pos
        assert s.sample_begin == s.first_snapshot * S
        assert s.sample_end == (s.first_snapshot + s.n_snapshots - 1) * S + K
        pos += s.n_snapshots
    # consecutive shards overlap by exactly the history halo
    for a, b in zip(shards, shards[1:]):
        assert a.sample_end - b.sample_begin == ovl
    assert sharding.job_throughput([10, 10], [1.0, 2.0]) == 10.0


def test_stream_stride_rule():
    """doa_stream_stride_bytes: at least the stream, 16-byte aligned, and 4.5 KiB into the 8 KiB period the HBM channels
    repeat with, so that up to 16 streams laid out with it start at 16 distinct offsets modulo 8 KiB (include/doa_hip.h)."""
    from doa import _lib
    f = _lib.lib.doa_stream_stride_bytes
    for nbytes in (0, 8, 8 * 1024 * 4096, 8 * 1000 * 37, 8191, 8192, 8193, 123456792):
        s = int(f(nbytes))
        assert s >= nbytes and s % 16 == 0 and s % 8192 == 4608 and s - nbytes < 8192 + 4608 + 1
    s = int(f(8 * 1024 * 4096))
    assert len({(k * s) % 8192 for k in range(16)}) == 16
