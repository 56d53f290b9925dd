#!/usr/bin/env python3
"""Mints the committed golden vectors (tests/golden/*.npz) from the oracle.

The reference stores no golden vectors and cannot be run in the build image, so these
fixtures are produced by the numpy/LAPACK oracle (oracle/doa_oracle.py: the same cgemm / cheevd /
cgeev routines Armadillo forwards to, plus the fp64 evaluation of the same formulas) on seeded
inputs restating the reference's QA scenarios and simulation flowgraphs (tests/scenarios.py).
Each file holds inputs AND expected outputs, so the GPU box needs neither scipy's LAPACK nor this
script to check the HIP path against them:
    x       [N, T] complex64   input streams (history samples included)
    R32     [n, N*N] complex64 autocorrelate items (fp32 path)          R64: fp64 evaluation
    spec32  [n, P] float32     MUSIC dB spectrum, LAPACK-fp32 path       spec64: fp64 evaluation
    Q64     [n, P] float64     null spectrum, fp64
    PN64    [n, N, N] c128     noise projector, fp64
    val32/loc32, val64/loc64   find_local_max(M, P, 0, 180) on spec32 / float32(spec64)
    root32, root64 [n, M]      Root-MUSIC angles (deg)
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import importlib.util
_spec = importlib.util.spec_from_file_location("doa_sim_standalone", os.path.join(ROOT, "gr-doa_amd", "python", "doa", "sim.py"))
sim = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(sim)
import types
_doa = types.ModuleType("doa"); _doa.sim = sim; sys.modules.setdefault("doa", _doa)
import doa_oracle as oracle                                   # noqa: E402
from scenarios import SCENARIOS, make_input                  # noqa: E402

GOLDEN = ["qa_music_aoa23", "qa_root_aoa52", "grc_music_sim", "grc_root_sim", "bench_cfg2", "bench_cfg3", "three_ant_fb", "five_ant"]
N_ITEMS = 4   # keep the fixtures small (tens of KB each)

for name in GOLDEN:
    c, x = make_input(name)
    n = min(N_ITEMS, c["n"])
    S = c["K"] - c["ovl"]
    x = np.ascontiguousarray(x[:, : (n - 1) * S + c["K"]])
    N, M, P, d = c["N"], c["M"], c["P"], c["d"]
    R32 = oracle.autocorrelate(x, c["K"], c["ovl"], c["fb"], n)
    R64 = oracle.autocorrelate(x, c["K"], c["ovl"], c["fb"], n, precision="f64")
    spec32 = oracle.music_lin_array(R32, d, M, N, P, "f32")
    spec64, Q64, PN64 = oracle.music_lin_array(R32, d, M, N, P, "f64", return_parts=True)
    val32, loc32 = oracle.find_local_max(spec32, M, P, 0.0, 180.0)
    val64, loc64 = oracle.find_local_max(spec64.astype(np.float32), M, P, 0.0, 180.0)
    root32 = oracle.root_music(R32, d, M, N, "f32")
    root64 = oracle.root_music(R32, d, M, N, "f64")
    out = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(out, x=x, R32=R32, R64=R64.astype(np.complex128), spec32=spec32, spec64=spec64, Q64=Q64, PN64=PN64,
                        val32=val32, loc32=loc32, val64=val64, loc64=loc64, root32=root32, root64=root64,
                        cfg=np.array([N, M, P, c["K"], c["ovl"], c["fb"], n], dtype=np.int64), d=np.float32(d))
    print(f"{name}: {os.path.getsize(out) / 1024:.0f} KiB")
