"""Seeded synthetic multichannel streams for tests and the benchmark.

Signal model (SURVEY §8d, and the reference's own generators: the array manifold of
examples/@wpi_twinrx_doa_testbench/wpi_twinrx_doa_testbench.m:60-64, the tone + noise model of
music_test_input_gen.m:33-38,97-107 and of apps/run_MUSIC_lin_array_simulation.grc):

    x[n, t] = sum_m a_n(theta_m) * exp(j 2 pi f_m t) + sigma * w[n, t]
    a_n(theta) = exp(-j 2 pi cos(theta) * loc_n),  loc_n = d * ((N-1)/2 - n)

with per-antenna i.i.d. unit-variance complex Gaussian w.  Streams are stream-major `[N][T]`
complex64, matching GNU Radio's N separate stream pointers.
"""
from __future__ import annotations

import numpy as np


def array_locations(norm_spacing: float, num_ant_ele: int) -> np.ndarray:
    n = np.arange(num_ant_ele, dtype=np.float64)
    return float(norm_spacing) * ((num_ant_ele - 1) / 2.0 - n)


def manifold(norm_spacing: float, num_ant_ele: int, theta_deg) -> np.ndarray:
    """[N, M] array manifold matrix for the given directions (degrees)."""
    loc = array_locations(norm_spacing, num_ant_ele)
    th = np.deg2rad(np.atleast_1d(np.asarray(theta_deg, dtype=np.float64)))
    return np.exp(-2j * np.pi * np.cos(th)[None, :] * loc[:, None])


def tone_frequencies(num_targets: int) -> np.ndarray:
    """Distinct normalised tone frequencies (cycles/sample), one per source."""
    return 0.03125 * (np.arange(num_targets) + 1) + 0.0107


def make_streams(num_ant_ele: int, n_samples: int, theta_deg, norm_spacing: float, snr_db: float = 20.0,
                 seed: int = 0, freqs=None, per_source_noise=None) -> np.ndarray:
    """[N, n_samples] complex64.  snr_db is the per-source, per-antenna SNR (unit-amplitude tones,
    noise variance 10^(-snr/10)); snr_db=None gives noise-free streams (the rank-deficient case).
    per_source_noise: optional list of noise amplitudes added to each *source* before the manifold,
    as the simulation flowgraphs do (apps/run_MUSIC_lin_array_simulation.grc)."""
    rng = np.random.default_rng(seed)
    th = np.atleast_1d(np.asarray(theta_deg, dtype=np.float64))
    M = th.shape[0]
    f = tone_frequencies(M) if freqs is None else np.asarray(freqs, dtype=np.float64)
    t = np.arange(n_samples, dtype=np.float64)
    src = np.exp(2j * np.pi * f[:, None] * t[None, :])                      # [M, T]
    if per_source_noise is not None:
        amp = np.asarray(per_source_noise, dtype=np.float64)[:, None]
        src = src + amp * (rng.standard_normal(src.shape) + 1j * rng.standard_normal(src.shape))
    x = manifold(norm_spacing, num_ant_ele, th) @ src                        # [N, T]
    if snr_db is not None:
        sigma = 10.0 ** (-float(snr_db) / 20.0)
        w = (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape)) / np.sqrt(2.0)
        x = x + sigma * w
    return np.ascontiguousarray(x.astype(np.complex64))


def make_batch_streams(num_ant_ele: int, snapshot_size: int, batch: int, norm_spacing: float,
                       num_targets: int = 1, snr_db: float = 20.0, seed: int = 0,
                       theta_range=(20.0, 160.0)):
    """Benchmark workload: `batch` back-to-back snapshots (overlap 0), each with its own random
    directions drawn uniformly from theta_range.  Returns (streams [N, batch*K] complex64,
    thetas [batch, M] float64)."""
    rng = np.random.default_rng(seed)
    K = snapshot_size
    thetas = rng.uniform(theta_range[0], theta_range[1], size=(batch, num_targets))
    if num_targets > 1:   # keep sources resolvable
        thetas = np.sort(thetas, axis=1)
        thetas += np.arange(num_targets)[None, :] * 4.0
        thetas = np.clip(thetas, 5.0, 175.0)
    f = tone_frequencies(num_targets)
    t = np.arange(K, dtype=np.float64)
    loc = array_locations(norm_spacing, num_ant_ele)
    src = np.exp(2j * np.pi * f[:, None] * t[None, :])                      # [M, K]
    A = np.exp(-2j * np.pi * np.cos(np.deg2rad(thetas))[:, None, :] * loc[None, :, None])   # [B, N, M]
    x = A @ src[None, :, :]                                                  # [B, N, K]
    sigma = 10.0 ** (-float(snr_db) / 20.0)
    w = (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape)) / np.sqrt(2.0)
    x = x + sigma * w
    streams = np.ascontiguousarray(np.transpose(x, (1, 0, 2)).reshape(num_ant_ele, batch * K).astype(np.complex64))
    return streams, thetas


def make_batch_streams_torch(num_ant_ele: int, snapshot_size: int, batch: int, norm_spacing: float,
                             num_targets: int = 1, snr_db: float = 20.0, seed: int = 0, device="cuda",
                             theta_range=(20.0, 160.0)):
    """Same workload generated on the device with torch (benchmark setup, excluded from timing).
    Returns (list of N contiguous complex64 device tensors of batch*K samples, thetas [batch, M])."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    K = snapshot_size
    thetas = torch.rand((batch, num_targets), generator=g, device=device, dtype=torch.float64)
    thetas = theta_range[0] + (theta_range[1] - theta_range[0]) * thetas
    if num_targets > 1:
        thetas, _ = torch.sort(thetas, dim=1)
        thetas = thetas + torch.arange(num_targets, device=device, dtype=torch.float64)[None, :] * 4.0
        thetas = thetas.clamp(5.0, 175.0)
    f = torch.as_tensor(tone_frequencies(num_targets), device=device)
    t = torch.arange(K, device=device, dtype=torch.float64)
    loc = torch.as_tensor(array_locations(norm_spacing, num_ant_ele), device=device)
    src = torch.exp(2j * torch.pi * f[:, None] * t[None, :])                                  # [M, K]
    phase = -2.0 * torch.pi * torch.cos(torch.deg2rad(thetas))[:, None, :] * loc[None, :, None]
    A = torch.exp(1j * phase)                                                                 # [B, N, M]
    x = torch.matmul(A, src[None, :, :])                                                      # [B, N, K]
    sigma = 10.0 ** (-float(snr_db) / 20.0)
    wr = torch.randn(x.shape, generator=g, device=device, dtype=torch.float64)
    wi = torch.randn(x.shape, generator=g, device=device, dtype=torch.float64)
    x = x + (sigma / (2.0 ** 0.5)) * torch.complex(wr, wi)
    x = x.to(torch.complex64).permute(1, 0, 2).reshape(num_ant_ele, batch * K).contiguous()
    return stream_slab_torch([x[n] for n in range(num_ant_ele)]), thetas.cpu().numpy()


def stream_slab_torch(streams):
    """Copies N equally long complex64 device tensors into ONE allocation at the distance the library recommends
    (doa_stream_stride_bytes: streams whose addresses agree modulo 8 KiB share HBM channels; include/doa_hip.h) and
    returns the N views.  The layout a device-resident producer should use for the streams it hands to
    autocorrelate / music_pipeline.work_dev."""
    import torch
    from ._lib import lib

    n, nbytes = len(streams), streams[0].numel() * 8
    stride = int(lib.doa_stream_stride_bytes(nbytes))
    slab = torch.empty(n * stride + 4096, dtype=torch.uint8, device=streams[0].device)
    off0 = (-slab.data_ptr()) % 4096
    out = []
    for k, s in enumerate(streams):
        v = slab[off0 + k * stride: off0 + k * stride + nbytes].view(torch.complex64)
        v.copy_(s)
        out.append(v)
    return out
