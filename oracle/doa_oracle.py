"""CPU oracle for the gr-doa hot path (TEST INFRASTRUCTURE — never imported by the product).

    autocorrelate -> MUSIC_lin_array (+ find_local_max) / rootMUSIC_linear_array

This is a numpy/scipy restatement of the reference's Armadillo arithmetic, one function per
reference routine, each citing the reference file:line it follows (paths relative to the
read-only reference tree).  Armadillo itself only forwards to BLAS/LAPACK for the heavy steps
(`cgemm`, `cheevd`, `cgeev`), so the fp32 path below calls those *same* LAPACK routines through
scipy's bundled OpenBLAS (`scipy.linalg.lapack.cheevd/cgeev`) and keeps every intermediate in the
type the reference keeps it in (complex64 / float32, with the few double-precision scalar
sub-expressions the C++ promotes).

PARITY STATUS: **parity unpinned** with respect to the reference's own outputs.  The reference
stores no golden vectors (its QA tests mint inputs and expectations at run time through Octave,
`python/qa_*.py`), and neither Armadillo, GNU Radio nor Octave exist in the build image, so the
reference could not be run.  What pins this oracle instead (tests/test_cpu_oracle_pins.py):
  * the reference's own deterministic QA scenario for find_local_max
    (`python/test001_findpeaks.m`, `python/test002_findpeaks.m`) checked against an independent
    peak finder (`scipy.signal.find_peaks`) to the reference's 5-decimal tolerance;
  * the reference's QA scenarios and tolerances for autocorrelate / MUSIC / Root-MUSIC
    (|dR| <= 1.0, |d angle| <= 2.0 deg: `python/qa_autocorrelate.py:82`,
    `python/qa_MUSIC_lin_array.py:96`, `python/qa_rootMUSIC_linear_array.py:87`) on seeded inputs;
  * a numpy restatement of the reference's Octave golden model (examples/@wpi_twinrx_doa_testbench/
    MUSIC.m, rMUSIC.m, autocorrelate.m), which the fp64 path below equals to rounding;
  * an fp64 evaluation (`precision="f64"`: zheevd/zgeev, double tables) of the same formulas.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this file.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is part of the image; keep the import local to the functions that need LAPACK
    from scipy.linalg import lapack as _lapack
except Exception:  # pragma: no cover
    _lapack = None

_F32 = np.float32
_C64 = np.complex64


# --------------------------------------------------------------------------------------------
# antenna_correction  (lib/antenna_correction_impl.cc) — the block in front of autocorrelate
# --------------------------------------------------------------------------------------------
def antenna_correction_gains(config_text: str, num_ant_ele: int) -> np.ndarray:
    """Constructor of doa::antenna_correction.  lib/antenna_correction_impl.cc:56-73: the file is read
    with `infile >> GainEst >> PhaseEst` (whitespace-separated floats, stops at the first token that
    does not parse); g_k = gr_complex(1.0/GainEst, 0) * exp(gr_complex(0, -PhaseEst)); more than
    num_ant_ele pairs / fewer -> std::invalid_argument."""
    tokens = config_text.split()
    vals = []
    for tok in tokens:
        try:
            vals.append(_F32(float(tok)))
        except ValueError:
            break
    pairs = len(vals) // 2
    if pairs > num_ant_ele:
        raise ValueError("Configuration file has too many inputs.")
    if pairs != num_ant_ele:
        raise ValueError("Configuration file does not have enough inputs.")
    g = np.empty(num_ant_ele, dtype=_C64)
    for k in range(num_ant_ele):
        gain, phase = vals[2 * k], vals[2 * k + 1]
        a = _F32(1.0 / float(gain))                                   # double division -> float
        e = _C64(complex(np.cos(_F32(-phase), dtype=_F32), np.sin(_F32(-phase), dtype=_F32)))   # exp(complex<float>(0,-p))
        g[k] = _C64(_C64(complex(a, 0.0)) * e)
    return g


def antenna_correction(input_items: np.ndarray, gains: np.ndarray) -> np.ndarray:
    """work(): out_k[i] = g_k * in_k[i] in complex<float>.  lib/antenna_correction_impl.cc:85-99."""
    x = np.asarray(input_items, dtype=_C64)
    return (np.asarray(gains, dtype=_C64)[:, None] * x).astype(_C64)


def phase_correct_phases(config_text: str):
    """python/phase_correct_hier.py:33-45 (read_config_file): lines split at '\n'; `filter(f, lines)` keeps a line when
    f(line) = float(line) is truthy -- a line that does not parse gives False, and so does a line that parses to 0.0."""
    out = []
    for line in config_text.split("\n"):
        try:
            v = float(line)
        except ValueError:
            continue
        if v:
            out.append(v)
    return out


def phase_correct(input_items: np.ndarray, phases) -> np.ndarray:
    """python/phase_correct_hier.py:86-104: port 0 -> blocks.copy; port p+1 -> multiply_const_vcc((numpy.exp(1j*phase_p),)),
    i.e. a complex<float> product with the double-precision exponential rounded to gr_complex."""
    x = np.asarray(input_items, dtype=_C64)
    if len(phases) != x.shape[0] - 1:
        raise ValueError("Not valid number of phase estimates")
    g = np.array([1.0 + 0.0j] + [np.exp(1j * float(p)) for p in phases], dtype=np.complex128).astype(_C64)
    return (g[:, None] * x).astype(_C64)


# --------------------------------------------------------------------------------------------
# autocorrelate  (lib/autocorrelate_impl.cc)
# --------------------------------------------------------------------------------------------
def gr_history_prepend(streams: np.ndarray, overlap_size: int) -> np.ndarray:
    """GNU Radio `set_history(overlap+1)` view of a stream start.

    lib/autocorrelate_impl.cc:57 — the scheduler hands `input_items[k]` pointing `overlap`
    samples *before* the first new sample and the history is zero-filled at stream start, so
    snapshot 0 begins with `overlap` zeros.  streams: [N, T] complex64 -> [N, overlap+T].
    """
    streams = np.asarray(streams, dtype=_C64)
    pad = np.zeros((streams.shape[0], overlap_size), dtype=_C64)
    return np.concatenate([pad, streams], axis=1)


def autocorrelate_noutput(n_samples_with_history: int, snapshot_size: int, overlap_size: int) -> int:
    """Number of whole windows in a buffer of `n_samples_with_history` samples per stream.

    lib/autocorrelate_impl.cc:75-80 (forecast): each output needs `snapshot-overlap` new items;
    window i spans [i*nonoverlap, i*nonoverlap + snapshot).
    """
    nonoverlap = snapshot_size - overlap_size
    if n_samples_with_history < snapshot_size:
        return 0
    return (n_samples_with_history - snapshot_size) // nonoverlap + 1


def autocorrelate(input_items: np.ndarray, snapshot_size: int, overlap_size: int,
                  avg_method: int, output_matrices: int | None = None,
                  precision: str = "f32") -> np.ndarray:
    """general_work of doa::autocorrelate.  lib/autocorrelate_impl.cc:83-118.

    input_items: [N, >= (n-1)*(K-ovl)+K] complex64, *including* the history samples (window i
    starts at sample i*(K-ovl), exactly `input_items[k] + i*d_nonoverlap_size`, :98).
    Returns [n, N*N] complex64; each item is the column-major N x N matrix the block writes
    (:103), R[a,b] = (1/K) sum_t x_a[t] conj(x_b[t]) (:106), optionally followed by
    R <- 0.5 R + (0.5/K) J conj(R) J (:107-108; note the second term is divided by K again —
    reproduced as written, also in examples/@wpi_twinrx_doa_testbench/autocorrelate.m:42-44).
    """
    x = np.asarray(input_items, dtype=_C64)
    n_in, total = x.shape
    K = int(snapshot_size)
    S = K - int(overlap_size)
    if output_matrices is None:
        output_matrices = autocorrelate_noutput(total, K, overlap_size)
    out = np.empty((output_matrices, n_in * n_in), dtype=_C64)
    J = np.fliplr(np.eye(n_in))                                   # :61-62
    cdt, rdt = (_C64, _F32) if precision == "f32" else (np.complex128, np.float64)
    for i in range(output_matrices):
        X = x[:, i * S:i * S + K].T.astype(cdt)                   # K x N, :95-100
        if X.shape[0] != K:
            raise ValueError("input shorter than the windows requested")
        # (1.0/K) * X.st() * conj(X): Armadillo folds the scalar into cgemm's alpha (as eT).
        R = (rdt(1.0 / K) * (X.T @ np.conj(X))).astype(cdt)       # :106
        if avg_method == 1:
            JRJ = (J.astype(cdt) @ np.conj(R) @ J.astype(cdt)).astype(cdt)
            R = (rdt(0.5) * R + rdt(0.5 / K) * JRJ).astype(cdt)   # :107-108
        out[i] = R.astype(_C64).reshape(-1, order="F")            # column-major, :103
    return out


# --------------------------------------------------------------------------------------------
# MUSIC_lin_array  (lib/MUSIC_lin_array_impl.cc)
# --------------------------------------------------------------------------------------------
def music_array_loc(norm_spacing: float, num_ant_ele: int) -> np.ndarray:
    """lib/MUSIC_lin_array_impl.cc:57-61: float(norm_spacing)*0.5*(N-1-2n), double expr -> float."""
    d = float(_F32(norm_spacing))
    return np.array([d * 0.5 * (num_ant_ele - 1 - 2 * nn) for nn in range(num_ant_ele)], dtype=_F32)


def music_theta_grid(pspectrum_len: int) -> np.ndarray:
    """lib/MUSIC_lin_array_impl.cc:64-72: float accumulator theta += 180.0/P (sum in double,
    stored to float each step), d_theta[i] = float(pi*theta/180.0)."""
    theta = np.zeros(pspectrum_len, dtype=_F32)
    theta_prev = _F32(0.0)
    step = 180.0 / pspectrum_len
    for ii in range(1, pspectrum_len):
        t = _F32(float(theta_prev) + step)
        theta_prev = t
        theta[ii] = _F32(np.pi * float(t) / 180.0)
    return theta


def music_steering(norm_spacing: float, num_ant_ele: int, pspectrum_len: int,
                   precision: str = "f32") -> np.ndarray:
    """Steering table d_vii_matrix (N x P).  lib/MUSIC_lin_array_impl.cc:75-86,98-104.

    amv: v = exp(i * ((-1.0*2*pi*cos(theta)) * array_loc)); the scalar is a double expression
    that Armadillo casts to float when it scales the fcolvec, the product scalar*loc is a float
    multiply, and exp is std::exp(complex<float>) = (cosf(y), sinf(y)) for a zero real part.
    precision="f64" is the formula itself in double: the float-accumulated theta grid is kept (it
    is a stored float member and a documented quirk of the block), but the element positions
    d*0.5*(N-1-2n), the scalar -2*pi*cos(theta), the phases and exp are not rounded to float.
    """
    loc = music_array_loc(norm_spacing, num_ant_ele)
    theta = music_theta_grid(pspectrum_len)
    if precision == "f32":
        # ASSUMPTION (parity unpinned): the reference writes an unqualified `cos(theta)` on a float (:103).  With <cmath>
        # in scope and `using namespace std` that resolves to std::cos(float); through <math.h> alone to ::cos(double) on
        # the promoted argument.  Which one its toolchain picked cannot be told from the source; the double form is taken
        # here.  The two differ by at most one float ulp of the scalar (<= 4e-7 in the phase), far inside the reference's own
        # fp32 error budget, and the f64 path below -- the one the device is held to -- does not depend on it.
        k = (-1.0 * 2 * np.pi * np.cos(theta.astype(np.float64))).astype(_F32)   # scalar -> float
        phase = (k[None, :] * loc[:, None]).astype(_F32)                          # float multiply
        return (np.cos(phase) + 1j * np.sin(phase)).astype(_C64)
    d = float(_F32(norm_spacing))
    loc64 = np.array([d * 0.5 * (num_ant_ele - 1 - 2 * nn) for nn in range(num_ant_ele)], dtype=np.float64)
    k = -1.0 * 2 * np.pi * np.cos(theta.astype(np.float64))
    phase = k[None, :] * loc64[:, None]
    return np.cos(phase) + 1j * np.sin(phase)


def _eig_sym(R: np.ndarray, precision: str):
    """arma::eig_sym (default "dc") = LAPACK ?heevd(jobz='V', uplo='U'), ascending eigenvalues;
    only the upper triangle is referenced.  lib/MUSIC_lin_array_impl.cc:128."""
    if precision == "f32":
        w, v, info = _lapack.cheevd(np.asarray(R, dtype=_C64), compute_v=1, lower=0)
    else:
        w, v, info = _lapack.zheevd(np.asarray(R, dtype=np.complex128), compute_v=1, lower=0)
    if info != 0:
        raise np.linalg.LinAlgError(f"heevd info={info}")
    return w, v


def noise_projector(R_item: np.ndarray, num_targets: int, num_ant_ele: int,
                    precision: str = "f32") -> np.ndarray:
    """U_N U_N^H with U_N = the N-M eigenvectors of the smallest eigenvalues.
    lib/MUSIC_lin_array_impl.cc:124-133 (identically lib/rootMUSIC_linear_array_impl.cc:108-116)."""
    N = num_ant_ele
    R = np.asarray(R_item).reshape(N, N, order="F")               # column-major item, :124
    _, V = _eig_sym(R, precision)
    U_N = V[:, :N - num_targets]                                  # :131
    return U_N @ U_N.conj().T                                     # :133


def music_null_spectrum(P_N: np.ndarray, A: np.ndarray) -> np.ndarray:
    """Q_i = Re( a_i^H P_N a_i ), evaluated as (row * P_N) * col like Armadillo's 3-operand
    product for a 1xN * NxN * Nx1 chain.  lib/MUSIC_lin_array_impl.cc:137-139."""
    T = A.conj().T @ P_N                                          # P x N   (d_vii_matrix_trans rows)
    Q = np.einsum("in,ni->i", T, A)
    return Q.real


def music_db_from_q(Q: np.ndarray, precision: str = "f32") -> np.ndarray:
    """lib/MUSIC_lin_array_impl.cc:140-142 on one item's null spectrum: out = 1.0/Q (a double division stored
    to float), out_vec = 10*log10(out_vec/out_vec.max()).  An angle is at exactly 0 dB iff its ROUNDED
    reciprocal equals the largest one; index_max / find_local_max then take the first such angle."""
    rdt = _F32 if precision == "f32" else np.float64
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (1.0 / np.asarray(Q, dtype=rdt).astype(np.float64)).astype(rdt)
        out = (out / out.max()).astype(rdt)
        return (rdt(10.0) * np.log10(out)).astype(rdt)


def music_lin_array(R_items: np.ndarray, norm_spacing: float, num_targets: int, num_ant_ele: int,
                    pspectrum_len: int, precision: str = "f32", return_parts: bool = False):
    """work() of doa::MUSIC_lin_array.  lib/MUSIC_lin_array_impl.cc:108-150.

    R_items: [n, N*N] complex64 (column-major items).  Returns [n, P] float32 dB spectrum,
    10*log10(out/max(out)) with out = 1/Re(Q) (:140-142).  With return_parts also returns the
    per-item null spectrum Q and projector P_N (the quantities parity is graded on).
    """
    R_items = np.asarray(R_items)
    n = R_items.shape[0]
    N, P = num_ant_ele, pspectrum_len
    A = music_steering(norm_spacing, N, P, precision)
    rdt = _F32 if precision == "f32" else np.float64
    spec = np.empty((n, P), dtype=rdt)
    Qs = np.empty((n, P), dtype=rdt)
    PNs = np.empty((n, N, N), dtype=_C64 if precision == "f32" else np.complex128)
    for item in range(n):
        P_N = noise_projector(R_items[item], num_targets, N, precision)
        Q = music_null_spectrum(P_N, A).astype(rdt)
        spec[item] = music_db_from_q(Q, precision)
        Qs[item] = Q
        PNs[item] = P_N
    if return_parts:
        return spec, Qs, PNs
    return spec


# --------------------------------------------------------------------------------------------
# calibrate_lin_array  (lib/calibrate_lin_array_impl.cc)
# --------------------------------------------------------------------------------------------
def calibrate_pilot_vector(norm_spacing: float, num_ant_ele: int, pilot_angle: float, precision: str = "f32"):
    """ctor: amv(v_temp, d_array_loc, pi*pilot_angle/180).  lib/calibrate_lin_array_impl.cc:57-70,83-89."""
    loc = music_array_loc(norm_spacing, num_ant_ele)
    theta = _F32(np.pi * float(_F32(pilot_angle)) / 180.0)
    if precision == "f32":
        k = _F32(-1.0 * 2 * np.pi * np.cos(float(theta)))
        ph = (k * loc).astype(_F32)
        return (np.cos(ph) + 1j * np.sin(ph)).astype(_C64)
    d = float(_F32(norm_spacing))
    loc64 = np.array([d * 0.5 * (num_ant_ele - 1 - 2 * nn) for nn in range(num_ant_ele)])
    ph = -1.0 * 2 * np.pi * np.cos(float(theta)) * loc64
    return np.cos(ph) + 1j * np.sin(ph)


def calibrate_lin_array(R_items: np.ndarray, norm_spacing: float, num_ant_ele: int, pilot_angle: float,
                        precision: str = "f32") -> np.ndarray:
    """work() of doa::calibrate_lin_array, literally: two eig_sym per item.
    lib/calibrate_lin_array_impl.cc:98-134.  Returns [n, N] complex; each row carries LAPACK's arbitrary
    unit-modulus eigenvector phase (compare after `calibrate_normalise`)."""
    N = num_ant_ele
    v = calibrate_pilot_vector(norm_spacing, N, pilot_angle, precision)
    cdt = _C64 if precision == "f32" else np.complex128
    R_items = np.asarray(R_items)
    out = np.empty((R_items.shape[0], N), dtype=cdt)
    for item in range(R_items.shape[0]):
        R = R_items[item].reshape(N, N, order="F")
        _, V = _eig_sym(R, precision)
        U_S = V[:, N - 1:N]                                       # :118
        U_S_sq = (U_S @ U_S.conj().T).astype(cdt)                  # :119
        W = (np.diag(np.conj(v)).astype(cdt) @ U_S_sq @ np.diag(v).astype(cdt)).astype(cdt)   # :122
        _, WV = _eig_sym(W, precision)                             # :123
        out[item] = WV[:, N - 1]                                   # :125
    return out


def calibrate_normalise(est: np.ndarray) -> np.ndarray:
    """Remove the arbitrary unit-modulus factor: rotate each row so that element 0 is real >= 0."""
    est = np.asarray(est)
    ph = np.where(np.abs(est[:, :1]) > 0, np.conj(est[:, :1]) / np.maximum(np.abs(est[:, :1]), 1e-300), 1.0)
    return est * ph


# --------------------------------------------------------------------------------------------
# find_local_max  (lib/find_local_max_impl.cc, lib/find_local_max_impl.h)
# --------------------------------------------------------------------------------------------
def find_local_max_x_axis(vector_len: int, x_min: float, x_max: float) -> np.ndarray:
    """lib/find_local_max_impl.cc:60-69: float accumulation x += (x_max-x_min)/L, all in float."""
    x = np.empty(vector_len, dtype=_F32)
    x[0] = _F32(x_min)
    x_prev = _F32(x_min)
    step = _F32(_F32(_F32(x_max) - _F32(x_min)) / _F32(vector_len))
    for ii in range(1, vector_len):
        x_prev = _F32(x_prev + step)
        x[ii] = x_prev
    return x


def _index_max(v: np.ndarray) -> int:
    """arma index_max (op_max::direct_max): best starts at -inf and is replaced only by a strictly
    greater element -> first occurrence of the maximum; NaNs never win; nothing > -inf -> 0."""
    ok = v > -np.inf                      # False for NaN and -inf
    if not ok.any():
        return 0
    return int(np.argmax(np.where(ok, v, -np.inf)))


def _peak_indices_one(v: np.ndarray) -> np.ndarray:
    """find_one_local_peak_indx: index_max (first occurrence).  lib/find_local_max_impl.h:53-56."""
    return np.array([_index_max(v)], dtype=np.int64)


def _peak_indices_many(v: np.ndarray, num_max_vals: int) -> np.ndarray:
    """find_more_than_one_local_peak_indxs.  lib/find_local_max_impl.cc:80-165."""
    L = v.shape[0]
    with np.errstate(invalid="ignore"):
        d = np.diff(v)
    # :89 sign(diff(in_vec)).  A NaN difference (NaN input, or inf - inf) is taken as sign 0, i.e. a flat --
    # the three-way (x > 0) ? 1 : (x < 0) ? -1 : 0 form; an Armadillo whose sign() propagates NaN would differ,
    # but non-finite vectors are outside anything the reference tests or can produce from finite input.
    s = np.where(d > 0, 1.0, np.where(d < 0, -1.0, 0.0)).astype(_F32)
    flats = np.nonzero(s == 0)[0]                                 # :92
    for idx in flats[::-1]:                                       # :94-107, right to left, in place
        nxt = min(int(idx) + 1, s.shape[0] - 1)
        s[idx] = 1.0 if s[nxt] >= 0 else -1.0
    all_pk = np.nonzero(np.diff(s) == -2)[0] + 1                  # :114
    all_pks = v[all_pk]
    # sort_index(all_pks, "descend") (:137) — Armadillo's non-stable sort; ties are ordered by
    # position here (lowest index first), the one documented deviation (tie order is unspecified
    # in the reference).
    order = np.argsort(-all_pks.astype(np.float64), kind="stable")
    n_valid = order.shape[0]
    pk = np.empty(num_max_vals, dtype=np.int64)
    if n_valid >= num_max_vals:                                   # :141-144
        pk[:] = all_pk[order[:num_max_vals]]
    else:                                                         # :145-163
        if n_valid == 0:
            fill = _index_max(v)                                  # global arg-max
        else:
            # NOTE (reproduced quirk): the reference assigns `all_pks_sorted_indx(0)` — the
            # position of the best peak *inside the peak list*, not its index in the input
            # vector — and uses that number as the fill index (:153,160).
            fill = int(order[0])
        for ind in range(num_max_vals):
            pk[ind] = all_pk[order[ind]] if ind < n_valid else fill
    return pk


def find_local_max(in_items: np.ndarray, num_max_vals: int, vector_len: int,
                   x_min: float, x_max: float):
    """work() of doa::find_local_max.  lib/find_local_max_impl.cc:167-194.

    Returns (max_vals [n, M], arg_max [n, M]); port 0 = in_vec(pk_indxs) in descending-value
    order, port 1 = sort(x_axis(pk_indxs), "descend") — the two ports are sorted independently
    (:186-188).
    """
    v_all = np.asarray(in_items, dtype=_F32).reshape(-1, vector_len)
    x_axis = find_local_max_x_axis(vector_len, x_min, x_max)
    n = v_all.shape[0]
    vals = np.empty((n, num_max_vals), dtype=_F32)
    locs = np.empty((n, num_max_vals), dtype=_F32)
    for item in range(n):
        v = v_all[item]
        pk = _peak_indices_one(v) if num_max_vals == 1 else _peak_indices_many(v, num_max_vals)
        vals[item] = v[pk]
        locs[item] = np.sort(x_axis[pk])[::-1]
    return vals, locs


# --------------------------------------------------------------------------------------------
# rootMUSIC_linear_array  (lib/rootMUSIC_linear_array_impl.cc)
# --------------------------------------------------------------------------------------------
def root_music_polynomial(P_N: np.ndarray) -> np.ndarray:
    """get_roots_polynomial, coefficient part.  lib/rootMUSIC_linear_array_impl.cc:68-80.
    u[ii+N-1] = sum(diag(P_N, ii)) for sub-diagonals ii<0, mirrored conjugates above, trace in
    the middle; then u <- (-1/u[2N-2]) u."""
    N = P_N.shape[0]
    u = np.zeros(2 * N - 1, dtype=P_N.dtype)
    for ii in range(-N + 1, 0):
        u[ii + N - 1] = np.sum(np.diagonal(P_N, offset=ii))
        u[N - 1 - ii] = np.conj(u[ii + N - 1])
    u[N - 1] = np.sum(np.diagonal(P_N))
    u = (P_N.dtype.type(-1.0) / u[2 * N - 2]) * u
    return u.astype(P_N.dtype)


def root_music_roots(P_N: np.ndarray, precision: str = "f32") -> np.ndarray:
    """Companion matrix (ones on the first sub-diagonal, last column = u[0..2N-3]) and its
    eigenvalues via eig_gen = LAPACK ?geev without vectors.
    lib/rootMUSIC_linear_array_impl.cc:55-58,82-86."""
    N = P_N.shape[0]
    n = 2 * N - 2
    u = root_music_polynomial(P_N)
    C = np.zeros((n, n), dtype=P_N.dtype)
    C[np.arange(1, n), np.arange(0, n - 1)] = 1.0
    C[:, n - 1] = u[:n]
    if precision == "f32":
        w, _, _, info = _lapack.cgeev(C.astype(_C64), compute_vl=0, compute_vr=0)
    else:
        w, _, _, info = _lapack.zgeev(C.astype(np.complex128), compute_vl=0, compute_vr=0)
    if info != 0:
        raise np.linalg.LinAlgError(f"geev info={info}")
    return w


def root_music_select(roots: np.ndarray, norm_spacing: float, num_targets: int, precision: str = "f32") -> np.ndarray:
    """The selection stage of work(), lib/rootMUSIC_linear_array_impl.cc:122-145, on a given set of
    polynomial roots: dist = 1 - |z| (:122); keep dist > 0, strictly inside (:125-127); num_targets
    times take index_min(dist), angle = 180 acos(arg(z) / (2 pi d)) / pi, mark the root used with inf
    (:131-141) -- so with fewer interior roots than targets the remaining picks hit an inf entry,
    arg(inf + 0i) = 0, i.e. 90 degrees; with none, index_min runs on an empty vector (an Armadillo
    error: ValueError here); |arg / (2 pi d)| > 1 gives NaN -- then sort ascending (:144)."""
    M = num_targets
    d = float(_F32(norm_spacing))
    rdt = _F32 if precision == "f32" else np.float64
    roots = np.asarray(roots)
    with np.errstate(invalid="ignore"):
        dist = (rdt(1.0) - np.abs(roots)).astype(rdt)             # :122
        inside = np.nonzero(dist > 0.0)[0]                        # :125
    roots_in = roots[inside].copy()
    dist_in = dist[inside].copy()
    aoa = np.empty(M, dtype=_F32)
    for ii in range(M):                                           # :131-141
        if dist_in.shape[0] == 0:
            raise ValueError("no root strictly inside the unit circle (Armadillo index_min error)")
        k = int(np.argmin(dist_in))
        z = roots_in[k]
        if np.isinf(z.real):
            ang = rdt(0.0)                                        # arg(inf+0i) = 0 -> 90 deg
        else:
            ang = rdt(np.angle(z))                                # std::arg in the root's type
        with np.errstate(invalid="ignore"):
            aoa[ii] = _F32(180.0 * np.arccos(float(ang) / (2 * np.pi * d)) / np.pi)
        dist_in[k] = np.inf
        roots_in[k] = complex(np.inf, 0.0)
    return np.sort(aoa)                                           # :144


def root_music(R_items: np.ndarray, norm_spacing: float, num_targets: int, num_ant_ele: int,
               precision: str = "f32") -> np.ndarray:
    """work() of doa::rootMUSIC_linear_array.  lib/rootMUSIC_linear_array_impl.cc:90-152.
    Returns [n, M] float32 angles in degrees, ascending."""
    R_items = np.asarray(R_items)
    n = R_items.shape[0]
    N, M = num_ant_ele, num_targets
    out = np.empty((n, M), dtype=_F32)
    for item in range(n):
        P_N = noise_projector(R_items[item], M, N, precision)
        roots = root_music_roots(P_N, precision)
        out[item] = root_music_select(roots, norm_spacing, M, precision)
    return out


# --------------------------------------------------------------------------------------------
# whole path, as the flowgraph wires it (apps/run_MUSIC_lin_array_simulation.grc)
# --------------------------------------------------------------------------------------------
def music_pipeline(input_items: np.ndarray, snapshot_size: int, overlap_size: int, avg_method: int,
                   norm_spacing: float, num_targets: int, pspectrum_len: int,
                   output_matrices: int | None = None, precision: str = "f32"):
    """autocorrelate -> MUSIC_lin_array -> find_local_max(M, P, 0, 180)."""
    N = np.asarray(input_items).shape[0]
    R = autocorrelate(input_items, snapshot_size, overlap_size, avg_method, output_matrices)
    spec = music_lin_array(R, norm_spacing, num_targets, N, pspectrum_len, precision)
    vals, locs = find_local_max(spec.astype(_F32), num_targets, pspectrum_len, 0.0, 180.0)
    return R, spec, vals, locs


# ---- downstream of the path: vector_to_streams + compass averaging ------------
def compass_mean(argmax_items: np.ndarray, num_streams: int) -> np.ndarray:
    """blocks.vector_to_streams(float, num_streams) followed by one compass per stream, each doing
    `numpy.mean(input_items[0])` over the items of the work call (python/compass.py:134-136; wiring
    apps/run_MUSIC_lin_array_simulation.py:199,236-239).  numpy.mean on the float32 stream, exactly
    as the reference calls it."""
    a = np.asarray(argmax_items, dtype=np.float32).reshape(-1, num_streams)
    if a.shape[0] == 0:
        return np.full(num_streams, np.nan, np.float32)
    return np.array([np.mean(np.ascontiguousarray(a[:, m])) for m in range(num_streams)], dtype=np.float32)


# ---- upstream of the path: the simulation flowgraphs' signal front end --------
def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
    counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32.  Pinned by the Random123 known-answer
    vectors in tests/test_cpu_oracle_pins.py."""
    c = np.array(counter, dtype=np.uint64, copy=True)
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        n0 = (p1 >> s32) ^ c[..., 1] ^ k0
        n2 = (p0 >> s32) ^ c[..., 3] ^ k1
        c = np.stack([n0, p1 & mask, n2, p0 & mask], axis=-1)
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return c.astype(np.uint32)


def _box_muller(w0: np.ndarray, w1: np.ndarray) -> np.ndarray:
    u1 = ((w0 >> np.uint32(9)).astype(np.float64) + 0.5) * 2.0 ** -23
    u2 = ((w1 >> np.uint32(9)).astype(np.float64) + 0.5) * 2.0 ** -23
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2) + 1j * r * np.sin(2.0 * np.pi * u2)


def sim_noise_stream(seed: int, stream_id: int, first_sample: int, n_samples: int) -> np.ndarray:
    """Complex standard-normal pairs (g + j g') of one noise stream: Philox counter =
    (sample-pair index lo, hi, stream id, 0), key = seed; words (0,1) -> sample 2i, (2,3) -> 2i+1."""
    assert first_sample % 2 == 0
    pairs = np.arange(first_sample // 2, (first_sample + n_samples + 1) // 2, dtype=np.uint64)
    ctr = np.stack([pairs & np.uint64(0xFFFFFFFF), pairs >> np.uint64(32),
                    np.full_like(pairs, stream_id), np.zeros_like(pairs)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    g = np.stack([_box_muller(r[:, 0], r[:, 1]), _box_muller(r[:, 2], r[:, 3])], axis=-1).reshape(-1)
    return g[:n_samples]


def sim_source(num_ant_ele: int, norm_spacing: float, theta_deg, tone_freq, n_samples: int, tone_ampl=None,
               source_noise_ampl=None, antenna_noise_sigma: float = 0.0, seed: int = 0,
               first_sample: int = 0) -> np.ndarray:
    """sig_source_c + noise_source_c per source -> add -> multiply_matrix_cc(array manifold)
    (apps/run_MUSIC_lin_array_simulation.py:66-74,204-210), plus optional per-antenna noise
    sigma (g + j g')/sqrt(2) (music_test_input_gen.m:97-107).  float64 evaluation -> complex64."""
    th = np.deg2rad(np.atleast_1d(np.asarray(theta_deg, dtype=np.float32)).astype(np.float64))
    f = np.atleast_1d(np.asarray(tone_freq, dtype=np.float64))
    M = th.shape[0]
    amp = np.ones(M) if tone_ampl is None else np.atleast_1d(np.asarray(tone_ampl, np.float32)).astype(np.float64)
    sn = np.zeros(M) if source_noise_ampl is None else np.atleast_1d(np.asarray(source_noise_ampl, np.float32)).astype(np.float64)
    d = float(np.float32(norm_spacing))
    loc = d * ((num_ant_ele - 1) / 2.0 - np.arange(num_ant_ele))                       # :70
    A = np.exp(-2j * np.pi * np.cos(th)[None, :] * loc[:, None]).astype(np.complex64).astype(np.complex128)
    t = np.arange(first_sample, first_sample + n_samples, dtype=np.float64)
    src = np.empty((M, n_samples), np.complex128)
    for m in range(M):
        cyc = f[m] * t
        src[m] = amp[m] * np.exp(2j * np.pi * (cyc - np.floor(cyc)))
        if sn[m] != 0.0:
            src[m] += sn[m] * sim_noise_stream(seed, m, first_sample, n_samples)
    x = A @ src
    sigma = float(np.float32(antenna_noise_sigma))
    if sigma != 0.0:
        for n in range(num_ant_ele):
            x[n] += sigma / np.sqrt(2.0) * sim_noise_stream(seed, M + n, first_sample, n_samples)
    return np.ascontiguousarray(x.astype(np.complex64))
