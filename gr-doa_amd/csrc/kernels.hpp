// kernels.hpp — launch entry points shared between the per-block C-ABI files and the pipeline.
#pragma once
#include "common.hpp"

namespace doa {

// K1  (autocorrelate.hip)
// d_gain_outer: optional N*N float2 table w[a + b*N] = g_a conj(g_b) (fused antenna correction), or NULL
size_t autocorrelate_workspace_bytes(int N, int K, int ovl, int n_out);
int launch_autocorrelate(int N, int K, int ovl, int avg, int n_out, const void *const *d_in, void *d_out,
                         hipStream_t st, const void *d_gain_outer = nullptr, void *d_workspace = nullptr);

// Host-built tables of MUSIC_lin_array (music.hip): z_i = exp(j*psi_i), psi_i = k_i * d with
// k_i = float(-2*pi*cos(theta_i)) on the reference's float-accumulated theta grid.
struct MusicTables {
    int N = 0, M = 0, P = 0;
    float norm_spacing = 0.f;
    DevBuf d_z;   // P float2  (float scan)
    DevBuf d_zd;  // P double2 (double scan)
    int build(float norm_spacing, int num_targets, int num_ant_ele, int pspectrum_len);
    void release() { d_z.release(); d_zd.release(); }
};

// coefficient record per item: [u0, Re u1, Im u1, ..., Re u_{N-1}, Im u_{N-1}, pad] = 2N floats
inline int coef_stride(int N) { return 2 * N; }

// K2+K3: batched Hermitian EVD + noise projector + diagonal sums.  d_coef (float records, for the
// scan), d_coef_d (double records, same layout, for the root finder) and d_pn may each be NULL.
// d_cheb (optional; N <= 4, evd_bits == 64): a second record per item, 8 doubles, holding the null spectrum's polynomial
// in the form the lean scan kernel evaluates, Q = A(c) + s B(c): [a0, a1, a2, a3, b0, b1, b2, 0] (music_scan_impl.hpp,
// ChebQ) -- the change of basis is per item, so it belongs to the kernel that runs once per item.
int launch_music_evd(int N, int M, int n_items, const void *d_R, void *d_coef, void *d_coef_d, void *d_pn,
                     int evd_bits, hipStream_t st, void *d_cheb = nullptr);
inline bool music_uses_cheb(int N, int bits) { return N <= 4 && bits == 64; }
constexpr int kChebRecord = 8;      // doubles per item
// diagnostics: items that left the signal-subspace fast path of K2+K3 for the Jacobi fall-back since the last reset
long long evd_fallback_count(bool reset);
// the calling thread's current device's counter (allocated on first use; nullptr if that fails) and, for the tests, the device
// an allocation lives on
unsigned long long *evd_fallback_counter();
int evd_fallback_counter_device(const void *p);
// calibrate_lin_array (calibrate.hip): d_pilot = N float2 (pilot steering vector), d_out = n_items*N float2
int launch_calibrate(int N, int n_items, const void *d_R, const void *d_pilot, void *d_out, int bits, hipStream_t st);
// K4: spectrum scan in float (bits == 32, float coefficient records) or double (bits == 64, double
// records).  d_q (un-normalised null spectrum, P floats per item) may be NULL.
// With `peaks` (+ d_max/d_argmax) the find_local_max step is fused into the scan when the fast path
// applies; *peaks_done tells the caller whether it still has to launch K5 itself.
struct PeakTables;
int launch_music_scan(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec, void *d_q,
                      hipStream_t st, const PeakTables *peaks = nullptr, void *d_max = nullptr,
                      void *d_argmax = nullptr, bool *peaks_done = nullptr, bool store_spectrum = true,
                      const void *d_cheb = nullptr);

// K5 (find_local_max.hip)
struct PeakTables {
    int M = 0, L = 0;
    float x_min = 0.f, x_max = 0.f;
    DevBuf d_x;  // L floats, float-accumulated x axis
    int build(int num_max_vals, int vector_len, float x_min, float x_max);
    void release() { d_x.release(); }
};
int launch_find_local_max(const PeakTables &t, int n_items, const void *d_in, void *d_max, void *d_argmax,
                          hipStream_t st);

// K6 (root_music.hip): polynomial roots from the DOUBLE coefficient records -> angles.
// d_roots (optional, diagnostics): the 2N-2 roots found per item as double2, in the kernel's lane order.
int launch_root_music(int N, int M, float norm_spacing, int n_items, const void *d_coef, void *d_out,
                      void *d_status, hipStream_t st, void *d_roots = nullptr);
// the selection stage of K6 alone, on caller-supplied roots (n_items x (2N-2) double2): diagnostics
int launch_root_select(int N, int M, float norm_spacing, int n_items, const void *d_roots, void *d_out, void *d_status,
                       hipStream_t st);

}  // namespace doa
