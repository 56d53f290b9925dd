// music.hip — MUSIC_lin_array on gfx950: host-side constructor tables, the K4 dispatch and the C ABI.
//
// Replaces gr::doa::MUSIC_lin_array (reference lib/MUSIC_lin_array_impl.cc):
//   ctor  :47-87,98-104  element positions, float-accumulated theta grid, steering vectors
//   work  :121-144       eig_sym (LAPACK cheevd 'V','U', ascending) -> U_N = first N-M vectors
//                        -> P_N = U_N U_N^H -> out[i] = 1/Re(a_i^H P_N a_i) -> 10 log10(out/max)
//
// Where the pieces live:
//   * K2+K3 (EVD, noise projector, diagonal sums u_l = sum_r P_N[r+l, r]): music_evd.hip, jacobi.hpp.
//   * K4 (scan): for a ULA a_i^H P_N a_i is Hermitian-Toeplitz in the element index,
//         Q(psi_i) = u_0 + 2 Re sum_{l=1}^{N-1} u_l z_i^l,   z_i = exp(j psi_i), psi_i = k_i d,
//     so each angle costs N-1 complex Horner steps instead of N^2+N complex MACs (HBM-bound: 8N B in,
//     4P B out per item).  Kernels in music_scan_impl.hpp, one translation unit per polynomial size.
#include "kernels.hpp"
#include "music_scan.hpp"
#include "peak_device.hpp"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace doa {

// ---------------------------------------------------------------------------------------------
// host-side tables (constructor work of the reference block)
// ---------------------------------------------------------------------------------------------
int MusicTables::build(float norm_spacing_, int num_targets, int num_ant_ele, int pspectrum_len)
{
    N = num_ant_ele; M = num_targets; P = pspectrum_len; norm_spacing = norm_spacing_;
    // theta grid: float accumulator, sum formed in double (lib/MUSIC_lin_array_impl.cc:64-72)
    std::vector<float> theta(P);
    theta[0] = 0.0f;
    float theta_prev = 0.0f;
    for (int ii = 1; ii < P; ii++) {
        float th = (float)(theta_prev + 180.0 / P);
        theta_prev = th;
        theta[ii] = (float)(M_PI * th / 180.0);
    }
    // amv (:98-104): a_i[n] = exp(j * (-2 pi cos(theta_i)) * loc_n), loc_n = d*0.5*(N-1-2n) (:57-61), so
    // the phase step between neighbouring elements is psi_i = -2 pi cos(theta_i) * d.  Evaluated in
    // double from the float theta grid; the reference's float roundings of the scalar, of loc_n and of
    // each phase (~2e-7 rad) are not replayed (they are part of its own fp32 error budget).
    std::vector<float2> z(P);
    std::vector<double2> zd(P);
    const double d = (double)norm_spacing;
    for (int ii = 0; ii < P; ii++) {
        const double psi = -1.0 * 2 * M_PI * std::cos((double)theta[ii]) * d;
        z[ii] = make_float2((float)std::cos(psi), (float)std::sin(psi));
        zd[ii] = make_double2(std::cos(psi), std::sin(psi));
    }
    int rc = d_z.reserve(sizeof(float2) * (size_t)P);
    if (rc == DOA_OK) rc = d_zd.reserve(sizeof(double2) * (size_t)P);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(d_z.p, z.data(), sizeof(float2) * (size_t)P, hipMemcpyHostToDevice));
    DOA_HIP_TRY(hipMemcpy(d_zd.p, zd.data(), sizeof(double2) * (size_t)P, hipMemcpyHostToDevice));
    return DOA_OK;
}

// ---------------------------------------------------------------------------------------------
// K4: spectrum scan — kernels in music_scan_impl.hpp, one translation unit per polynomial size
// ---------------------------------------------------------------------------------------------
DOA_SCAN_SIZES(DOA_SCAN_EXTERN)

int launch_music_scan(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec, void *d_q,
                      hipStream_t st, const PeakTables *peaks, void *d_max, void *d_argmax, bool *peaks_done,
                      bool store_spectrum, const void *d_cheb)
{
    if (peaks_done) *peaks_done = false;
    if (n_items <= 0) return DOA_OK;
    ScanPeakArgs pk;
    pk.cheb = (bits == 64) ? d_cheb : nullptr;
    if (peaks && d_max && d_argmax && peaks->L == t.P) {
        pk.xaxis = peaks->d_x.as<float>(); pk.val = (float *)d_max; pk.loc = (float *)d_argmax; pk.M = peaks->M;
        pk.store = store_spectrum;
    }
    bool done = false;
    // compiled polynomial sizes: 2, 3, 4, 6, 8, 12, 16 (an array of n elements uses the next size up
    // with zero high-order coefficients)
    const int n = t.N;
    if (n < 2 || n > DOA_MAX_ANT_ELE) {
        set_error("MUSIC: num_ant_ele=%d outside the built range 2..%d", n, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    if (n == 2) done = launch_scan_n<2>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n == 3) done = launch_scan_n<3>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n == 4) done = launch_scan_n<4>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 6) done = launch_scan_n<6>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 8) done = launch_scan_n<8>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else if (n <= 12) done = launch_scan_n<12>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    else done = launch_scan_n<16>(t, bits, n_items, d_coef, d_spec, d_q, pk, st);
    if (peaks_done) *peaks_done = done;
    DOA_HIP_TRY(hipGetLastError());
    return DOA_OK;
}

}  // namespace doa

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct doa_MUSIC_lin_array {
    doa::MusicTables tab;
    int bits = 64;   // internal precision of EVD + scan (doa_set_internal_precision)
    int device = 0;
    long long items_total = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_in, d_out, d_coef, d_pn, d_q, d_cheb;
};

static int music_validate(const char *who, float norm_spacing, int num_targets, int num_ant_ele)
{
    // grc/doa_MUSIC_lin_array.xml:33-35 (inputs > 0, inputs > num_targets, norm_spacing <= 0.5); the
    // reference ctor does not validate (num_targets >= N would index cols(0,-1)), so create fails here.
    if (num_ant_ele <= 0 || num_targets <= 0 || num_targets >= num_ant_ele) {
        doa::set_error("%s: need 0 < num_targets < num_ant_ele (got %d, %d)", who, num_targets, num_ant_ele);
        return DOA_ERR_INVALID_ARG;
    }
    if (!(norm_spacing > 0.0f) || norm_spacing > 0.5f) {
        doa::set_error("%s: need 0 < norm_spacing <= 0.5 (got %g)", who, (double)norm_spacing);
        return DOA_ERR_INVALID_ARG;
    }
    if (num_ant_ele > DOA_MAX_ANT_ELE) {
        doa::set_error("%s: num_ant_ele=%d exceeds DOA_MAX_ANT_ELE=%d", who, num_ant_ele, DOA_MAX_ANT_ELE);
        return DOA_ERR_UNSUPPORTED;
    }
    return DOA_OK;
}

extern "C" {

doa_MUSIC_lin_array_t *doa_MUSIC_lin_array_create(float norm_spacing, int num_targets, int num_ant_ele,
                                                  int pspectrum_len)
{
    doa::clear_error();
    if (music_validate("MUSIC_lin_array", norm_spacing, num_targets, num_ant_ele) != DOA_OK) return nullptr;
    if (pspectrum_len <= 0) {
        doa::set_error("MUSIC_lin_array: pspectrum_len must be > 0 (got %d)", pspectrum_len);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_MUSIC_lin_array();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->device = dev;
    h->bits = doa::internal_precision_bits();
    if (h->tab.build(norm_spacing, num_targets, num_ant_ele, pspectrum_len) != DOA_OK ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        if (!*doa_last_error()) doa::set_error("MUSIC_lin_array: device setup failed");
        doa_MUSIC_lin_array_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_MUSIC_lin_array_destroy(doa_MUSIC_lin_array_t *h)
{
    if (!h) return;
    h->tab.release();
    h->d_in.release(); h->d_out.release(); h->d_coef.release(); h->d_pn.release(); h->d_q.release(); h->d_cheb.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

long long doa_MUSIC_lin_array_items_total(const doa_MUSIC_lin_array_t *h) { return h ? h->items_total : 0; }

int doa_MUSIC_lin_array_work_dev(doa_MUSIC_lin_array_t *h, int noutput_items, const void *d_input_items0,
                                 void *d_output_items0, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!d_input_items0 || !d_output_items0))) {
        doa::set_error("MUSIC_lin_array_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int N = h->tab.N;
    int rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(N) * sizeof(double));
    const bool cheb = doa::music_uses_cheb(N, h->bits);
    if (rc == DOA_OK && cheb) rc = h->d_cheb.reserve((size_t)noutput_items * doa::kChebRecord * sizeof(double));
    if (rc != DOA_OK) return rc;
    const bool dbl = (h->bits == 64);
    rc = doa::launch_music_evd(N, h->tab.M, noutput_items, d_input_items0, dbl ? nullptr : h->d_coef.p,
                               dbl ? h->d_coef.p : nullptr, nullptr, h->bits, st, cheb ? h->d_cheb.p : nullptr);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_scan(h->tab, h->bits, noutput_items, h->d_coef.p, d_output_items0, nullptr, st, nullptr, nullptr, nullptr,
                                nullptr, true, cheb ? h->d_cheb.p : nullptr);
    if (rc != DOA_OK) return rc;
    h->items_total += noutput_items;
    return noutput_items;
}

int doa_MUSIC_lin_array_work(doa_MUSIC_lin_array_t *h, int noutput_items, const void *input_items0,
                             void *output_items0)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!input_items0 || !output_items0))) {
        doa::set_error("MUSIC_lin_array_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int N = h->tab.N, P = h->tab.P;
    const size_t in_bytes = (size_t)noutput_items * N * N * sizeof(float2);
    const size_t out_bytes = (size_t)noutput_items * P * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(out_bytes);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_MUSIC_lin_array_work_dev(h, noutput_items, h->d_in.p, h->d_out.p, h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

int doa_MUSIC_lin_array_debug(doa_MUSIC_lin_array_t *h, int noutput_items, const void *input_items0,
                              void *projector_out, void *null_spectrum_out)
{
    doa::clear_error();
    if (!h || noutput_items <= 0 || !input_items0) {
        doa::set_error("MUSIC_lin_array_debug: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int N = h->tab.N, P = h->tab.P;
    const size_t in_bytes = (size_t)noutput_items * N * N * sizeof(float2);
    const size_t sp_bytes = (size_t)noutput_items * P * sizeof(float);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(sp_bytes);
    if (rc == DOA_OK) rc = h->d_q.reserve(sp_bytes);
    if (rc == DOA_OK) rc = h->d_pn.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_coef.reserve((size_t)noutput_items * doa::coef_stride(N) * sizeof(double));
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    const bool dbl = (h->bits == 64);
    rc = doa::launch_music_evd(N, h->tab.M, noutput_items, h->d_in.p, dbl ? nullptr : h->d_coef.p,
                               dbl ? h->d_coef.p : nullptr, h->d_pn.p, h->bits, h->stream);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_scan(h->tab, h->bits, noutput_items, h->d_coef.p, h->d_out.p, h->d_q.p, h->stream);
    if (rc != DOA_OK) return rc;
    if (projector_out)
        DOA_HIP_TRY(hipMemcpyAsync(projector_out, h->d_pn.p, in_bytes, hipMemcpyDeviceToHost, h->stream));
    if (null_spectrum_out)
        DOA_HIP_TRY(hipMemcpyAsync(null_spectrum_out, h->d_q.p, sp_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

long long doa_hip_evd_fallback_count(int reset)
{
    doa::clear_error();
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return -1;
    return doa::evd_fallback_count(reset != 0);
}

int doa_hip_evd_fallback_counter_device_debug(void)
{
    doa::clear_error();
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return -1;
    return doa::evd_fallback_counter_device(doa::evd_fallback_counter());
}

int doa_MUSIC_lin_array_set_internal_precision(doa_MUSIC_lin_array_t *h, int bits)
{
    doa::clear_error();
    if (!h || (bits != 32 && bits != 64)) { doa::set_error("MUSIC_lin_array_set_internal_precision: need a handle and bits = 32 or 64"); return DOA_ERR_INVALID_ARG; }
    h->bits = bits;
    return DOA_OK;
}

}  // extern "C"
