// calibrate.hip — gr::doa::calibrate_lin_array on gfx950 (reuses the batched EVD).
//
// Replaces reference lib/calibrate_lin_array_impl.cc:
//   ctor :46-75   element positions, pilot steering vector v = exp(j*(-2 pi cos(theta_p))*loc),
//                 theta_p = pi*pilot_angle/180 (float)
//   work :98-134  per covariance item: eig_sym -> U_S = eigenvector of the largest eigenvalue ->
//                 W = diag(conj v) U_S U_S^H diag(v) -> eig_sym(W) -> output its last eigenvector
//                 (Soon, Tong, Huang, Liu: gains/phases = eigenvector of W for the unit eigenvalue).
// W is rank one, so that eigenvector is conj(v) .* U_S up to a unit-modulus factor that the reference
// inherits from LAPACK's (arbitrary) eigenvector phase; here the factor is fixed so that element 0 — the
// reference antenna — is real and non-negative.  One launch of the group-parallel Jacobi kernel
// (music.hip) does all of it; no second EVD is needed.
#include "kernels.hpp"

#include <cmath>
#include <vector>

struct doa_calibrate_lin_array {
    int N = 0, bits = 64, device = 0;
    hipStream_t stream = nullptr;
    doa::DevBuf d_pilot, d_in, d_out;
};

extern "C" {

doa_calibrate_lin_array_t *doa_calibrate_lin_array_create(float norm_spacing, int num_ant_ele, float pilot_angle)
{
    doa::clear_error();
    // grc/doa_calibrate_lin_array.xml:27-28
    if (num_ant_ele <= 1 || num_ant_ele > DOA_MAX_ANT_ELE || !(norm_spacing > 0.0f) || norm_spacing > 0.5f) {
        doa::set_error("calibrate_lin_array: need 1 < num_ant_ele <= %d and 0 < norm_spacing <= 0.5 (got %d, %g)",
                       DOA_MAX_ANT_ELE, num_ant_ele, (double)norm_spacing);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_calibrate_lin_array();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->N = num_ant_ele; h->device = dev; h->bits = doa::internal_precision_bits();
    // pilot steering vector, as the reference builds it (:57-70): float locations, float theta, the
    // scalar -2 pi cos(theta) cast to float, float phase, float exp
    std::vector<float2> v(num_ant_ele);
    const float theta = (float)(M_PI * pilot_angle / 180.0);
    const float k = (float)(-1.0 * 2 * M_PI * std::cos((double)theta));
    for (int nn = 0; nn < num_ant_ele; nn++) {
        const float loc = (float)(norm_spacing * 0.5 * (num_ant_ele - 1 - 2 * nn));
        const float ph = k * loc;
        v[nn] = make_float2(cosf(ph), sinf(ph));
    }
    if (h->d_pilot.reserve(sizeof(float2) * num_ant_ele) != DOA_OK ||
        hipMemcpy(h->d_pilot.p, v.data(), sizeof(float2) * num_ant_ele, hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        if (!*doa_last_error()) doa::set_error("calibrate_lin_array: device setup failed");
        doa_calibrate_lin_array_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_calibrate_lin_array_destroy(doa_calibrate_lin_array_t *h)
{
    if (!h) return;
    h->d_pilot.release(); h->d_in.release(); h->d_out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int doa_calibrate_lin_array_work_dev(doa_calibrate_lin_array_t *h, int noutput_items, const void *d_input_items0,
                                     void *d_output_items0, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!d_input_items0 || !d_output_items0))) {
        doa::set_error("calibrate_lin_array_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    int rc = doa::launch_calibrate(h->N, noutput_items, d_input_items0, h->d_pilot.p, d_output_items0, h->bits,
                                   static_cast<hipStream_t>(hip_stream));
    return rc == DOA_OK ? noutput_items : rc;
}

int doa_calibrate_lin_array_work(doa_calibrate_lin_array_t *h, int noutput_items, const void *input_items0,
                                 void *output_items0)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || (noutput_items > 0 && (!input_items0 || !output_items0))) {
        doa::set_error("calibrate_lin_array_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    const size_t in_bytes = (size_t)noutput_items * h->N * h->N * sizeof(float2);
    const size_t out_bytes = (size_t)noutput_items * h->N * sizeof(float2);
    int rc = h->d_in.reserve(in_bytes);
    if (rc == DOA_OK) rc = h->d_out.reserve(out_bytes);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(h->d_in.p, input_items0, in_bytes, hipMemcpyHostToDevice, h->stream));
    rc = doa_calibrate_lin_array_work_dev(h, noutput_items, h->d_in.p, h->d_out.p, h->stream);
    if (rc < 0) return rc;
    DOA_HIP_TRY(hipMemcpyAsync(output_items0, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    DOA_HIP_TRY(hipStreamSynchronize(h->stream));
    return noutput_items;
}

int doa_calibrate_lin_array_set_internal_precision(doa_calibrate_lin_array_t *h, int bits)
{
    doa::clear_error();
    if (!h || (bits != 32 && bits != 64)) { doa::set_error("calibrate_lin_array_set_internal_precision: need a handle and bits = 32 or 64"); return DOA_ERR_INVALID_ARG; }
    h->bits = bits;
    return DOA_OK;
}

}  // extern "C"
