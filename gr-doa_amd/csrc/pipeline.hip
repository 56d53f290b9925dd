// pipeline.hip — autocorrelate -> MUSIC_lin_array -> find_local_max on device-resident streams.
//
// The batch entry point behind the benchmark: the three blocks exactly as
// apps/run_MUSIC_lin_array_simulation.grc wires them (autocorrelate(N, K, ovl, avg) ->
// MUSIC_lin_array(d, M, N, P) -> find_local_max(M, P, 0.0, 180.0)), as four launches on one
// stream: K1 covariance, K2+K3 EVD/projector/diagonal sums, K4 scan, K5 peak pick.  Intermediates
// (covariances, coefficient records, spectra) stay in HBM-resident buffers owned by the handle
// unless the caller asks for them.
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>

namespace doa {
bool find_local_max_fast_ok(int L, const void *d_in);
int launch_find_local_max_serial(const PeakTables &t, int n_items, const void *d_in, void *d_max, void *d_argmax,
                                 void *d_scratch, hipStream_t st);
}  // namespace doa

struct doa_music_pipeline {
    int N = 0, K = 0, ovl = 0, avg = 0;
    int max_batch = 0;
    int bits = 64;
    int device = 0;
    doa::MusicTables music;
    doa::PeakTables peaks;
    doa::DevBuf d_cov, d_coef, d_spec, d_scratch, d_gain;
    bool has_gain = false;
};

extern "C" {

doa_music_pipeline_t *doa_music_pipeline_create(int inputs, int snapshot_size, int overlap_size, int avg_method,
                                                float norm_spacing, int num_targets, int pspectrum_len, int max_batch)
{
    doa::clear_error();
    if (inputs <= 0 || inputs > DOA_MAX_ANT_ELE || snapshot_size <= 0 || overlap_size < 0 ||
        overlap_size >= snapshot_size) {
        doa::set_error("music_pipeline: bad autocorrelate parameters (inputs=%d snapshot=%d overlap=%d)", inputs,
                       snapshot_size, overlap_size);
        return nullptr;
    }
    if (num_targets <= 0 || num_targets >= inputs || num_targets > DOA_MAX_PEAKS || !(norm_spacing > 0.0f) ||
        norm_spacing > 0.5f || pspectrum_len <= 0 || max_batch <= 0) {
        doa::set_error("music_pipeline: bad MUSIC parameters (norm_spacing=%g num_targets=%d inputs=%d "
                       "pspectrum_len=%d max_batch=%d)", (double)norm_spacing, num_targets, inputs, pspectrum_len,
                       max_batch);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_music_pipeline();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->N = inputs; h->K = snapshot_size; h->ovl = overlap_size; h->avg = avg_method;
    h->max_batch = max_batch; h->device = dev;
    h->bits = doa::internal_precision_bits();
    int rc = h->music.build(norm_spacing, num_targets, inputs, pspectrum_len);
    if (rc == DOA_OK) rc = h->peaks.build(num_targets, pspectrum_len, 0.0f, 180.0f);
    if (rc == DOA_OK) rc = h->d_cov.reserve((size_t)max_batch * inputs * inputs * sizeof(float2));
    if (rc == DOA_OK) rc = h->d_coef.reserve((size_t)max_batch * doa::coef_stride(inputs) * sizeof(double));
    if (rc == DOA_OK) rc = h->d_spec.reserve((size_t)max_batch * pspectrum_len * sizeof(float));
    if (rc != DOA_OK) {
        doa_music_pipeline_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_music_pipeline_destroy(doa_music_pipeline_t *h)
{
    if (!h) return;
    h->music.release();
    h->peaks.release();
    h->d_cov.release(); h->d_coef.release(); h->d_spec.release(); h->d_scratch.release(); h->d_gain.release();
    delete h;
}

int doa_music_pipeline_fuse_antenna_correction(doa_music_pipeline_t *h, const float *gains_re_im)
{
    doa::clear_error();
    if (!h) return DOA_ERR_INVALID_ARG;
    if (!gains_re_im) { h->has_gain = false; return DOA_OK; }
    const int N = h->N;
    float2 w[DOA_MAX_ANT_ELE * DOA_MAX_ANT_ELE];
    for (int b = 0; b < N; b++)
        for (int a = 0; a < N; a++) {
            const float ar = gains_re_im[2 * a], ai = gains_re_im[2 * a + 1], br = gains_re_im[2 * b], bi = gains_re_im[2 * b + 1];
            w[a + b * N] = make_float2(ar * br + ai * bi, ai * br - ar * bi);   // g_a conj(g_b)
        }
    int rc = h->d_gain.reserve(sizeof(float2) * N * N);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(h->d_gain.p, w, sizeof(float2) * N * N, hipMemcpyHostToDevice));
    h->has_gain = true;
    return DOA_OK;
}

int doa_music_pipeline_work_dev(doa_music_pipeline_t *h, int noutput_items, const void *const *d_input_items,
                                void *d_cov_out, void *d_spectrum_out, void *d_max_out, void *d_argmax_out,
                                void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !d_input_items || (noutput_items > 0 && (!d_max_out || !d_argmax_out))) {
        doa::set_error("music_pipeline_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("music_pipeline_work_dev: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // DOA_PIPE_SKIP=cov,evd,scan: profiling aid that drops stages (outputs are then meaningless)
    unsigned skip = 0;      // read per call so that a profiling script can populate the intermediates first
    if (const char *e = getenv("DOA_PIPE_SKIP")) {
        if (strstr(e, "cov")) skip |= 1;
        if (strstr(e, "evd")) skip |= 2;
        if (strstr(e, "scan")) skip |= 4;
    }
    void *cov = d_cov_out ? d_cov_out : h->d_cov.p;
    void *spec = d_spectrum_out ? d_spectrum_out : h->d_spec.p;
    int rc = DOA_OK;
    if (!(skip & 1)) rc = doa::launch_autocorrelate(h->N, h->K, h->ovl, h->avg, noutput_items, d_input_items, cov, st,
                                                     h->has_gain ? h->d_gain.p : nullptr);
    if (rc != DOA_OK) return rc;
    const bool dbl = (h->bits == 64);
    if (!(skip & 2))
        rc = doa::launch_music_evd(h->N, h->music.M, noutput_items, cov, dbl ? nullptr : h->d_coef.p,
                                   dbl ? h->d_coef.p : nullptr, nullptr, h->bits, st);
    if (rc != DOA_OK) return rc;
    bool peaks_done = false;
    if (skip & 4) return noutput_items;
    rc = doa::launch_music_scan(h->music, h->bits, noutput_items, h->d_coef.p, spec, nullptr, st, &h->peaks, d_max_out,
                                d_argmax_out, &peaks_done);
    if (rc != DOA_OK) return rc;
    if (peaks_done) return noutput_items;
    if (doa::find_local_max_fast_ok(h->peaks.L, spec)) {
        rc = doa::launch_find_local_max(h->peaks, noutput_items, spec, d_max_out, d_argmax_out, st);
    } else {
        rc = h->d_scratch.reserve((size_t)h->max_batch * h->peaks.L);
        if (rc == DOA_OK)
            rc = doa::launch_find_local_max_serial(h->peaks, noutput_items, spec, d_max_out, d_argmax_out,
                                                   h->d_scratch.p, st);
    }
    return rc == DOA_OK ? noutput_items : rc;
}

}  // extern "C"
