// root_pipeline.hip — autocorrelate -> rootMUSIC_linear_array on device-resident streams: the Root-MUSIC branch of the hot
// path as ONE handle, the chain apps/run_RootMUSIC_lin_array_simulation.grc wires (autocorrelate(N, K, ovl, avg) ->
// rootMUSIC_linear_array(d, M, N); reference work being chained: lib/autocorrelate_impl.cc:83-118 ->
// lib/rootMUSIC_linear_array_impl.cc:90-152).  Three launches per batch -- K1 covariance, K2+K3 EVD / projector / diagonal
// sums (double records), K6 roots + selection -- with the same entry points, lanes, detached form and error contract as
// music_pipeline (pipeline.hip, pipeline_lanes.hpp).  Until round 4 this branch had to be driven as two block handles and
// two calls per batch, which left it host-bound above ~20 us per 4096-snapshot step.
#include "kernels.hpp"
#include "pipeline_lanes.hpp"

#include <cstdlib>
#include <cstring>

struct doa_root_pipeline {
    int N = 0, K = 0, ovl = 0, avg = 0, M = 0;
    float norm_spacing = 0.f;
    int max_batch = 0;
    int bits = 64;
    int device = 0;
    doa::DevBuf d_cov, d_coef, d_status, d_gain;
    bool has_gain = false;
    // host-pointer entry point only: two copy/compute lanes
    hipStream_t hst[2] = {nullptr, nullptr};
    doa::DevBuf d_in[2], d_res;
    doa::DevBuf d_work[2];
    doa::PinnedBuf h_stage, h_status;
    int fail_chunk = -1;
    enum { kCoef = 0, kCov, kStatus, kWork };
    doa::PipeLanes lanes;
};

namespace {
struct RootWs {
    void *coef;       // double coefficient records of the chain's items
    void *status;     // one int per item (1 = no root strictly inside the unit circle), or the caller's buffer
    void *work;       // K1's piece sums (overlapping windows), or NULL
};
size_t coef_bytes(const doa_root_pipeline *h, size_t items) { return items * doa::coef_stride(h->N) * sizeof(double); }

// K1 -> EVD -> roots for n items on `st`
int run_chain(doa_root_pipeline *h, int n, const void *const *d_in, void *cov, void *angles, const RootWs &ws, hipStream_t st)
{
    int rc = doa::launch_autocorrelate(h->N, h->K, h->ovl, h->avg, n, d_in, cov, st, h->has_gain ? h->d_gain.p : nullptr, ws.work);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_music_evd(h->N, h->M, n, cov, nullptr, ws.coef, nullptr, h->bits, st);
    if (rc != DOA_OK) return rc;
    rc = doa::launch_root_music(h->N, h->M, h->norm_spacing, n, ws.coef, angles, ws.status, st);
    return rc == DOA_OK ? n : rc;
}
// first item whose status word is set, or -1
int first_flagged(const int *status, int n)
{
    for (int i = 0; i < n; i++)
        if (status[i] != 0) return i;
    return -1;
}
int numeric_error(int item)
{
    doa::set_error("root_pipeline: item %d has no root strictly inside the unit circle (the reference raises in "
                   "arma::index_min here, lib/rootMUSIC_linear_array_impl.cc:129)", item);
    return DOA_ERR_NUMERIC;
}
}  // namespace

extern "C" {

doa_root_pipeline_t *doa_root_pipeline_create(int inputs, int snapshot_size, int overlap_size, int avg_method,
                                              float norm_spacing, int num_targets, int max_batch)
{
    doa::clear_error();
    if (inputs <= 0 || inputs > DOA_MAX_ANT_ELE || snapshot_size <= 0 || overlap_size < 0 || overlap_size >= snapshot_size) {
        doa::set_error("root_pipeline: bad autocorrelate parameters (inputs=%d snapshot=%d overlap=%d)", inputs, snapshot_size,
                       overlap_size);
        return nullptr;
    }
    if (inputs < 2 || num_targets <= 0 || num_targets >= inputs || num_targets > DOA_MAX_PEAKS || !(norm_spacing > 0.0f) ||
        norm_spacing > 0.5f || max_batch <= 0) {
        doa::set_error("root_pipeline: bad Root-MUSIC parameters (norm_spacing=%g num_targets=%d inputs=%d max_batch=%d)",
                       (double)norm_spacing, num_targets, inputs, max_batch);
        return nullptr;
    }
    int dev = 0;
    if (doa::ensure_device(&dev) != DOA_OK) return nullptr;
    auto *h = new (std::nothrow) doa_root_pipeline();
    if (!h) { doa::set_error("out of memory"); return nullptr; }
    h->N = inputs; h->K = snapshot_size; h->ovl = overlap_size; h->avg = avg_method; h->M = num_targets;
    h->norm_spacing = norm_spacing; h->max_batch = max_batch; h->device = dev;
    h->bits = doa::internal_precision_bits();
    int rc = h->d_cov.reserve((size_t)max_batch * inputs * inputs * sizeof(float2));
    if (rc == DOA_OK) rc = h->d_coef.reserve(coef_bytes(h, (size_t)max_batch));
    if (rc == DOA_OK) rc = h->d_status.reserve((size_t)max_batch * sizeof(int));
    if (const size_t ws = doa::autocorrelate_workspace_bytes(inputs, snapshot_size, overlap_size, max_batch); ws && rc == DOA_OK)
        rc = h->d_work[0].reserve(ws);
    if (rc != DOA_OK) {
        doa_root_pipeline_destroy(h);
        return nullptr;
    }
    return h;
}

void doa_root_pipeline_destroy(doa_root_pipeline_t *h)
{
    if (!h) return;
    h->d_cov.release(); h->d_coef.release(); h->d_status.release(); h->d_gain.release(); h->d_res.release();
    h->h_stage.release(); h->h_status.release();
    for (auto &b : h->d_work) b.release();
    for (auto &b : h->d_in) b.release();
    for (auto st : h->hst)
        if (st) (void)hipStreamDestroy(st);
    h->lanes.release();
    delete h;
}

int doa_root_pipeline_fuse_antenna_correction(doa_root_pipeline_t *h, const float *gains_re_im)
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
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    int rc = h->d_gain.reserve(sizeof(float2) * N * N);
    if (rc != DOA_OK) return rc;
    DOA_HIP_TRY(hipMemcpy(h->d_gain.p, w, sizeof(float2) * N * N, hipMemcpyHostToDevice));
    h->has_gain = true;
    return DOA_OK;
}

int doa_root_pipeline_work_dev(doa_root_pipeline_t *h, int noutput_items, const void *const *d_input_items, void *d_cov_out,
                               void *d_angles_out, int *d_status_out, void *hip_stream)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !d_input_items || (noutput_items > 0 && !d_angles_out)) {
        doa::set_error("root_pipeline_work_dev: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("root_pipeline_work_dev: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    RootWs ws{h->d_coef.p, d_status_out ? (void *)d_status_out : h->d_status.p, h->d_work[0].p};
    return run_chain(h, noutput_items, d_input_items, d_cov_out ? d_cov_out : h->d_cov.p, d_angles_out, ws,
                     static_cast<hipStream_t>(hip_stream));
}

int doa_root_pipeline_set_lanes(doa_root_pipeline_t *h, int n_lanes)
{
    doa::clear_error();
    if (!h || h->lanes.set_count(n_lanes) != DOA_OK) {
        doa::set_error("root_pipeline_set_lanes: need 1 <= n_lanes <= %d", doa::PipeLanes::kMaxLanes);
        return DOA_ERR_INVALID_ARG;
    }
    return DOA_OK;
}

int doa_root_pipeline_set_lane_streams(doa_root_pipeline_t *h, int n_lanes, void *const *hip_streams)
{
    doa::clear_error();
    if (!h || n_lanes < 1 || n_lanes > doa::PipeLanes::kMaxLanes || !hip_streams) {
        doa::set_error("root_pipeline_set_lane_streams: need 1 <= n_lanes <= %d and the streams", doa::PipeLanes::kMaxLanes);
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    return h->lanes.adopt(n_lanes, hip_streams);
}

int doa_root_pipeline_synchronize(doa_root_pipeline_t *h)
{
    doa::clear_error();
    if (!h) { doa::set_error("root_pipeline_synchronize: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    return h->lanes.synchronize();
}

int doa_root_pipeline_work_dev_batches(doa_root_pipeline_t *h, int n_batches, int noutput_items, const void *const *d_input_items,
                                       void *const *d_cov_out, void *const *d_angles_out, int *const *d_status_out,
                                       void *hip_stream)
{
    doa::clear_error();
    if (!h || n_batches < 0 || noutput_items < 0 || !d_input_items || !d_angles_out) {
        doa::set_error("root_pipeline_work_dev_batches: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("root_pipeline_work_dev_batches: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    for (int b = 0; b < n_batches; b++)
        if (noutput_items > 0 && !d_angles_out[b]) {
            doa::set_error("root_pipeline_work_dev_batches: batch %d has no angle output pointer", b);
            return DOA_ERR_INVALID_ARG;
        }
    if (n_batches == 0 || noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int N = h->N;
    const int fail_at = h->lanes.fail_batch;
    h->fail_chunk = -1;
    if (h->lanes.n_lanes == 1 && hip_stream != DOA_STREAM_DETACHED) {     // nothing to overlap: the caller's stream itself, no events
        hipStream_t caller = static_cast<hipStream_t>(hip_stream);
        h->lanes.fail_batch = -1;
        for (int b = 0; b < n_batches; b++) {
            if (fail_at == b) {
                doa::set_error("root_pipeline_work_dev_batches: injected failure in batch %d", b);
                (void)hipStreamSynchronize(caller);
                return DOA_ERR_HIP;
            }
            RootWs ws{h->d_coef.p, (d_status_out && d_status_out[b]) ? (void *)d_status_out[b] : h->d_status.p, h->d_work[0].p};
            const int rc = run_chain(h, noutput_items, d_input_items + (size_t)b * N,
                                     (d_cov_out && d_cov_out[b]) ? d_cov_out[b] : h->d_cov.p, d_angles_out[b], ws, caller);
            if (rc < 0) { (void)hipStreamSynchronize(caller); return rc; }
        }
        return n_batches * noutput_items;
    }
    bool need_cov = !d_cov_out, need_status = !d_status_out;
    for (int b = 0; b < n_batches && !(need_cov && need_status); b++) {
        if (d_cov_out && !d_cov_out[b]) need_cov = true;
        if (d_status_out && !d_status_out[b]) need_status = true;
    }
    const size_t work_bytes = doa::autocorrelate_workspace_bytes(N, h->K, h->ovl, h->max_batch);
    using H = doa_root_pipeline;
    auto prepare = [&](doa::PipeLane &ln) -> int {
        int rc = ln.buf[H::kCoef].reserve(coef_bytes(h, (size_t)h->max_batch));
        if (rc == DOA_OK && need_cov) rc = ln.buf[H::kCov].reserve((size_t)h->max_batch * N * N * sizeof(float2));
        if (rc == DOA_OK && need_status) rc = ln.buf[H::kStatus].reserve((size_t)h->max_batch * sizeof(int));
        if (rc == DOA_OK && work_bytes) rc = ln.buf[H::kWork].reserve(work_bytes);
        return rc;
    };
    auto launch = [&](int b, doa::PipeLane &ln) -> int {
        RootWs ws{ln.buf[H::kCoef].p, (d_status_out && d_status_out[b]) ? (void *)d_status_out[b] : ln.buf[H::kStatus].p,
                  ln.buf[H::kWork].p};
        return run_chain(h, noutput_items, d_input_items + (size_t)b * N,
                         (d_cov_out && d_cov_out[b]) ? d_cov_out[b] : ln.buf[H::kCov].p, d_angles_out[b], ws, ln.st);
    };
    const int rc = h->lanes.run_batches("root_pipeline_work_dev_batches", n_batches, hip_stream, prepare, launch);
    return rc < 0 ? rc : n_batches * noutput_items;
}

int doa_root_pipeline_work(doa_root_pipeline_t *h, int noutput_items, const void *const *input_items, void *cov_out,
                           void *angles_out)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !input_items || (noutput_items > 0 && !angles_out)) {
        doa::set_error("root_pipeline_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("root_pipeline_work: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    const int N = h->N, M = h->M;
    for (int k = 0; k < N; k++)
        if (!input_items[k]) { doa::set_error("root_pipeline_work: input_items[%d] is NULL", k); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    if (const int src = doa::ensure_stream_pair(h->hst); src != DOA_OK) return src;
    const size_t nonoverlap = (size_t)(h->K - h->ovl);
    const size_t n_all = (size_t)noutput_items;
    // result block on the device: [angles | status] per call, sections 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t ang_b = n_all * M * sizeof(float), st_b = n_all * sizeof(int);
    // Scheduler-sized calls: one page-locked staging buffer, one copy each way (as music_pipeline; pipeline.hip has the
    // measurements)
    {
        static const size_t kSmallCallBytes = [] { const char *e = getenv("DOA_PIPE_SMALL_CALL_KB"); return (size_t)(e ? atoi(e) : 2048) << 10; }();
        const size_t span = (n_all - 1) * nonoverlap + h->K;
        const size_t span_al = (span + 1) & ~(size_t)1;
        const size_t in_bytes = span_al * N * sizeof(float2);
        const size_t cov_b = cov_out ? n_all * N * N * sizeof(float2) : 0;
        const size_t off_st = up(ang_b), off_cov = off_st + up(st_b);
        const size_t out_bytes = off_cov + cov_b;
        if (in_bytes <= kSmallCallBytes && out_bytes <= kSmallCallBytes) {
            int rc = h->h_stage.reserve(in_bytes > out_bytes ? in_bytes : out_bytes);
            if (rc == DOA_OK) rc = h->d_in[0].reserve(in_bytes);
            if (rc == DOA_OK) rc = h->d_res.reserve(out_bytes);
            if (rc != DOA_OK) return rc;
            hipStream_t st = h->hst[0];
            char *hs = h->h_stage.as<char>();
            const void *d_ptrs[DOA_MAX_ANT_ELE];
            for (int k = 0; k < N; k++) {
                memcpy(hs + (size_t)k * span_al * sizeof(float2), input_items[k], span * sizeof(float2));
                d_ptrs[k] = h->d_in[0].as<float2>() + (size_t)k * span_al;
            }
            // from the upload on, every exit synchronises the stream first (a copy still reading the staging buffer would
            // race the next call's memcpy into it)
            auto staged = [&]() -> int {
                DOA_HIP_TRY(hipMemcpyAsync(h->d_in[0].p, hs, in_bytes, hipMemcpyHostToDevice, st));
                char *dr = h->d_res.as<char>();
                if (h->fail_chunk == 0) { doa::set_error("root_pipeline_work: injected failure"); return DOA_ERR_HIP; }
                RootWs ws{h->d_coef.p, dr + off_st, h->d_work[0].p};
                const int rr = run_chain(h, noutput_items, d_ptrs, cov_out ? (void *)(dr + off_cov) : h->d_cov.p, dr, ws, st);
                if (rr < 0) return rr;
                DOA_HIP_TRY(hipMemcpyAsync(hs, dr, out_bytes, hipMemcpyDeviceToHost, st));
                return DOA_OK;
            };
            rc = staged();
            h->fail_chunk = -1; h->lanes.fail_batch = -1;
            const hipError_t se = hipStreamSynchronize(st);
            if (rc < 0) return rc;
            if (se != hipSuccess) { doa::set_error("root_pipeline_work: %s", hipGetErrorString(se)); return DOA_ERR_HIP; }
            memcpy(angles_out, hs, ang_b);
            if (cov_out) memcpy(cov_out, hs + off_cov, cov_b);
            if (const int bad = first_flagged(reinterpret_cast<const int *>(hs + off_st), noutput_items); bad >= 0) return numeric_error(bad);
            return noutput_items;
        }
    }
    // chunks of ~32 MiB of new samples alternating over two streams (as music_pipeline)
    size_t chunk = ((size_t)32 << 20) / (nonoverlap * N * sizeof(float2));
    chunk = chunk < 1 ? 1 : (chunk > n_all ? n_all : chunk);
    const size_t span_max = (chunk - 1) * nonoverlap + h->K;
    const size_t span_al = doa::stream_stride_bytes(span_max * sizeof(float2)) / sizeof(float2);
    int rc = h->d_res.reserve(up(ang_b) + st_b);
    if (rc == DOA_OK) rc = h->h_status.reserve(st_b);
    for (auto &b : h->d_in)
        if (rc == DOA_OK) rc = b.reserve(span_al * N * sizeof(float2));
    if (const size_t ws = doa::autocorrelate_workspace_bytes(N, h->K, h->ovl, (int)chunk); ws && rc == DOA_OK)
        rc = h->d_work[1].reserve(ws);
    if (rc != DOA_OK) return rc;
    float *d_ang = h->d_res.as<float>();
    int *d_st = reinterpret_cast<int *>(h->d_res.as<char>() + up(ang_b));
    int lane = 0, chunk_index = 0;
    auto enqueue_chunk = [&](size_t s0, size_t n, hipStream_t st) -> int {
        const size_t span = (n - 1) * nonoverlap + h->K;
        const void *d_ptrs[DOA_MAX_ANT_ELE];
        for (int k = 0; k < N; k++) {
            float2 *dst = h->d_in[lane].as<float2>() + k * span_al;
            const float2 *src = static_cast<const float2 *>(input_items[k]) + s0 * nonoverlap;
            DOA_HIP_TRY(hipMemcpyAsync(dst, src, span * sizeof(float2), hipMemcpyHostToDevice, st));
            d_ptrs[k] = dst;
        }
        if (h->fail_chunk == chunk_index) { doa::set_error("root_pipeline_work: injected failure in chunk %d", chunk_index); return DOA_ERR_HIP; }
        float2 *cov = h->d_cov.as<float2>() + s0 * N * N;
        RootWs ws{static_cast<char *>(h->d_coef.p) + coef_bytes(h, s0), d_st + s0, h->d_work[lane].p};
        const int rr = run_chain(h, (int)n, d_ptrs, cov, d_ang + s0 * M, ws, st);
        if (rr < 0) return rr;
        if (cov_out)
            DOA_HIP_TRY(hipMemcpyAsync(static_cast<float2 *>(cov_out) + s0 * N * N, cov, n * N * N * sizeof(float2),
                                       hipMemcpyDeviceToHost, st));
        DOA_HIP_TRY(hipMemcpyAsync(static_cast<float *>(angles_out) + s0 * M, d_ang + s0 * M, n * M * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        DOA_HIP_TRY(hipMemcpyAsync(h->h_status.as<int>() + s0, d_st + s0, n * sizeof(int), hipMemcpyDeviceToHost, st));
        return DOA_OK;
    };
    for (size_t s0 = 0; s0 < n_all; s0 += chunk, lane ^= 1, chunk_index++) {
        const size_t n = (n_all - s0 < chunk) ? n_all - s0 : chunk;
        rc = enqueue_chunk(s0, n, h->hst[lane]);
        if (rc < 0) break;
    }
    h->fail_chunk = -1; h->lanes.fail_batch = -1;
    // the loop's exit -- normal or not -- synchronises BOTH lanes: caller-owned host buffers must be quiet when this returns
    for (auto st : h->hst) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && rc >= 0) { doa::set_error("root_pipeline_work: %s", hipGetErrorString(e)); rc = DOA_ERR_HIP; }
    }
    if (rc < 0) return rc;
    if (const int bad = first_flagged(h->h_status.as<int>(), noutput_items); bad >= 0) return numeric_error(bad);
    return noutput_items;
}

int doa_root_pipeline_inject_failure(doa_root_pipeline_t *h, int chunk_index)
{
    doa::clear_error();
    if (!h || chunk_index < -1) { doa::set_error("root_pipeline_inject_failure: bad arguments"); return DOA_ERR_INVALID_ARG; }
    h->fail_chunk = chunk_index;
    h->lanes.fail_batch = chunk_index;
    return DOA_OK;
}

int doa_root_pipeline_lanes_idle(doa_root_pipeline_t *h)
{
    doa::clear_error();
    if (!h) { doa::set_error("root_pipeline_lanes_idle: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    for (auto st : h->hst)
        if (st && hipStreamQuery(st) != hipSuccess) return 0;
    return h->lanes.idle() ? 1 : 0;
}

int doa_root_pipeline_set_internal_precision(doa_root_pipeline_t *h, int bits)
{
    doa::clear_error();
    if (!h || (bits != 32 && bits != 64)) { doa::set_error("root_pipeline_set_internal_precision: need a handle and bits = 32 or 64"); return DOA_ERR_INVALID_ARG; }
    h->bits = bits;
    return DOA_OK;
}

}  // extern "C"
