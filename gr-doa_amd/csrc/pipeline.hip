// pipeline.hip — autocorrelate -> MUSIC_lin_array -> find_local_max on device-resident streams.
//
// The batch entry point behind the benchmark: the three blocks exactly as
// apps/run_MUSIC_lin_array_simulation.grc wires them (autocorrelate(N, K, ovl, avg) ->
// MUSIC_lin_array(d, M, N, P) -> find_local_max(M, P, 0.0, 180.0)), as four launches on one
// stream: K1 covariance, K2+K3 EVD/projector/diagonal sums, K4 scan, K5 peak pick.  Intermediates
// (covariances, coefficient records, spectra) stay in HBM-resident buffers owned by the handle
// unless the caller asks for them.
#include "kernels.hpp"
#include "pipeline_lanes.hpp"

#include <cstdlib>
#include <cstring>
#include <string>

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
    unsigned stages = 7;           // bit 0 K1, bit 1 K2+K3, bit 2 K4+K5 (doa_music_pipeline_set_stages)
    doa::MusicTables music;
    doa::PeakTables peaks;
    doa::DevBuf d_cov, d_coef, d_cheb, d_spec, d_scratch, d_gain;
    bool has_gain = false;
    // host-pointer entry point only: two copy/compute lanes
    hipStream_t hst[2] = {nullptr, nullptr};
    doa::DevBuf d_in[2], d_res;
    doa::DevBuf d_work[2];          // K1's piece sums (overlapping windows), one per copy/compute lane
    doa::PinnedBuf h_stage;         // scheduler-sized calls: one page-locked staging buffer, one copy each way
    int fail_chunk = -1;            // doa_music_pipeline_inject_failure, host-pointer entry: one-shot, cleared by every call (tests)
    // doa_music_pipeline_work_dev_batches: the library's own overlap lanes (pipeline_lanes.hpp); a lane's buffers:
    enum { kCoef = 0, kCheb, kCov, kSpec, kWork, kScratch };
    doa::PipeLanes lanes;
};

// what one K1 -> EVD -> scan chain needs besides its inputs and outputs
struct PipeWs {
    void *coef;             // coefficient records of the chain's items
    void *cheb;             // N <= 4, double: their pre-transformed twins for the lean scan kernel (else NULL)
    void *spec_scratch;     // P floats per item, used when the caller does not want the spectrum
    void *work;             // K1's piece sums (overlapping windows), or NULL
    doa::DevBuf *scratch;   // grown on demand: the serial peak pick of unusual vector lengths
    size_t scratch_item_off;
};

// K1 -> EVD -> scan (+ peak) for n items on `st` with the workspace `ws`.
// `spec` == nullptr: nobody wants the spectrum (angles-only call): ws.spec_scratch serves as scratch and the lean
// scan kernel neither converts the row to dB nor writes it.
static int run_k1(doa_music_pipeline *h, int n, const void *const *d_in, void *cov, const PipeWs &ws, hipStream_t st)
{
    if (~h->stages & 1u) return DOA_OK;         // doa_music_pipeline_set_stages (profiling aid; all stages in production)
    return doa::launch_autocorrelate(h->N, h->K, h->ovl, h->avg, n, d_in, cov, st, h->has_gain ? h->d_gain.p : nullptr, ws.work);
}
static int run_evd_scan(doa_music_pipeline *h, int n, void *cov, void *spec, void *mx, void *am, const PipeWs &ws, hipStream_t st)
{
    const bool store_spec = (spec != nullptr);
    if (!spec) spec = ws.spec_scratch;
    const unsigned skip = ~h->stages & 7u;
    int rc = DOA_OK;
    const bool dbl = (h->bits == 64);
    void *coef = ws.coef;
    if (!(skip & 2))
        rc = doa::launch_music_evd(h->N, h->music.M, n, cov, dbl ? nullptr : coef, dbl ? coef : nullptr, nullptr, h->bits, st, ws.cheb);
    if (rc != DOA_OK) return rc;
    bool peaks_done = false;
    if (skip & 4) return n;
    rc = doa::launch_music_scan(h->music, h->bits, n, coef, spec, nullptr, st, &h->peaks, mx, am, &peaks_done, store_spec, ws.cheb);
    if (rc != DOA_OK) return rc;
    if (peaks_done) return n;
    if (doa::find_local_max_fast_ok(h->peaks.L, spec)) {
        rc = doa::launch_find_local_max(h->peaks, n, spec, mx, am, st);
    } else {
        rc = ws.scratch->reserve((size_t)h->max_batch * h->peaks.L);
        if (rc == DOA_OK)
            rc = doa::launch_find_local_max_serial(h->peaks, n, spec, mx, am,
                                                   static_cast<char *>(ws.scratch->p) + ws.scratch_item_off * h->peaks.L, st);
    }
    return rc == DOA_OK ? n : rc;
}
static int run_ws(doa_music_pipeline *h, int n, const void *const *d_in, void *cov, void *spec, void *mx, void *am,
                  const PipeWs &ws, hipStream_t st)
{
    const int rc = run_k1(h, n, d_in, cov, ws, st);
    return rc != DOA_OK ? rc : run_evd_scan(h, n, cov, spec, mx, am, ws, st);
}

// the handle's own single workspace (work_dev and the host-pointer entry): records / scratch rows at item offset
// `item_off` (chunks in flight on different streams must not share them)
static int run_dev(doa_music_pipeline *h, int n, const void *const *d_in, void *cov, void *spec, void *mx, void *am,
                   size_t item_off, hipStream_t st, int lane = 0)
{
    const bool dbl = (h->bits == 64);
    PipeWs ws;
    ws.coef = static_cast<char *>(h->d_coef.p) + item_off * doa::coef_stride(h->N) * (dbl ? sizeof(double) : sizeof(float));
    ws.cheb = h->d_cheb.p ? static_cast<char *>(h->d_cheb.p) + item_off * doa::kChebRecord * sizeof(double) : nullptr;
    ws.spec_scratch = static_cast<char *>(h->d_spec.p) + item_off * h->peaks.L * sizeof(float);
    ws.work = h->d_work[lane].p;
    ws.scratch = &h->d_scratch;
    ws.scratch_item_off = item_off;
    return run_ws(h, n, d_in, cov, spec, mx, am, ws, st);
}

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
    if (rc == DOA_OK && doa::music_uses_cheb(inputs, h->bits)) rc = h->d_cheb.reserve((size_t)max_batch * doa::kChebRecord * sizeof(double));
    if (rc == DOA_OK) rc = h->d_spec.reserve((size_t)max_batch * pspectrum_len * sizeof(float));
    if (const size_t ws = doa::autocorrelate_workspace_bytes(inputs, snapshot_size, overlap_size, max_batch); ws && rc == DOA_OK)
        rc = h->d_work[0].reserve(ws);
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
    h->d_cov.release(); h->d_coef.release(); h->d_cheb.release(); h->d_spec.release(); h->d_scratch.release(); h->d_gain.release();
    h->d_res.release(); h->h_stage.release();
    for (auto &b : h->d_work) b.release();
    for (auto &b : h->d_in) b.release();
    for (auto st : h->hst)
        if (st) (void)hipStreamDestroy(st);
    h->lanes.release();
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

int doa_music_pipeline_set_stages(doa_music_pipeline_t *h, int stage_mask)
{
    doa::clear_error();
    if (!h || stage_mask < 0 || stage_mask > 7) { doa::set_error("music_pipeline_set_stages: bad arguments"); return DOA_ERR_INVALID_ARG; }
    h->stages = (unsigned)stage_mask;
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
    void *cov = d_cov_out ? d_cov_out : h->d_cov.p;
    void *spec = d_spectrum_out;                       // NULL = angles only (run_dev)
    return run_dev(h, noutput_items, d_input_items, cov, spec, d_max_out, d_argmax_out, 0,
                   static_cast<hipStream_t>(hip_stream));
}

int doa_music_pipeline_set_lanes(doa_music_pipeline_t *h, int n_lanes)
{
    doa::clear_error();
    if (!h || h->lanes.set_count(n_lanes) != DOA_OK) {
        doa::set_error("music_pipeline_set_lanes: need 1 <= n_lanes <= %d", doa::PipeLanes::kMaxLanes);
        return DOA_ERR_INVALID_ARG;
    }
    return DOA_OK;
}

int doa_music_pipeline_work_dev_batches(doa_music_pipeline_t *h, int n_batches, int noutput_items,
                                        const void *const *d_input_items, void *const *d_cov_out,
                                        void *const *d_spectrum_out, void *const *d_max_out, void *const *d_argmax_out,
                                        void *hip_stream)
{
    doa::clear_error();
    if (!h || n_batches < 0 || noutput_items < 0 || !d_input_items || !d_max_out || !d_argmax_out) {
        doa::set_error("music_pipeline_work_dev_batches: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("music_pipeline_work_dev_batches: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    for (int b = 0; b < n_batches; b++)
        if (noutput_items > 0 && (!d_max_out[b] || !d_argmax_out[b])) {
            doa::set_error("music_pipeline_work_dev_batches: batch %d has no peak output pointers", b);
            return DOA_ERR_INVALID_ARG;
        }
    if (n_batches == 0 || noutput_items == 0) return 0;
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    const int N = h->N;
    const int fail_at = h->lanes.fail_batch;
    h->fail_chunk = -1;                                          // (the test aid is one-shot for whichever entry comes next)
    if (h->lanes.n_lanes == 1 && hip_stream != DOA_STREAM_DETACHED) {     // nothing to overlap: the caller's stream itself, no events
        hipStream_t caller = static_cast<hipStream_t>(hip_stream);
        h->lanes.fail_batch = -1;
        for (int b = 0; b < n_batches; b++) {
            if (fail_at == b) {
                doa::set_error("music_pipeline_work_dev_batches: injected failure in batch %d", b);
                (void)hipStreamSynchronize(caller);              // the contract of an error return: nothing of the call still runs
                return DOA_ERR_HIP;
            }
            void *cov = (d_cov_out && d_cov_out[b]) ? d_cov_out[b] : h->d_cov.p;
            const int rc = run_dev(h, noutput_items, d_input_items + (size_t)b * N, cov, d_spectrum_out ? d_spectrum_out[b] : nullptr,
                                   d_max_out[b], d_argmax_out[b], 0, caller);
            if (rc < 0) { (void)hipStreamSynchronize(caller); return rc; }
        }
        return n_batches * noutput_items;
    }
    const bool dbl = (h->bits == 64);
    bool need_cov = !d_cov_out, need_spec = !d_spectrum_out;
    for (int b = 0; b < n_batches && !(need_cov && need_spec); b++) {
        if (d_cov_out && !d_cov_out[b]) need_cov = true;
        if (d_spectrum_out && !d_spectrum_out[b]) need_spec = true;
    }
    const size_t work_bytes = doa::autocorrelate_workspace_bytes(N, h->K, h->ovl, h->max_batch);
    using H = doa_music_pipeline;
    auto prepare = [&](doa::PipeLane &ln) -> int {
        int rc = ln.buf[H::kCoef].reserve((size_t)h->max_batch * doa::coef_stride(N) * (dbl ? sizeof(double) : sizeof(float)));
        if (rc == DOA_OK && doa::music_uses_cheb(N, h->bits)) rc = ln.buf[H::kCheb].reserve((size_t)h->max_batch * doa::kChebRecord * sizeof(double));
        if (rc == DOA_OK && need_cov) rc = ln.buf[H::kCov].reserve((size_t)h->max_batch * N * N * sizeof(float2));
        if (rc == DOA_OK && need_spec) rc = ln.buf[H::kSpec].reserve((size_t)h->max_batch * h->peaks.L * sizeof(float));
        if (rc == DOA_OK && work_bytes) rc = ln.buf[H::kWork].reserve(work_bytes);
        return rc;
    };
    auto launch = [&](int b, doa::PipeLane &ln) -> int {
        PipeWs ws;
        ws.coef = ln.buf[H::kCoef].p; ws.cheb = ln.buf[H::kCheb].p; ws.spec_scratch = ln.buf[H::kSpec].p; ws.work = ln.buf[H::kWork].p;
        ws.scratch = &ln.buf[H::kScratch]; ws.scratch_item_off = 0;
        void *cov = (d_cov_out && d_cov_out[b]) ? d_cov_out[b] : ln.buf[H::kCov].p;
        return run_ws(h, noutput_items, d_input_items + (size_t)b * N, cov, d_spectrum_out ? d_spectrum_out[b] : nullptr, d_max_out[b],
                      d_argmax_out[b], ws, ln.st);
    };
    const int rc = h->lanes.run_batches("music_pipeline_work_dev_batches", n_batches, hip_stream, prepare, launch);
    return rc < 0 ? rc : n_batches * noutput_items;
}

int doa_music_pipeline_set_lane_streams(doa_music_pipeline_t *h, int n_lanes, void *const *hip_streams)
{
    doa::clear_error();
    if (!h || n_lanes < 1 || n_lanes > doa::PipeLanes::kMaxLanes || !hip_streams) {
        doa::set_error("music_pipeline_set_lane_streams: need 1 <= n_lanes <= %d and the streams", doa::PipeLanes::kMaxLanes);
        return DOA_ERR_INVALID_ARG;
    }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    return h->lanes.adopt(n_lanes, hip_streams);
}

int doa_music_pipeline_synchronize(doa_music_pipeline_t *h)
{
    doa::clear_error();
    if (!h) { doa::set_error("music_pipeline_synchronize: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    return h->lanes.synchronize();
}

int doa_music_pipeline_work(doa_music_pipeline_t *h, int noutput_items, const void *const *input_items, void *cov_out,
                            void *spectrum_out, void *max_out, void *argmax_out)
{
    doa::clear_error();
    if (!h || noutput_items < 0 || !input_items || (noutput_items > 0 && (!max_out || !argmax_out))) {
        doa::set_error("music_pipeline_work: bad arguments");
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items > h->max_batch) {
        doa::set_error("music_pipeline_work: noutput_items=%d exceeds max_batch=%d", noutput_items, h->max_batch);
        return DOA_ERR_INVALID_ARG;
    }
    if (noutput_items == 0) return 0;
    const int N = h->N, M = h->music.M, P = h->peaks.L;
    for (int k = 0; k < N; k++)
        if (!input_items[k]) { doa::set_error("music_pipeline_work: input_items[%d] is NULL", k); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    if (const int src = doa::ensure_stream_pair(h->hst); src != DOA_OK) return src;
    const size_t nonoverlap = (size_t)(h->K - h->ovl);
    // Scheduler-sized calls (a GNU Radio work() hands over a few to a few hundred items): the fixed costs are what
    // counts -- N pageable host-to-device copies, up to four copies back and two stream synchronisations.  Below
    // kSmallCallBytes of input the window of every stream is packed into ONE page-locked staging buffer (the same
    // bytes the driver's pageable path would stage, minus its per-copy set-up), crosses PCIe as ONE transfer, and the
    // angles (+ the optional covariances / spectra) come back as ONE transfer through the same buffer.  Measured: 8 items
    // 105 -> 53 us per call, 64 items 151 -> 116 us; from 8 MiB up the host-side packing costs more than it saves.
    {
        static const size_t kSmallCallBytes = [] { const char *e = getenv("DOA_PIPE_SMALL_CALL_KB"); return (size_t)(e ? atoi(e) : 2048) << 10; }();
        const size_t span = (size_t)(noutput_items - 1) * nonoverlap + h->K;
        const size_t span_al = (span + 1) & ~(size_t)1;
        const size_t in_bytes = span_al * N * sizeof(float2);
        const size_t n = (size_t)noutput_items;
        // sections of the result block [max | argmax | cov | spectrum], each on a 256-byte boundary (the scan kernels'
        // 16-byte stores need aligned spectrum rows)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t pk_b = n * M * sizeof(float);
        const size_t cov_b = cov_out ? n * N * N * sizeof(float2) : 0, spec_b = spectrum_out ? n * P * sizeof(float) : 0;
        const size_t off_am = up(pk_b), off_cov = off_am + up(pk_b), off_spec = off_cov + up(cov_b);
        const size_t out_bytes = off_spec + spec_b;
        // (the gate looks at BOTH directions: a short-snapshot, long-spectrum call returns n * P * 4 bytes, and staging
        // hundreds of MiB of spectra through the page-locked buffer -- plus a host memcpy each -- is what the chunked
        // path below exists to avoid)
        if (in_bytes <= kSmallCallBytes && out_bytes <= kSmallCallBytes) {
            const size_t res_min = (size_t)h->max_batch * M * 2 * sizeof(float);
            int rc = h->h_stage.reserve(in_bytes > out_bytes ? in_bytes : out_bytes);
            if (rc == DOA_OK) rc = h->d_in[0].reserve(in_bytes);
            if (rc == DOA_OK) rc = h->d_res.reserve(out_bytes > res_min ? out_bytes : res_min);
            if (rc != DOA_OK) return rc;
            hipStream_t st = h->hst[0];
            char *hs = h->h_stage.as<char>();
            const void *d_ptrs[DOA_MAX_ANT_ELE];
            for (int k = 0; k < N; k++) {
                memcpy(hs + (size_t)k * span_al * sizeof(float2), input_items[k], span * sizeof(float2));
                d_ptrs[k] = h->d_in[0].as<float2>() + (size_t)k * span_al;
            }
            // From the upload on, every exit synchronises the stream first: a copy still reading the staging buffer
            // would race the next call's memcpy into it.
            auto staged = [&]() -> int {
                DOA_HIP_TRY(hipMemcpyAsync(h->d_in[0].p, hs, in_bytes, hipMemcpyHostToDevice, st));
                char *dr = h->d_res.as<char>();
                float *d_mx = reinterpret_cast<float *>(dr), *d_am = reinterpret_cast<float *>(dr + off_am);
                void *d_cov = cov_out ? (void *)(dr + off_cov) : h->d_cov.p;
                void *d_spec = spectrum_out ? (void *)(dr + off_spec) : nullptr;            // NULL = angles only
                const int rr = (h->fail_chunk == 0) ? (doa::set_error("music_pipeline_work: injected failure"), DOA_ERR_HIP)
                                                    : run_dev(h, noutput_items, d_ptrs, d_cov, d_spec, d_mx, d_am, 0, st, 0);
                if (rr < 0) return rr;
                // (the staging buffer is free again: the upload was enqueued before the kernels on the same stream, and
                // the download below is ordered behind them)
                DOA_HIP_TRY(hipMemcpyAsync(hs, dr, out_bytes, hipMemcpyDeviceToHost, st));
                return DOA_OK;
            };
            rc = staged();
            h->fail_chunk = -1; h->lanes.fail_batch = -1;
            const hipError_t se = hipStreamSynchronize(st);
            if (rc < 0) return rc;
            if (se != hipSuccess) { doa::set_error("music_pipeline_work: %s", hipGetErrorString(se)); return DOA_ERR_HIP; }
            memcpy(max_out, hs, pk_b);
            memcpy(argmax_out, hs + off_am, pk_b);
            if (cov_out) memcpy(cov_out, hs + off_cov, cov_b);
            if (spectrum_out) memcpy(spectrum_out, hs + off_spec, spec_b);
            return noutput_items;
        }
    }
    // Chunks of ~32 MiB of new samples alternate over two streams: while one chunk's results travel back
    // the next chunk's samples travel in (PCIe is full duplex; the kernels themselves are ~1 % of a
    // chunk's transfer time).  The overlap needs page-locked caller buffers; pageable ones still work.
    size_t chunk = ((size_t)32 << 20) / (nonoverlap * N * sizeof(float2));
    chunk = chunk < 1 ? 1 : (chunk > (size_t)noutput_items ? (size_t)noutput_items : chunk);
    const size_t span_max = (chunk - 1) * nonoverlap + h->K;
    // distance between the device copies of the streams: 16-B aligned and staggered against the 8 KiB aliasing period
    const size_t span_al = doa::stream_stride_bytes(span_max * sizeof(float2)) / sizeof(float2);
    int rc = h->d_res.reserve((size_t)h->max_batch * M * 2 * sizeof(float));
    for (auto &b : h->d_in)
        if (rc == DOA_OK) rc = b.reserve(span_al * N * sizeof(float2));
    if (const size_t ws = doa::autocorrelate_workspace_bytes(N, h->K, h->ovl, (int)chunk); ws && rc == DOA_OK)
        rc = h->d_work[1].reserve(ws);                 // lane 0 uses the workspace create() sized for max_batch
    if (rc != DOA_OK) return rc;
    float *d_mx = h->d_res.as<float>(), *d_am = d_mx + (size_t)h->max_batch * M;
    int lane = 0, chunk_index = 0;
    // One chunk: uploads, kernels, downloads, all on its lane's stream.  A failure anywhere in it ends the loop, and the
    // loop's exit -- normal or not -- synchronises BOTH lanes: the other lane may still be copying to or from caller-owned
    // host buffers, which must be quiet before this call returns.
    auto enqueue_chunk = [&](size_t s0, size_t n, hipStream_t st) -> int {
        const size_t span = (n - 1) * nonoverlap + h->K;
        const void *d_ptrs[DOA_MAX_ANT_ELE];
        for (int k = 0; k < N; k++) {
            float2 *dst = h->d_in[lane].as<float2>() + k * span_al;
            const float2 *src = static_cast<const float2 *>(input_items[k]) + s0 * nonoverlap;
            DOA_HIP_TRY(hipMemcpyAsync(dst, src, span * sizeof(float2), hipMemcpyHostToDevice, st));
            d_ptrs[k] = dst;
        }
        if (h->fail_chunk == chunk_index) { doa::set_error("music_pipeline_work: injected failure in chunk %d", chunk_index); return DOA_ERR_HIP; }
        float2 *cov = h->d_cov.as<float2>() + s0 * N * N;
        float *spec = spectrum_out ? h->d_spec.as<float>() + s0 * P : nullptr;         // NULL = angles only
        const int rr = run_dev(h, (int)n, d_ptrs, cov, spec, d_mx + s0 * M, d_am + s0 * M, s0, st, lane);
        if (rr < 0) return rr;
        if (cov_out)
            DOA_HIP_TRY(hipMemcpyAsync(static_cast<float2 *>(cov_out) + s0 * N * N, cov, n * N * N * sizeof(float2),
                                       hipMemcpyDeviceToHost, st));
        if (spectrum_out)
            DOA_HIP_TRY(hipMemcpyAsync(static_cast<float *>(spectrum_out) + s0 * P, spec, n * P * sizeof(float),
                                       hipMemcpyDeviceToHost, st));
        DOA_HIP_TRY(hipMemcpyAsync(static_cast<float *>(max_out) + s0 * M, d_mx + s0 * M, n * M * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        DOA_HIP_TRY(hipMemcpyAsync(static_cast<float *>(argmax_out) + s0 * M, d_am + s0 * M, n * M * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        return DOA_OK;
    };
    for (size_t s0 = 0; s0 < (size_t)noutput_items; s0 += chunk, lane ^= 1, chunk_index++) {
        const size_t n = ((size_t)noutput_items - s0 < chunk) ? (size_t)noutput_items - s0 : chunk;
        rc = enqueue_chunk(s0, n, h->hst[lane]);
        if (rc < 0) break;
    }
    h->fail_chunk = -1; h->lanes.fail_batch = -1;
    for (auto st : h->hst) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && rc >= 0) { doa::set_error("music_pipeline_work: %s", hipGetErrorString(e)); rc = DOA_ERR_HIP; }
    }
    return rc < 0 ? rc : noutput_items;
}

int doa_music_pipeline_inject_failure(doa_music_pipeline_t *h, int chunk_index)
{
    doa::clear_error();
    if (!h || chunk_index < -1) { doa::set_error("music_pipeline_inject_failure: bad arguments"); return DOA_ERR_INVALID_ARG; }
    h->fail_chunk = chunk_index;            // whichever entry is called next consumes it: the host entry (chunk index) ...
    h->lanes.fail_batch = chunk_index;      // ... or the batches entry (batch index); both clear both
    return DOA_OK;
}

int doa_music_pipeline_lanes_idle(doa_music_pipeline_t *h)
{
    doa::clear_error();
    if (!h) { doa::set_error("music_pipeline_lanes_idle: bad arguments"); return DOA_ERR_INVALID_ARG; }
    if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
    for (auto st : h->hst)
        if (st && hipStreamQuery(st) != hipSuccess) return 0;
    return h->lanes.idle() ? 1 : 0;
}

int doa_music_pipeline_set_internal_precision(doa_music_pipeline_t *h, int bits)
{
    doa::clear_error();
    if (!h || (bits != 32 && bits != 64)) { doa::set_error("music_pipeline_set_internal_precision: need a handle and bits = 32 or 64"); return DOA_ERR_INVALID_ARG; }
    if (doa::music_uses_cheb(h->N, bits)) {
        if (int brc = doa::bind_device(h->device); brc != DOA_OK) return brc;
        const int rc = h->d_cheb.reserve((size_t)h->max_batch * doa::kChebRecord * sizeof(double));
        if (rc != DOA_OK) return rc;
    }
    h->bits = bits;
    return DOA_OK;
}

}  // extern "C"
