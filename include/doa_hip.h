/*
 * doa_hip.h — C ABI of libdoa_hip.so, the MI355X (gfx950) implementation of gr-doa's hot path
 *
 *     autocorrelate -> MUSIC_lin_array (+ find_local_max)  /  rootMUSIC_linear_array
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Every block of the
 * reference that sits on the path gets one opaque handle type with
 *
 *     doa_X_create(<the reference's make() arguments>)   -> handle, or NULL + doa_last_error()
 *     doa_X_work(h, noutput_items, <host pointers laid out like the GNU Radio item buffers>)
 *     doa_X_work_dev(h, noutput_items, <device pointers, same layouts>, hipStream_t as void*)
 *     doa_X_destroy(h)
 *
 * `work` returns the number of items produced (what the reference's work()/general_work()
 * returns) or a negative doa_status on failure; it never falls back to a CPU path: if no HIP
 * device / kernel is usable the call fails and doa_last_error() says why.
 *
 * Reference interfaces replaced (paths relative to the gr-doa tree):
 *   doa_autocorrelate_*          include/doa/autocorrelate.h:56, lib/autocorrelate_impl.cc:47-118
 *   doa_MUSIC_lin_array_*        include/doa/MUSIC_lin_array.h:56, lib/MUSIC_lin_array_impl.cc:47-150
 *   doa_find_local_max_*         include/doa/find_local_max.h:56, lib/find_local_max_impl.cc:47-194
 *   doa_rootMUSIC_linear_array_* include/doa/rootMUSIC_linear_array.h:54,
 *                                lib/rootMUSIC_linear_array_impl.cc:46-152
 *   doa_music_pipeline_*         the three blocks as wired by apps/run_MUSIC_lin_array_simulation.grc
 *                                (autocorrelate -> MUSIC_lin_array -> find_local_max(M, P, 0, 180))
 *   doa_root_pipeline_*          the two blocks as wired by apps/run_RootMUSIC_lin_array_simulation.grc
 *                                (autocorrelate -> rootMUSIC_linear_array)
 *
 * Threading: like GNU Radio's thread-per-block scheduler assumes, different handles may be used
 * from different threads concurrently; one handle must not be used from two threads at once.
 * Ownership: the caller owns every buffer it passes; the library owns its device tables, staging
 * buffers and (for the host-pointer entry points) one HIP stream per handle.
 * What the library does to the HOST PROCESS besides that: the first handle created on a device makes it create four
 * throw-away non-blocking HIP streams and run one 64-byte memset on each; they (and 64 bytes of device memory) stay alive for
 * the life of the process.  The HIP runtime maps streams onto its hardware queues lazily, and streams that caused a queue to be
 * created overlap kernels measurably worse than later ones (26 against 33 us per pipeline step); priming
 * the pool once makes every stream created afterwards -- the library's and the application's -- one of the good kind.
 * Set DOA_HIP_NO_QUEUE_PRIMING=1 in the environment to switch this off.
 * The pipeline handles also PROBE the lane streams they create (first doa_*_pipeline_work_dev_batches / chunked host call on
 * a handle): the runtime may put two streams on one hardware queue, where their kernels run one after the other (measured: a
 * 4-lane step at 44 instead of 37.5 us with one foreign stream alive), and nothing in the HIP API tells; so every new lane
 * and the lanes accepted before it run a one-wave kernel that sleeps 150 us and time-stamps itself, and a lane whose interval
 * does not overlap the others' is replaced (csrc/lane_streams.hip).  About a millisecond per handle, once;
 * DOA_HIP_NO_LANE_PROBE=1 switches it off.  Streams handed in with doa_*_pipeline_set_lane_streams are the caller's and
 * are taken as they are.
 * Diagnostics, profiling and fault-injection entry points used by the test suite are exported by the same library but
 * declared in doa_hip_test.h, not here: this header is the drop-in boundary only.
 */
#ifndef DOA_HIP_H
#define DOA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define DOA_HIP_API __attribute__((visibility("default")))
#else
#define DOA_HIP_API
#endif

typedef enum doa_status {
    DOA_OK = 0,
    DOA_ERR_INVALID_ARG = -1, /* a constructor/work argument violates the block's contract        */
    DOA_ERR_NO_DEVICE = -2,   /* no HIP device (or the HIP runtime failed to initialise)         */
    DOA_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed                      */
    DOA_ERR_UNSUPPORTED = -4, /* valid for the reference, not built into this library (size caps) */
    DOA_ERR_NUMERIC = -5      /* the reference would raise here (e.g. no root inside the circle)  */
} doa_status;

/* Largest array the HIP kernels are instantiated for (the reference's flowgraphs use 4, its QA
 * tests 4/8/16). */
#define DOA_MAX_ANT_ELE 16
/* Largest num_max_vals / num_targets handled by the peak-pick kernel. */
#define DOA_MAX_PEAKS 16

/* Thread-local description of the last failure on the calling thread ("" if none). */
DOA_HIP_API const char *doa_last_error(void);
/* Library/ABI version, bumped when a signature changes. */
DOA_HIP_API int doa_hip_abi_version(void);
/* Number of visible HIP devices (0 if none / runtime unusable). Does not create a context. */
DOA_HIP_API int doa_hip_device_count(void);

/* Layout advice for the N input streams of autocorrelate / music_pipeline on the device (the N stream pointers of
 * gr::doa::autocorrelate's general_work, lib/autocorrelate_impl.cc:83-100, when they are device memory): the
 * recommended distance in bytes between the first samples of consecutive streams that hold `stream_bytes` bytes each.
 * Every wave of the covariance kernel reads the same sample range of all N streams at the same time; streams whose
 * addresses agree modulo 8 KiB meet in the same HBM channels (measured on MI355X: 5.86 TB/s for N = 4 streams
 * 32 MiB apart against 6.25 TB/s with the distance returned here, 5.0 against 5.8 TB/s at N = 8).  The value is a
 * multiple of 16 (streams stay 16-byte aligned) and at least stream_bytes; any layout is accepted by the kernels, this
 * one is what the library's own staging buffers use. */
DOA_HIP_API size_t doa_stream_stride_bytes(size_t stream_bytes);

/* Internal precision of the batched Hermitian eigendecomposition and of the null-spectrum
 * evaluation used by MUSIC / Root-MUSIC / pipeline handles created *after* the call:
 * 64 (default) = double Jacobi + double Horner scan, 32 = float for both.  Item formats stay
 * complex64 in / float32 out either way; Root-MUSIC always finds its roots in double.
 * 64 is the parity configuration: it is what the tests pin to the fp64 evaluation of the
 * reference's formulas.  32 is an opt-in, NON-parity mode: it shares the reference's single
 * precision but not its LAPACK rounding sequence, so at spectrum nulls (where float Q is
 * cancellation-dominated) it differs from the reference by as much as two correct fp32
 * implementations differ from each other; it is tested to a loose bound only.
 * Returns DOA_OK or DOA_ERR_INVALID_ARG. */
DOA_HIP_API int doa_set_internal_precision(int bits);
DOA_HIP_API int doa_get_internal_precision(void);
/* The call above only sets the process-wide DEFAULT a handle copies when it is created (two threads that want handles of
 * different precisions would race on it); the precision is a property of the HANDLE and can be set on it directly, at
 * any time between two work calls: doa_X_set_internal_precision(h, 32 | 64) for the four handle types that run the
 * eigendecomposition (declared with their blocks below). */

/* ---------------------------------------------------------------------------------------------
 * autocorrelate — gr::doa::autocorrelate::make(inputs, snapshot_size, overlap_size, avg_method)
 *   (include/doa/autocorrelate.h:56).  gr::block with history overlap_size+1
 *   (lib/autocorrelate_impl.cc:57) and forecast nonoverlap*noutput (:75-80).
 * Item layouts: input_items[k] = stream k, gr_complex (float re, im), pointing at the first
 *   history sample exactly like general_work's input_items[k]; window i is the snapshot_size
 *   samples starting at input_items[k] + i*(snapshot_size-overlap_size) (:95-100).
 *   output = noutput_items column-major inputs x inputs gr_complex matrices (:103).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_autocorrelate doa_autocorrelate_t;

DOA_HIP_API doa_autocorrelate_t *doa_autocorrelate_create(int inputs, int snapshot_size,
                                                          int overlap_size, int avg_method);
DOA_HIP_API void doa_autocorrelate_destroy(doa_autocorrelate_t *h);
/* set_history() value the shell must apply: overlap_size + 1. */
DOA_HIP_API int doa_autocorrelate_history(const doa_autocorrelate_t *h);
/* forecast(): new input items required per stream for noutput_items outputs. */
DOA_HIP_API int doa_autocorrelate_forecast(const doa_autocorrelate_t *h, int noutput_items);
/* Samples per stream that must be readable behind input_items[k]:
 * (noutput_items-1)*(snapshot-overlap) + snapshot. */
DOA_HIP_API long long doa_autocorrelate_input_span(const doa_autocorrelate_t *h, int noutput_items);
/* general_work on host buffers; the shell then calls consume_each(forecast(noutput_items)). */
DOA_HIP_API int doa_autocorrelate_work(doa_autocorrelate_t *h, int noutput_items,
                                       const void *const *input_items, void *output_items0);
/* Same on device buffers: d_input_items is a HOST array of `inputs` DEVICE pointers. Asynchronous
 * on `hip_stream` (a hipStream_t, NULL = the default stream). */
DOA_HIP_API int doa_autocorrelate_work_dev(doa_autocorrelate_t *h, int noutput_items,
                                           const void *const *d_input_items, void *d_output_items0,
                                           void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * MUSIC_lin_array — gr::doa::MUSIC_lin_array::make(norm_spacing, num_targets, num_ant_ele,
 *   pspectrum_len) (include/doa/MUSIC_lin_array.h:56).  gr::sync_block.
 * Item layouts: input = column-major num_ant_ele^2 gr_complex (only the upper triangle is
 *   significant, as with LAPACK uplo='U'); output = pspectrum_len floats, dB normalised to the
 *   item's maximum (lib/MUSIC_lin_array_impl.cc:49-50,124-142).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_MUSIC_lin_array doa_MUSIC_lin_array_t;

DOA_HIP_API doa_MUSIC_lin_array_t *doa_MUSIC_lin_array_create(float norm_spacing, int num_targets,
                                                              int num_ant_ele, int pspectrum_len);
DOA_HIP_API void doa_MUSIC_lin_array_destroy(doa_MUSIC_lin_array_t *h);
DOA_HIP_API int doa_MUSIC_lin_array_work(doa_MUSIC_lin_array_t *h, int noutput_items,
                                         const void *input_items0, void *output_items0);
DOA_HIP_API int doa_MUSIC_lin_array_work_dev(doa_MUSIC_lin_array_t *h, int noutput_items,
                                             const void *d_input_items0, void *d_output_items0,
                                             void *hip_stream);
/* Diagnostics for tests: for 4 < num_ant_ele <= 16 and num_targets <= 4 (2 num_targets <= num_ant_ele), internal
 * precision 64, the noise projector of MUSIC / Root-MUSIC / music_pipeline handles is computed from the SIGNAL subspace
 * (shifted orthogonal iteration, every result checked by its residual and by a certificate that the subspace found is the
 * one of the num_targets largest eigenvalues), and an item that fails any check takes the full Jacobi
 * eigendecomposition instead.  This returns how many items took that fall-back since the last reset (process-wide, all
 * handles and all devices: one 64-bit device counter per device, each kernel adds into the counter of the device it runs on;
 * synchronises the devices), or -1 without a device. */
DOA_HIP_API long long doa_hip_evd_fallback_count(int reset);
/* Items processed so far — the counter the reference prints from its destructor
 * (lib/MUSIC_lin_array_impl.cc:92-95,146). */
DOA_HIP_API long long doa_MUSIC_lin_array_items_total(const doa_MUSIC_lin_array_t *h);
DOA_HIP_API int doa_MUSIC_lin_array_set_internal_precision(doa_MUSIC_lin_array_t *h, int bits);

/* ---------------------------------------------------------------------------------------------
 * find_local_max — gr::doa::find_local_max::make(num_max_vals, vector_len, x_min, x_max)
 *   (include/doa/find_local_max.h:56).  gr::sync_block with two outputs.
 * Item layouts: input = vector_len floats; output 0 = num_max_vals floats (peak values, descending
 *   value order); output 1 = num_max_vals floats (x-axis locations of those peaks, sorted
 *   descending on their own) (lib/find_local_max_impl.cc:49-50,186-188).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_find_local_max doa_find_local_max_t;

DOA_HIP_API doa_find_local_max_t *doa_find_local_max_create(int num_max_vals, int vector_len,
                                                            float x_min, float x_max);
DOA_HIP_API void doa_find_local_max_destroy(doa_find_local_max_t *h);
DOA_HIP_API int doa_find_local_max_work(doa_find_local_max_t *h, int noutput_items,
                                        const void *input_items0, void *output_items0,
                                        void *output_items1);
DOA_HIP_API int doa_find_local_max_work_dev(doa_find_local_max_t *h, int noutput_items,
                                            const void *d_input_items0, void *d_output_items0,
                                            void *d_output_items1, void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * rootMUSIC_linear_array — gr::doa::rootMUSIC_linear_array::make(norm_spacing, num_targets,
 *   num_ant_ele) (include/doa/rootMUSIC_linear_array.h:54).  gr::sync_block.
 * Item layouts: input as MUSIC_lin_array; output 0 = num_targets floats, angles in degrees,
 *   ascending (lib/rootMUSIC_linear_array_impl.cc:48-49,144-145).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_rootMUSIC_linear_array doa_rootMUSIC_linear_array_t;

DOA_HIP_API doa_rootMUSIC_linear_array_t *doa_rootMUSIC_linear_array_create(float norm_spacing,
                                                                            int num_targets,
                                                                            int num_ant_ele);
DOA_HIP_API void doa_rootMUSIC_linear_array_destroy(doa_rootMUSIC_linear_array_t *h);
DOA_HIP_API int doa_rootMUSIC_linear_array_work(doa_rootMUSIC_linear_array_t *h, int noutput_items,
                                                const void *input_items0, void *output_items0);
DOA_HIP_API int doa_rootMUSIC_linear_array_work_dev(doa_rootMUSIC_linear_array_t *h,
                                                    int noutput_items, const void *d_input_items0,
                                                    void *d_output_items0, void *hip_stream);
DOA_HIP_API int doa_rootMUSIC_linear_array_set_internal_precision(doa_rootMUSIC_linear_array_t *h, int bits);

/* ---------------------------------------------------------------------------------------------
 * antenna_correction — gr::doa::antenna_correction::make(num_ant_ele, config_filename)
 *   (include/doa/antenna_correction.h:55, lib/antenna_correction_impl.cc:47-99).  gr::sync_block,
 *   num_ant_ele gr_complex streams in and out: out_k[i] = g_k * in_k[i] with
 *   g_k = (1/gain_k) * exp(-j phase_k) read from a text file with one "gain phase" pair per line.
 *   create fails (message = the reference's std::invalid_argument text) when the file is missing
 *   or has too many / too few lines.
 * The block sits directly in front of autocorrelate; doa_autocorrelate_fuse_antenna_correction folds
 * it into K1 (R[a,b] *= g_a conj(g_b) before the forward-backward step) so the corrected streams are
 * never materialised.  (a "next" row beyond the north-star path.)
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_antenna_correction doa_antenna_correction_t;

DOA_HIP_API doa_antenna_correction_t *doa_antenna_correction_create(int num_ant_ele, const char *config_filename);
/* The same block from explicit per-stream complex gains (re, im interleaved) instead of a file: what
 * python/phase_correct_hier.py:90-97 builds out of multiply_const_vcc blocks (stream 0 untouched, stream p+1 times
 * exp(j phase_p)); the Python mirror doa.phase_correct_hier parses the phase file as the reference does and calls this. */
DOA_HIP_API doa_antenna_correction_t *doa_antenna_correction_create_gains(int num_ant_ele, const float *gains_re_im);
DOA_HIP_API void doa_antenna_correction_destroy(doa_antenna_correction_t *h);
/* Copies the num_ant_ele complex gains (re, im interleaved) out; returns num_ant_ele. */
DOA_HIP_API int doa_antenna_correction_gains(const doa_antenna_correction_t *h, float *gains_re_im);
DOA_HIP_API int doa_antenna_correction_work(doa_antenna_correction_t *h, int noutput_items,
                                            const void *const *input_items, void *const *output_items);
DOA_HIP_API int doa_antenna_correction_work_dev(doa_antenna_correction_t *h, int noutput_items,
                                                const void *const *d_input_items, void *const *d_output_items,
                                                void *hip_stream);
/* Fold a per-stream complex gain into this autocorrelate handle (gains_re_im: inputs pairs, or NULL
 * to remove it).  Equivalent to an antenna_correction block feeding the autocorrelate block. */
DOA_HIP_API int doa_autocorrelate_fuse_antenna_correction(doa_autocorrelate_t *h, const float *gains_re_im);

/* ---------------------------------------------------------------------------------------------
 * calibrate_lin_array — gr::doa::calibrate_lin_array::make(norm_spacing, num_ant_ele, pilot_angle)
 *   (include/doa/calibrate_lin_array.h, lib/calibrate_lin_array_impl.cc:36-134).  gr::sync_block:
 *   input = column-major num_ant_ele^2 gr_complex covariance items measured with one pilot source at
 *   pilot_angle degrees; output = num_ant_ele gr_complex per item, the estimated per-antenna complex
 *   responses (unit-norm vector).  The reference's output carries an arbitrary unit-modulus factor
 *   (LAPACK eigenvector phase); here element 0 is real and non-negative.  
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_calibrate_lin_array doa_calibrate_lin_array_t;

DOA_HIP_API doa_calibrate_lin_array_t *doa_calibrate_lin_array_create(float norm_spacing, int num_ant_ele,
                                                                      float pilot_angle);
DOA_HIP_API void doa_calibrate_lin_array_destroy(doa_calibrate_lin_array_t *h);
DOA_HIP_API int doa_calibrate_lin_array_work(doa_calibrate_lin_array_t *h, int noutput_items,
                                             const void *input_items0, void *output_items0);
DOA_HIP_API int doa_calibrate_lin_array_work_dev(doa_calibrate_lin_array_t *h, int noutput_items,
                                                 const void *d_input_items0, void *d_output_items0,
                                                 void *hip_stream);
DOA_HIP_API int doa_calibrate_lin_array_set_internal_precision(doa_calibrate_lin_array_t *h, int bits);

/* ---------------------------------------------------------------------------------------------
 * music_pipeline — autocorrelate -> MUSIC_lin_array -> find_local_max(num_targets, pspectrum_len,
 *   0, 180) on device-resident streams, the batch entry point the benchmark drives
 *   (apps/run_MUSIC_lin_array_simulation.grc wiring).  All pointers are DEVICE pointers except
 *   d_input_items itself (host array of device pointers).  d_cov_out and d_spectrum_out may be
 *   NULL when the caller does not want that intermediate materialised in its own buffer.  A NULL spectrum pointer is
 *   the ANGLES-ONLY mode: for the benchmark-shaped spectra (pspectrum_len 256 / 512 / 1024, polynomial size = array size)
 *   the scan kernel then neither converts the row to dB nor writes it (the maximum of a normalised row is 0 dB by
 *   construction and its position follows from the null spectrum itself); peaks and angles are bit-identical to a call
 *   that asks for the spectrum (21.1 against 25.3 us per 4096-snapshot step on MI355X, 500-step runs of round 3).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_music_pipeline doa_music_pipeline_t;

DOA_HIP_API doa_music_pipeline_t *doa_music_pipeline_create(int inputs, int snapshot_size,
                                                            int overlap_size, int avg_method,
                                                            float norm_spacing, int num_targets,
                                                            int pspectrum_len, int max_batch);
DOA_HIP_API void doa_music_pipeline_destroy(doa_music_pipeline_t *h);
/* Same fusion as doa_autocorrelate_fuse_antenna_correction, for the pipeline's K1. */
DOA_HIP_API int doa_music_pipeline_fuse_antenna_correction(doa_music_pipeline_t *h, const float *gains_re_im);
DOA_HIP_API int doa_music_pipeline_work_dev(doa_music_pipeline_t *h, int noutput_items,
                                            const void *const *d_input_items, void *d_cov_out,
                                            void *d_spectrum_out, void *d_max_out,
                                            void *d_argmax_out, void *hip_stream);
/* n_batches independent batches of noutput_items snapshots each (<= max_batch) in ONE call, overlapped by the library:
 * batch b runs its K1 -> EVD -> scan chain on one of the handle's own LANES (a HIP stream plus a private workspace; 4
 * by default, rotating from call to call), so that the HBM-bound covariance kernel of one batch runs beside the
 * issue-bound EVD / scan kernels of its neighbours -- the overlap a caller otherwise has to build from several handles on
 * several streams of its own (reference work being chained: lib/autocorrelate_impl.cc:83-118 ->
 * lib/MUSIC_lin_array_impl.cc:121-142 -> lib/find_local_max_impl.cc:167-194, wiring
 * apps/run_MUSIC_lin_array_simulation.grc:1099-1370).  Results are bit-identical to n_batches work_dev calls.
 *   d_input_items   HOST array of n_batches * inputs DEVICE pointers (batch b: entries b*inputs .. b*inputs+inputs-1)
 *   d_cov_out, d_spectrum_out  HOST arrays of n_batches DEVICE pointers; the array or single entries may be NULL (no
 *                   covariance copy wanted / angles-only mode for that batch, as in work_dev)
 *   d_max_out, d_argmax_out    HOST arrays of n_batches DEVICE pointers (required)
 *   hip_stream      a hipStream_t: the call is asynchronous like work_dev and ordered on that stream AS A WHOLE -- every lane
 *                   it uses starts behind the work the stream held at the call (one event) and the stream continues behind
 *                   the last batch of every lane (one event per lane): one fork and one join per call whatever n_batches
 *                   is.  Or DOA_STREAM_DETACHED: no ordering against any caller stream (the inputs must be complete when
 *                   the call is made); the caller joins with doa_music_pipeline_synchronize before it touches the
 *                   outputs.  Cross-stream events cost tens of microseconds on this runtime: callers
 *                   that submit many short calls want the detached form.
 * Returns n_batches * noutput_items or a negative doa_status; after an error the join has still been enqueued. */
#define DOA_STREAM_DETACHED ((void *)(size_t)-1)
DOA_HIP_API int doa_music_pipeline_work_dev_batches(doa_music_pipeline_t *h, int n_batches, int noutput_items,
                                                    const void *const *d_input_items, void *const *d_cov_out,
                                                    void *const *d_spectrum_out, void *const *d_max_out,
                                                    void *const *d_argmax_out, void *hip_stream);
/* Host-side join: returns when every lane of the handle has finished what work_dev_batches gave it. */
DOA_HIP_API int doa_music_pipeline_synchronize(doa_music_pipeline_t *h);
/* Number of lanes work_dev_batches spreads its batches over (1..8, default 4; 1 = everything on hip_stream itself). */
DOA_HIP_API int doa_music_pipeline_set_lanes(doa_music_pipeline_t *h, int n_lanes);
/* Lanes on streams the CALLER created (n_lanes hipStream_t; they stay the caller's, the handle only uses them).  For a
 * host program that draws its streams from a pool of its own (PyTorch, a GNU Radio buffer manager): HIP maps streams
 * onto a few hardware queues in creation order, and which streams share a queue with which decides how well kernels of
 * different lanes overlap (measured: 26 against 33 us per 4096-snapshot step for four lanes on one set of streams or
 * another) -- the program that owns the process's streams is the one that can choose. */
DOA_HIP_API int doa_music_pipeline_set_lane_streams(doa_music_pipeline_t *h, int n_lanes, void *const *hip_streams);
DOA_HIP_API int doa_music_pipeline_set_internal_precision(doa_music_pipeline_t *h, int bits);
/* The same three blocks on HOST buffers (the layouts the GNU Radio scheduler hands to the blocks'
 * work(): input_items[k] = stream k, doa_autocorrelate_input_span(noutput_items) samples; outputs
 * noutput_items items each).  cov_out and spectrum_out may be NULL: only the 2*num_targets floats
 * per snapshot then cross PCIe on the way back.  Samples are moved in ~32 MiB chunks alternating over
 * two streams owned by the handle (transfers of neighbouring chunks overlap when the caller's
 * buffers are page-locked); returns when every output has landed.  This path is PCIe-bound
 * (N*(snapshot-overlap)*8 B per snapshot in), see DESIGN.md §6. */
DOA_HIP_API int doa_music_pipeline_work(doa_music_pipeline_t *h, int noutput_items,
                                        const void *const *input_items, void *cov_out,
                                        void *spectrum_out, void *max_out, void *argmax_out);

/* ---------------------------------------------------------------------------------------------
 * root_pipeline — autocorrelate -> rootMUSIC_linear_array on device-resident streams: the Root-MUSIC branch of the hot
 *   path as one handle (the chain apps/run_RootMUSIC_lin_array_simulation.grc wires; reference work being chained:
 *   lib/autocorrelate_impl.cc:83-118 -> lib/rootMUSIC_linear_array_impl.cc:90-152).  Same conventions as music_pipeline:
 *   all pointers are DEVICE pointers except the pointer arrays themselves; d_cov_out may be NULL.  Output: num_targets
 *   floats per snapshot, angles in degrees, ascending (rootMUSIC_linear_array's output 0).
 *   d_status_out (optional): one int per snapshot, 1 = the polynomial has no root strictly inside the unit circle -- the
 *   case in which the reference raises inside arma::index_min and the host entry returns DOA_ERR_NUMERIC; the device entries
 *   are asynchronous and leave the check to the caller.  Results are bit-identical to the two block handles chained by hand
 *   (doa_autocorrelate_work_dev -> doa_rootMUSIC_linear_array_work_dev).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_root_pipeline doa_root_pipeline_t;

DOA_HIP_API doa_root_pipeline_t *doa_root_pipeline_create(int inputs, int snapshot_size, int overlap_size,
                                                          int avg_method, float norm_spacing, int num_targets,
                                                          int max_batch);
DOA_HIP_API void doa_root_pipeline_destroy(doa_root_pipeline_t *h);
/* Same fusion as doa_autocorrelate_fuse_antenna_correction, for the pipeline's K1. */
DOA_HIP_API int doa_root_pipeline_fuse_antenna_correction(doa_root_pipeline_t *h, const float *gains_re_im);
DOA_HIP_API int doa_root_pipeline_work_dev(doa_root_pipeline_t *h, int noutput_items,
                                           const void *const *d_input_items, void *d_cov_out,
                                           void *d_angles_out, int *d_status_out, void *hip_stream);
/* n_batches independent batches in ONE call, overlapped over the handle's lanes; arguments, stream semantics
 * (DOA_STREAM_DETACHED included), error contract and return value as doa_music_pipeline_work_dev_batches.
 *   d_cov_out, d_status_out   HOST arrays of n_batches DEVICE pointers; the array or single entries may be NULL
 *   d_angles_out              HOST array of n_batches DEVICE pointers (required) */
DOA_HIP_API int doa_root_pipeline_work_dev_batches(doa_root_pipeline_t *h, int n_batches, int noutput_items,
                                                   const void *const *d_input_items, void *const *d_cov_out,
                                                   void *const *d_angles_out, int *const *d_status_out,
                                                   void *hip_stream);
DOA_HIP_API int doa_root_pipeline_synchronize(doa_root_pipeline_t *h);
DOA_HIP_API int doa_root_pipeline_set_lanes(doa_root_pipeline_t *h, int n_lanes);
DOA_HIP_API int doa_root_pipeline_set_lane_streams(doa_root_pipeline_t *h, int n_lanes, void *const *hip_streams);
DOA_HIP_API int doa_root_pipeline_set_internal_precision(doa_root_pipeline_t *h, int bits);
/* The same chain on HOST buffers (the layouts the GNU Radio scheduler hands to the blocks' work()); cov_out may be NULL.
 * Scheduler-sized calls take one staged copy each way, large ones ~32 MiB chunks alternating over two streams; returns when
 * every output has landed: noutput_items, or DOA_ERR_NUMERIC if some item had no root inside the unit circle (the angles of
 * the other items are valid, that item's are NaN). */
DOA_HIP_API int doa_root_pipeline_work(doa_root_pipeline_t *h, int noutput_items,
                                       const void *const *input_items, void *cov_out, void *angles_out);

/* ---------------------------------------------------------------------------------------------
 * compass_mean — blocks.vector_to_streams(float, num_streams) + the averaging of doa.compass
 *   (reference python/compass.py:134-136: next_angle = numpy.mean(input_items[0]) over the items of
 *   one work call; wiring apps/run_MUSIC_lin_array_simulation.py:199,236-239).
 *   input item = num_streams floats (port 1 of find_local_max); output = num_streams floats, the
 *   mean of each de-interleaved stream over the ninput_items items (NaN for 0 items, as numpy).
 *   Returns the number of items consumed (= ninput_items).  The compass GUI is out of scope.
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_compass_mean doa_compass_mean_t;

DOA_HIP_API doa_compass_mean_t *doa_compass_mean_create(int num_streams);
DOA_HIP_API void doa_compass_mean_destroy(doa_compass_mean_t *h);
DOA_HIP_API int doa_compass_mean_work(doa_compass_mean_t *h, int ninput_items,
                                      const void *input_items0, float *next_angle);
DOA_HIP_API int doa_compass_mean_work_dev(doa_compass_mean_t *h, int ninput_items,
                                          const void *d_input_items0, float *d_next_angle,
                                          void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * sim_source — the signal front end of the simulation flowgraphs as one generator
 *   (reference apps/run_MUSIC_lin_array_simulation.py:66-74 array manifold, :204-210 sig_source_c +
 *   noise_source_c per source -> add -> multiply_matrix_cc):
 *     x_n[t] = sum_m A[n][m] (tone_ampl[m] e^{j 2 pi tone_freq[m] t} + source_noise_ampl[m] (g + j g'))
 *              + antenna_noise_sigma (g + j g') / sqrt(2)
 *   tone_freq in cycles per sample; tone_ampl / source_noise_ampl may be NULL (1 / 0).  Noise is
 *   Philox4x32-10 keyed by `seed` (counter = sample-pair index and noise stream), so sample ranges
 *   are reproducible independently of call boundaries; it is not GNU Radio's generator.
 *   work produces the next noutput_items samples of the num_ant_ele output streams (complex64 each);
 *   every call except the last of a run must ask for an even number (seek positions are even).
 * --------------------------------------------------------------------------------------------- */
typedef struct doa_sim_source doa_sim_source_t;

DOA_HIP_API doa_sim_source_t *doa_sim_source_create(int num_ant_ele, int num_sources,
                                                    float norm_spacing, const float *theta_deg,
                                                    const double *tone_freq, const float *tone_ampl,
                                                    const float *source_noise_ampl,
                                                    float antenna_noise_sigma,
                                                    unsigned long long seed);
DOA_HIP_API void doa_sim_source_destroy(doa_sim_source_t *h);
DOA_HIP_API int doa_sim_source_seek(doa_sim_source_t *h, long long sample_index);
DOA_HIP_API long long doa_sim_source_tell(const doa_sim_source_t *h);
DOA_HIP_API int doa_sim_source_work(doa_sim_source_t *h, int noutput_items,
                                    void *const *output_items);
DOA_HIP_API int doa_sim_source_work_dev(doa_sim_source_t *h, int noutput_items,
                                        void *const *d_output_items, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DOA_HIP_H */
