/*
 * doa_hip_test.h -- entry points of libdoa_hip.so that exist for the TEST SUITE and for profiling, not for applications:
 * diagnostics of intermediate results (what the parity tests compare against the oracle), a stage mask for timing one
 * kernel of the pipeline on valid intermediates, and fault injection for the error paths.  Same library, same symbols as
 * before round 4; kept out of doa_hip.h so that the drop-in boundary declares only what a gr-doa block shell binds.
 */
#ifndef DOA_HIP_TEST_H
#define DOA_HIP_TEST_H

#include "doa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics used by the parity tests (host buffers, synchronous): the noise-subspace projector
 * U_N U_N^H of each item (column-major num_ant_ele^2 gr_complex, lib/MUSIC_lin_array_impl.cc:133)
 * and the un-normalised null spectrum Q_i = Re(a_i^H P_N a_i) (:139), pspectrum_len floats per
 * item.  Either output pointer may be NULL. */
DOA_HIP_API int doa_MUSIC_lin_array_debug(doa_MUSIC_lin_array_t *h, int noutput_items,
                                          const void *input_items0, void *projector_out,
                                          void *null_spectrum_out);

/* Diagnostics for parity tests (as doa_MUSIC_lin_array_debug): besides the angles, the 2*num_ant_ele-2 polynomial
 * roots the solver found per item (roots_out: interleaved re, im doubles; may be NULL) and the per-item status
 * (status_out: 1 = no root strictly inside the unit circle, the case in which the reference raises inside
 * arma::index_min and doa_..._work returns DOA_ERR_NUMERIC; may be NULL).  Always returns noutput_items on success. */
DOA_HIP_API int doa_rootMUSIC_linear_array_debug(doa_rootMUSIC_linear_array_t *h, int noutput_items,
                                                 const void *input_items0, void *output_items0,
                                                 void *roots_out, int *status_out);
/* Diagnostics, second half: ONLY the root-selection stage of work() (lib/rootMUSIC_linear_array_impl.cc:122-145), run
 * on the device -- the very code the solver kernel ends in -- on CALLER-SUPPLIED roots (roots_in: noutput_items x
 * (2*num_ant_ele-2) interleaved re, im doubles, host memory), so that every branch of the rule can be driven with
 * hand-made root lists: fewer than num_targets roots strictly inside the unit circle (missing slots read 90 degrees,
 * :131-141), roots exactly on the circle (excluded by dist > 0, :125), equal distances (index_min takes the first),
 * no interior root at all (status 1 / NaN angles; the reference raises).  PARITY UNPINNED in two corners, both stated
 * in DESIGN.md section 5: the reference tests a FLOAT dist = 1 - |z| of cgeev's float roots, here dist is formed in double from
 * double roots (a root within 6e-8 of the circle is dropped there and kept here); NaN angles (|arg z| > 2 pi d) sort
 * last here, arma::sort's treatment of NaN depends on the Armadillo version. */
DOA_HIP_API int doa_rootMUSIC_linear_array_select_debug(doa_rootMUSIC_linear_array_t *h, int noutput_items,
                                                        const void *roots_in, void *output_items0,
                                                        int *status_out);

/* Profiling aid: which stages later work_dev calls on this handle launch (bit 0 = K1 covariance, bit 1 = K2+K3
 * EVD, bit 2 = K4+K5 scan + peak pick; default 7).  A dropped stage leaves its outputs as the previous call
 * wrote them, so a profiler can time one kernel on valid intermediates; not for production use. */
DOA_HIP_API int doa_music_pipeline_set_stages(doa_music_pipeline_t *h, int stage_mask);
/* Test aids for the error paths of doa_music_pipeline_work and doa_music_pipeline_work_dev_batches (not for production
 * use).  inject_failure: the NEXT such call on this handle behaves as if a HIP call had failed in chunk `chunk_index` (0 =
 * the first ~32 MiB chunk, or the only one of a scheduler-sized call) after that chunk's uploads were enqueued -- for the
 * batches entry: before batch `chunk_index` is launched, the earlier ones already running on their lanes; one-shot, -1
 * disarms.  Whatever fails inside either call, it returns only after every lane it used has been synchronised (the
 * detached form included), so nothing of a failed call is still running or copying afterwards; lanes_idle reports exactly
 * that (1 = all lanes idle, 0 = work pending, < 0 = error). */
DOA_HIP_API int doa_music_pipeline_inject_failure(doa_music_pipeline_t *h, int chunk_index);
DOA_HIP_API int doa_music_pipeline_lanes_idle(doa_music_pipeline_t *h);

/* The same two test aids for root_pipeline. */
DOA_HIP_API int doa_root_pipeline_inject_failure(doa_root_pipeline_t *h, int chunk_index);
DOA_HIP_API int doa_root_pipeline_lanes_idle(doa_root_pipeline_t *h);

/* The device (HIP ordinal) that holds the fall-back counter a K2+K3 launch made NOW -- with the calling thread's current
 * device -- would add into, or -1 when there is none.  The counters are per device since round 4 (a kernel must never add
 * into another device's memory); the test creates a handle, binds its device and checks that this equals it. */
DOA_HIP_API int doa_hip_evd_fallback_counter_device_debug(void);

/* The lane streams of the pipeline handles are probed when they are created (gr-doa_amd/csrc/lane_streams.hip): how many
 * lanes of the set created LAST in this process were seen to run their kernels side by side (-1: none created yet, or the
 * probing is switched off with DOA_HIP_NO_LANE_PROBE), and how many candidate streams were set aside on the way because they
 * shared a hardware queue with an accepted lane. */
DOA_HIP_API int doa_hip_lane_streams_verified_debug(void);
DOA_HIP_API int doa_hip_lane_streams_set_aside_debug(void);

#ifdef __cplusplus
}
#endif
#endif /* DOA_HIP_TEST_H */
