// gr::doa::root_music_pipeline — autocorrelate -> rootMUSIC_linear_array as ONE block: what a maintainer would drop into
// run_RootMUSIC_lin_array_simulation.grc in place of the two (reference wiring: apps/run_RootMUSIC_lin_array_simulation.grc,
// blocks doa_autocorrelate_0 -> doa_rootMUSIC_linear_array_0; work being chained: lib/autocorrelate_impl.cc:83-118 ->
// lib/rootMUSIC_linear_array_impl.cc:90-152).  Not a block of the reference; its ports are the two blocks' outer ports:
//   in   N streams of gr_complex, history overlap_size + 1 (as gr::doa::autocorrelate)
//   out0 vlen num_targets float   rootMUSIC_linear_array port 0: angles in degrees, ascending
// One work() call = one upload of the new samples, the whole chain on the device, one download of the angles: the covariance
// items never cross PCIe.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

class DOA_API root_music_pipeline : virtual public gr::block
{
public:
    typedef DOA_SPTR<root_music_pipeline> sptr;
    static sptr make(int inputs, int snapshot_size, int overlap_size, int avg_method, float norm_spacing, int num_targets);

    // For callers whose streams are ALREADY on the device: n_batches batches of noutput_items (<= max_batch()) snapshots per
    // call, overlapped by the library over the block's own lanes (doa_root_pipeline_work_dev_batches, include/doa_hip.h;
    // hip_stream a hipStream_t or DOA_STREAM_DETACHED, then synchronize_device()).  d_status_out may be NULL.
    virtual int work_device_batches(int n_batches, int noutput_items, const void *const *d_input_items, void *const *d_angles_out,
                                    int *const *d_status_out, void *hip_stream) = 0;
    virtual int synchronize_device() = 0;
    virtual int max_batch() const = 0;
};

}  // namespace doa
}  // namespace gr
