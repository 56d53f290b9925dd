// gr::doa::music_pipeline — autocorrelate -> MUSIC_lin_array -> find_local_max(num_targets, pspectrum_len, 0, 180)
// as ONE block: the hier block a maintainer would drop into run_MUSIC_lin_array_simulation.grc in place of
// the three (reference wiring: apps/run_MUSIC_lin_array_simulation.grc:1154-1205,1099-1150,1327-1370).  Not a
// block of the reference; its ports are the three blocks' outer ports:
//   in   N streams of gr_complex, history overlap_size + 1 (as gr::doa::autocorrelate)
//   out0 vlen num_targets float   find_local_max port 1: peak locations (degrees), descending
//   out1 vlen num_targets float   find_local_max port 0: peak values (dB), rank order          (optional)
//   out2 vlen pspectrum_len float MUSIC_lin_array's spectrum                                    (optional)
// One work() call = one upload of the new samples, the whole chain on the device, one download of what is
// connected: the covariance items and (unless out2 is connected) the spectra never cross PCIe.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

class DOA_API music_pipeline : virtual public gr::block
{
public:
    typedef DOA_SPTR<music_pipeline> sptr;
    static sptr make(int inputs, int snapshot_size, int overlap_size, int avg_method, float norm_spacing, int num_targets,
                     int pspectrum_len);

    // For callers whose streams are ALREADY on the device (another accelerator block upstream, a capture ring in HBM): the
    // same chain without the scheduler's host buffers -- n_batches batches of noutput_items (<= max_batch()) snapshots per
    // call, overlapped by the library over the block's own lanes (doa_music_pipeline_work_dev_batches, include/doa_hip.h:
    // pointer arrays as documented there; hip_stream a hipStream_t or DOA_STREAM_DETACHED, then synchronize_device()).
    // Returns the snapshots produced or WORK_DONE after logging, like general_work.
    virtual int work_device_batches(int n_batches, int noutput_items, const void *const *d_input_items,
                                    void *const *d_spectrum_out, void *const *d_max_out, void *const *d_argmax_out,
                                    void *hip_stream) = 0;
    virtual int synchronize_device() = 0;
    virtual int max_batch() const = 0;
};

}  // namespace doa
}  // namespace gr
