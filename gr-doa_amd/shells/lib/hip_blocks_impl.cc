// hip_blocks_impl.cc — the four hot-path GNU Radio block shells of gr-doa, implemented as thin
// forwarders to the C ABI of libdoa_hip.so (include/doa_hip.h).  No arithmetic lives here: the shells
// only keep the GNU Radio contracts the reference blocks have —
//   autocorrelate           gr::block, history = overlap+1, forecast/consume_each of nonoverlap*n
//                           (reference lib/autocorrelate_impl.cc:47-65,75-80,114)
//   MUSIC_lin_array         gr::sync_block, vlen N^2 complex -> vlen P float (…/MUSIC_lin_array_impl.cc:47-50)
//   find_local_max          gr::sync_block, two outputs of vlen M (…/find_local_max_impl.cc:47-50)
//   rootMUSIC_linear_array  gr::sync_block, io_signature(1, M, M floats), port 0 written
//                           (…/rootMUSIC_linear_array_impl.cc:46-49,96)
//   antenna_correction      gr::sync_block, N complex streams in/out (…/antenna_correction_impl.cc:47-52)
//   calibrate_lin_array     gr::sync_block, vlen N^2 complex -> vlen N complex (…/calibrate_lin_array_impl.cc:46-51)
//   music_pipeline          (not in the reference) the first three wired as in run_MUSIC_lin_array_simulation.grc,
//                           as one gr::block over doa_music_pipeline_work
//   root_music_pipeline     (not in the reference) autocorrelate -> rootMUSIC_linear_array wired as in
//                           run_RootMUSIC_lin_array_simulation.grc, as one gr::block over doa_root_pipeline_work
// — and turn a failing ABI call into the behaviour a GNU Radio block has for it: constructors throw
// std::runtime_error / std::invalid_argument (as antenna_correction_impl.cc:58-73 does), work()
// returns WORK_DONE (-1) after logging, which stops the flowgraph.
#include <doa/MUSIC_lin_array.h>
#include <doa/antenna_correction.h>
#include <doa/autocorrelate.h>
#include <doa/calibrate_lin_array.h>
#include <doa/find_local_max.h>
#include <doa/music_pipeline.h>
#include <doa/rootMUSIC_linear_array.h>
#include <doa/root_music_pipeline.h>

#include <doa_hip.h>

#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace doa {
namespace {

[[noreturn]] void throw_create(const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + doa_last_error());
}

int work_failed(const char *what, int status)
{
    std::cerr << what << ": libdoa_hip status " << status << " (" << doa_last_error() << ")" << std::endl;
    return gr::block::WORK_DONE;
}

// Flowgraph-rate plumbing.  Under GNU Radio's default 64 KiB buffers these blocks see calls of a
// handful of items, and a call costs 50-100 us of launches, copies and one stream synchronisation whatever its
// size.  Two scheduler knobs change that, both set here so that a flowgraph needs no edit:
//   set_min_output_buffer    items the block's output buffers must hold: the scheduler then hands over up to half of that
//                            whenever upstream has it ready -- larger calls, no added latency, no change at end of stream.
//                            Default: a BYTE budget, not an item count -- GNU Radio allocates at least min_items x item size
//                            per output ring (mapped twice by vmcircbuf), so a fixed 2048 items would be 8 MiB for a 1024-point
//                            spectrum and 128 MiB for a 16384-point one.  min_items = clamp(2 MiB / largest output item, 1,
//                            2048): 512 spectra of 1024 points, 2048 covariance matrices or angle pairs.
//                            DOA_GR_MIN_OUTPUT_BUFFER=<items> overrides (0 = GNU Radio's default sizing).
//   DOA_GR_OUTPUT_MULTIPLE   work() is only called with a multiple of this many items (default 1 = the reference's
//                            scheduling).  For gr::block's with a decimation (autocorrelate, music_pipeline) GNU
//                            Radio also sizes the INPUT buffers from it (2 x multiple x (snapshot - overlap) samples),
//                            which is what lets a radio-rate flowgraph reach the PCIe-bound rate; the price is
//                            GNU Radio's own: up to multiple - 1 trailing items are not processed when a finite
//                            source ends.
constexpr long kMinOutputBufferBytes = 2l << 20;
void apply_scheduling_hints(gr::block *b, size_t largest_output_item_bytes)
{
    const char *e = getenv("DOA_GR_MIN_OUTPUT_BUFFER");
    long min_buf;
    if (e) min_buf = atol(e);
    else {
        min_buf = kMinOutputBufferBytes / (long)(largest_output_item_bytes ? largest_output_item_bytes : 1);
        min_buf = min_buf < 1 ? 1 : (min_buf > 2048 ? 2048 : min_buf);
    }
    if (min_buf > 0) b->set_min_output_buffer(min_buf);
    e = getenv("DOA_GR_OUTPUT_MULTIPLE");
    const int mult = e ? atoi(e) : 1;
    if (mult > 1) b->set_output_multiple(mult);
}

// ------------------------------------------------------------------------------------------------
class autocorrelate_hip : public autocorrelate
{
    doa_autocorrelate_t *d_h;
    int d_nonoverlap;

public:
    autocorrelate_hip(int inputs, int snapshot_size, int overlap_size, int avg_method)
        : gr::block("autocorrelate", gr::io_signature::make(inputs, inputs, sizeof(gr_complex)),
                    gr::io_signature::make(1, 1, sizeof(gr_complex) * inputs * inputs)),
          d_h(doa_autocorrelate_create(inputs, snapshot_size, overlap_size, avg_method)),
          d_nonoverlap(snapshot_size - overlap_size)
    {
        if (!d_h) throw_create("doa::autocorrelate");
        set_history(doa_autocorrelate_history(d_h));          // overlap_size + 1
        apply_scheduling_hints(this, sizeof(gr_complex) * inputs * inputs);
    }
    ~autocorrelate_hip() override { doa_autocorrelate_destroy(d_h); }

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override
    {
        const int need = doa_autocorrelate_forecast(d_h, noutput_items);
        for (auto &n : ninput_items_required) n = need;
    }

    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override
    {
        const int produced = doa_autocorrelate_work(d_h, noutput_items, input_items.data(), output_items[0]);
        if (produced < 0) return work_failed("doa::autocorrelate", produced);
        consume_each(d_nonoverlap * produced);
        return produced;
    }
};

// ------------------------------------------------------------------------------------------------
class MUSIC_lin_array_hip : public MUSIC_lin_array
{
    doa_MUSIC_lin_array_t *d_h;

public:
    MUSIC_lin_array_hip(float norm_spacing, int num_targets, int num_ant_ele, int pspectrum_len)
        : gr::sync_block("MUSIC_lin_array", gr::io_signature::make(1, 1, sizeof(gr_complex) * num_ant_ele * num_ant_ele),
                         gr::io_signature::make(1, 1, sizeof(float) * pspectrum_len)),
          d_h(doa_MUSIC_lin_array_create(norm_spacing, num_targets, num_ant_ele, pspectrum_len))
    {
        if (!d_h) throw_create("doa::MUSIC_lin_array");
        apply_scheduling_hints(this, sizeof(float) * pspectrum_len);
    }
    ~MUSIC_lin_array_hip() override
    {
        // the reference prints this counter from its destructor (lib/MUSIC_lin_array_impl.cc:92-95)
        std::cout << "Total output items produced: " << doa_MUSIC_lin_array_items_total(d_h) << std::endl;
        doa_MUSIC_lin_array_destroy(d_h);
    }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override
    {
        const int produced = doa_MUSIC_lin_array_work(d_h, noutput_items, input_items[0], output_items[0]);
        return produced < 0 ? work_failed("doa::MUSIC_lin_array", produced) : produced;
    }
};

// ------------------------------------------------------------------------------------------------
class find_local_max_hip : public find_local_max
{
    doa_find_local_max_t *d_h;

public:
    find_local_max_hip(int num_max_vals, int vector_len, float x_min, float x_max)
        : gr::sync_block("find_local_max", gr::io_signature::make(1, 1, sizeof(float) * vector_len),
                         gr::io_signature::make2(2, 2, num_max_vals * sizeof(float), num_max_vals * sizeof(float))),
          d_h(doa_find_local_max_create(num_max_vals, vector_len, x_min, x_max))
    {
        if (!d_h) throw_create("doa::find_local_max");
        apply_scheduling_hints(this, num_max_vals * sizeof(float));
    }
    ~find_local_max_hip() override { doa_find_local_max_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override
    {
        const int produced = doa_find_local_max_work(d_h, noutput_items, input_items[0], output_items[0], output_items[1]);
        return produced < 0 ? work_failed("doa::find_local_max", produced) : produced;
    }
};

// ------------------------------------------------------------------------------------------------
class rootMUSIC_linear_array_hip : public rootMUSIC_linear_array
{
    doa_rootMUSIC_linear_array_t *d_h;

public:
    rootMUSIC_linear_array_hip(float norm_spacing, int num_targets, int num_ant_ele)
        : gr::sync_block("rootMUSIC_linear_array",
                         gr::io_signature::make(1, 1, sizeof(gr_complex) * num_ant_ele * num_ant_ele),
                         gr::io_signature::make(1, num_targets, num_targets * sizeof(float))),
          d_h(doa_rootMUSIC_linear_array_create(norm_spacing, num_targets, num_ant_ele))
    {
        if (!d_h) throw_create("doa::rootMUSIC_linear_array");
        apply_scheduling_hints(this, num_targets * sizeof(float));
    }
    ~rootMUSIC_linear_array_hip() override { doa_rootMUSIC_linear_array_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override
    {
        const int produced = doa_rootMUSIC_linear_array_work(d_h, noutput_items, input_items[0], output_items[0]);
        return produced < 0 ? work_failed("doa::rootMUSIC_linear_array", produced) : produced;
    }
};

// ------------------------------------------------------------------------------------------------
class antenna_correction_hip : public antenna_correction
{
    doa_antenna_correction_t *d_h;

public:
    antenna_correction_hip(int num_ant_ele, char *config_filename)
        : gr::sync_block("antenna_correction", gr::io_signature::make(num_ant_ele, num_ant_ele, sizeof(gr_complex)),
                         gr::io_signature::make(num_ant_ele, num_ant_ele, sizeof(gr_complex))),
          d_h(doa_antenna_correction_create(num_ant_ele, config_filename))
    {
        // the reference throws std::invalid_argument for a bad configuration file (antenna_correction_impl.cc:58-73)
        if (!d_h) throw std::invalid_argument(doa_last_error());
    }
    ~antenna_correction_hip() override { doa_antenna_correction_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override
    {
        const int produced = doa_antenna_correction_work(d_h, noutput_items, input_items.data(), output_items.data());
        return produced < 0 ? work_failed("doa::antenna_correction", produced) : produced;
    }
};

// ------------------------------------------------------------------------------------------------
class calibrate_lin_array_hip : public calibrate_lin_array
{
    doa_calibrate_lin_array_t *d_h;

public:
    calibrate_lin_array_hip(float norm_spacing, int num_ant_ele, float pilot_angle)
        : gr::sync_block("calibrate_lin_array", gr::io_signature::make(1, 1, num_ant_ele * num_ant_ele * sizeof(gr_complex)),
                         gr::io_signature::make(1, 1, num_ant_ele * sizeof(gr_complex))),
          d_h(doa_calibrate_lin_array_create(norm_spacing, num_ant_ele, pilot_angle))
    {
        if (!d_h) throw_create("doa::calibrate_lin_array");
        apply_scheduling_hints(this, num_ant_ele * sizeof(gr_complex));
    }
    ~calibrate_lin_array_hip() override { doa_calibrate_lin_array_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override
    {
        const int produced = doa_calibrate_lin_array_work(d_h, noutput_items, input_items[0], output_items[0]);
        return produced < 0 ? work_failed("doa::calibrate_lin_array", produced) : produced;
    }
};

// ------------------------------------------------------------------------------------------------
// autocorrelate -> MUSIC_lin_array -> find_local_max as one block over doa_music_pipeline_work: one upload, the
// whole chain on the device, one download of the connected ports (see include/doa/music_pipeline.h)
class music_pipeline_hip : public music_pipeline
{
    static constexpr int kMaxBatch = 4096;
    doa_music_pipeline_t *d_h;
    int d_nonoverlap, d_M, d_P;
    std::vector<float> d_max_scratch;                     // port 1 (peak values) when it is not connected

public:
    music_pipeline_hip(int inputs, int snapshot_size, int overlap_size, int avg_method, float norm_spacing,
                       int num_targets, int pspectrum_len)
        : gr::block("music_pipeline", gr::io_signature::make(inputs, inputs, sizeof(gr_complex)),
                    gr::io_signature::makev(1, 3, std::vector<int>{(int)(num_targets * sizeof(float)),
                                                                   (int)(num_targets * sizeof(float)),
                                                                   (int)(pspectrum_len * sizeof(float))})),
          d_h(doa_music_pipeline_create(inputs, snapshot_size, overlap_size, avg_method, norm_spacing, num_targets,
                                        pspectrum_len, kMaxBatch)),
          d_nonoverlap(snapshot_size - overlap_size), d_M(num_targets), d_P(pspectrum_len)
    {
        if (!d_h) throw_create("doa::music_pipeline");
        set_history(overlap_size + 1);                        // as doa::autocorrelate (autocorrelate_impl.cc:56-57)
        apply_scheduling_hints(this, (size_t)pspectrum_len * sizeof(float));
    }
    ~music_pipeline_hip() override { doa_music_pipeline_destroy(d_h); }

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override
    {
        for (auto &n : ninput_items_required) n = d_nonoverlap * noutput_items;     // autocorrelate_impl.cc:75-80
    }

    // device-resident callers (no scheduler buffers involved): several batches per call, overlapped over the handle's own lanes
    int work_device_batches(int n_batches, int noutput_items, const void *const *d_input_items, void *const *d_spectrum_out,
                            void *const *d_max_out, void *const *d_argmax_out, void *hip_stream) override
    {
        const int produced = doa_music_pipeline_work_dev_batches(d_h, n_batches, noutput_items, d_input_items, nullptr, d_spectrum_out,
                                                                 d_max_out, d_argmax_out, hip_stream);
        return produced < 0 ? work_failed("doa::music_pipeline", produced) : produced;
    }
    int synchronize_device() override { return doa_music_pipeline_synchronize(d_h) == 0 ? 0 : work_failed("doa::music_pipeline", -3); }
    int max_batch() const override { return kMaxBatch; }

    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override
    {
        float *argmax = static_cast<float *>(output_items[0]);
        float *maxv = output_items.size() > 1 ? static_cast<float *>(output_items[1]) : nullptr;
        float *spec = output_items.size() > 2 ? static_cast<float *>(output_items[2]) : nullptr;
        if (!maxv) {
            d_max_scratch.resize((size_t)kMaxBatch * d_M);
            maxv = d_max_scratch.data();
        }
        std::vector<const void *> in(input_items.size());
        int done = 0;
        while (done < noutput_items) {                        // the handle's workspace holds kMaxBatch items
            const int n = noutput_items - done < kMaxBatch ? noutput_items - done : kMaxBatch;
            for (size_t k = 0; k < in.size(); k++)
                in[k] = static_cast<const gr_complex *>(input_items[k]) + (size_t)done * d_nonoverlap;
            float *mv = output_items.size() > 1 ? maxv + (size_t)done * d_M : maxv;
            const int produced = doa_music_pipeline_work(d_h, n, in.data(), nullptr, spec ? spec + (size_t)done * d_P : nullptr,
                                                         mv, argmax + (size_t)done * d_M);
            if (produced < 0) return work_failed("doa::music_pipeline", produced);
            done += produced;
        }
        consume_each(d_nonoverlap * done);                    // autocorrelate_impl.cc:114
        return done;
    }
};

// ------------------------------------------------------------------------------------------------
class root_music_pipeline_hip : public root_music_pipeline
{
    static constexpr int kMaxBatch = 4096;
    doa_root_pipeline_t *d_h;
    int d_nonoverlap, d_M;

public:
    root_music_pipeline_hip(int inputs, int snapshot_size, int overlap_size, int avg_method, float norm_spacing, int num_targets)
        : gr::block("root_music_pipeline", gr::io_signature::make(inputs, inputs, sizeof(gr_complex)),
                    gr::io_signature::make(1, 1, num_targets * sizeof(float))),
          d_h(doa_root_pipeline_create(inputs, snapshot_size, overlap_size, avg_method, norm_spacing, num_targets, kMaxBatch)),
          d_nonoverlap(snapshot_size - overlap_size), d_M(num_targets)
    {
        if (!d_h) throw_create("doa::root_music_pipeline");
        set_history(overlap_size + 1);                        // as doa::autocorrelate (autocorrelate_impl.cc:56-57)
        apply_scheduling_hints(this, (size_t)num_targets * sizeof(float));
    }
    ~root_music_pipeline_hip() override { doa_root_pipeline_destroy(d_h); }

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override
    {
        for (auto &n : ninput_items_required) n = d_nonoverlap * noutput_items;     // autocorrelate_impl.cc:75-80
    }

    int work_device_batches(int n_batches, int noutput_items, const void *const *d_input_items, void *const *d_angles_out,
                            int *const *d_status_out, void *hip_stream) override
    {
        const int produced = doa_root_pipeline_work_dev_batches(d_h, n_batches, noutput_items, d_input_items, nullptr, d_angles_out,
                                                                d_status_out, hip_stream);
        return produced < 0 ? work_failed("doa::root_music_pipeline", produced) : produced;
    }
    int synchronize_device() override { return doa_root_pipeline_synchronize(d_h) == 0 ? 0 : work_failed("doa::root_music_pipeline", -3); }
    int max_batch() const override { return kMaxBatch; }

    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override
    {
        float *aoa = static_cast<float *>(output_items[0]);
        std::vector<const void *> in(input_items.size());
        int done = 0;
        while (done < noutput_items) {                        // the handle's workspace holds kMaxBatch items
            const int n = noutput_items - done < kMaxBatch ? noutput_items - done : kMaxBatch;
            for (size_t k = 0; k < in.size(); k++)
                in[k] = static_cast<const gr_complex *>(input_items[k]) + (size_t)done * d_nonoverlap;
            // (an item without a root inside the unit circle: DOA_ERR_NUMERIC -- the reference's work() throws there -- stops
            // the flowgraph like any other failure)
            const int produced = doa_root_pipeline_work(d_h, n, in.data(), nullptr, aoa + (size_t)done * d_M);
            if (produced < 0) return work_failed("doa::root_music_pipeline", produced);
            done += produced;
        }
        consume_each(d_nonoverlap * done);                    // autocorrelate_impl.cc:114
        return done;
    }
};

}  // namespace

music_pipeline::sptr music_pipeline::make(int inputs, int snapshot_size, int overlap_size, int avg_method,
                                          float norm_spacing, int num_targets, int pspectrum_len)
{
    return gnuradio::get_initial_sptr(new music_pipeline_hip(inputs, snapshot_size, overlap_size, avg_method, norm_spacing,
                                                             num_targets, pspectrum_len));
}
root_music_pipeline::sptr root_music_pipeline::make(int inputs, int snapshot_size, int overlap_size, int avg_method,
                                                    float norm_spacing, int num_targets)
{
    return gnuradio::get_initial_sptr(new root_music_pipeline_hip(inputs, snapshot_size, overlap_size, avg_method, norm_spacing,
                                                                  num_targets));
}
calibrate_lin_array::sptr calibrate_lin_array::make(float norm_spacing, int num_ant_ele, float pilot_angle)
{
    return gnuradio::get_initial_sptr(new calibrate_lin_array_hip(norm_spacing, num_ant_ele, pilot_angle));
}
antenna_correction::sptr antenna_correction::make(int num_ant_ele, char *config_filename)
{
    return gnuradio::get_initial_sptr(new antenna_correction_hip(num_ant_ele, config_filename));
}
autocorrelate::sptr autocorrelate::make(int inputs, int snapshot_size, int overlap_size, int avg_method)
{
    return gnuradio::get_initial_sptr(new autocorrelate_hip(inputs, snapshot_size, overlap_size, avg_method));
}
MUSIC_lin_array::sptr MUSIC_lin_array::make(float norm_spacing, int num_targets, int num_ant_ele, int pspectrum_len)
{
    return gnuradio::get_initial_sptr(new MUSIC_lin_array_hip(norm_spacing, num_targets, num_ant_ele, pspectrum_len));
}
find_local_max::sptr find_local_max::make(int num_max_vals, int vector_len, float x_min, float x_max)
{
    return gnuradio::get_initial_sptr(new find_local_max_hip(num_max_vals, vector_len, x_min, x_max));
}
rootMUSIC_linear_array::sptr rootMUSIC_linear_array::make(float norm_spacing, int num_targets, int num_ant_ele)
{
    return gnuradio::get_initial_sptr(new rootMUSIC_linear_array_hip(norm_spacing, num_targets, num_ant_ele));
}

}  // namespace doa
}  // namespace gr
