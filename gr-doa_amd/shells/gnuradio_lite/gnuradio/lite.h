// gnuradio_lite — a minimal stand-in for the slice of the GNU Radio 3.7 runtime API that the gr-doa
// hot-path block shells compile against (gr::block, gr::sync_block, gr::io_signature,
// gnuradio::get_initial_sptr, the gr_vector_* typedefs), plus a tiny single-threaded scheduler.
// It exists only because GNU Radio is not installable in the build image; when the real headers
// are present the shells include them instead (see shells/include/doa/api.h) and none of this is
// used.  Only behaviour the blocks rely on is modelled: history (zero pre-roll), forecast,
// consume_each, fixed-rate sync blocks, multi-port outputs.
#pragma once

#include <algorithm>
#include <chrono>
#include <complex>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace boost_lite {
template <class T> using shared_ptr = std::shared_ptr<T>;
}

namespace gr {

class io_signature
{
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, std::vector<int>(1, sizeof_stream_item)));
    }
    static sptr make2(int min_streams, int max_streams, int size1, int size2)
    {
        return sptr(new io_signature(min_streams, max_streams, std::vector<int>{size1, size2}));
    }
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizes)
    {
        return sptr(new io_signature(min_streams, max_streams, sizes));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int port) const
    {
        return d_sizes[std::min<size_t>((size_t)port, d_sizes.size() - 1)];
    }

private:
    io_signature(int mn, int mx, std::vector<int> sizes) : d_min(mn), d_max(mx), d_sizes(std::move(sizes)) {}
    int d_min, d_max;
    std::vector<int> d_sizes;
};

class block
{
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    virtual ~block() {}
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    unsigned history() const { return d_history; }
    void set_history(unsigned h) { d_history = h; }
    void set_output_multiple(int m) { d_output_multiple = m; }
    int output_multiple() const { return d_output_multiple; }
    // GNU Radio sizes a block's output buffers (and, through decimation x output_multiple, the buffers feeding it)
    // from these; the scheduler then hands work() up to half a buffer per call
    void set_min_output_buffer(long min_items) { d_min_output_buffer = min_items; }
    long min_output_buffer() const { return d_min_output_buffer; }
    virtual bool fixed_rate() const { return false; }
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required)
    {
        for (auto &n : ninput_items_required) n = noutput_items + (int)history() - 1;
    }
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
    void consume_each(int how_many) { d_consumed = how_many; }
    int last_consumed() const { return d_consumed; }   // scheduler side

protected:
    block() {}
    block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}
    // virtual-inheritance friendly late initialisation (the most-derived impl constructs gr::block)
    std::string d_name;
    io_signature::sptr d_in, d_out;
    unsigned d_history = 1;
    int d_output_multiple = 1;
    long d_min_output_buffer = -1;
    int d_consumed = 0;
};

class sync_block : public block
{
public:
    bool fixed_rate() const override { return true; }
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override
    {
        int r = work(noutput_items, input_items, output_items);
        if (r > 0) consume_each(r);
        return r;
    }

protected:
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : block(name, in, out) {}
};

// ---- a tiny scheduler: runs ONE block over whole input vectors in scheduler-sized calls ----------
namespace lite {

struct port_data {
    std::vector<char> bytes;
    size_t item_size = 0;
    size_t items() const { return item_size ? bytes.size() / item_size : 0; }
};

// Feeds `inputs` (one buffer per input port, items of the block's input item size) through `blk`
// with at most `max_noutput` items per call (what default-sized scheduler buffers would allow; a block that asked
// for larger buffers gets calls of up to half of min_output_buffer()), in multiples of output_multiple() -- a
// remainder smaller than one multiple is never processed, as in GNU Radio -- honouring
// history/forecast/consume_each; returns one buffer per output port.  work_seconds (optional) accumulates the
// wall time spent inside general_work().
inline std::vector<port_data> run_block(block &blk, const std::vector<port_data> &inputs, int n_out_ports,
                                        int max_noutput = 8, double *work_seconds = nullptr)
{
    if (blk.min_output_buffer() > 0) max_noutput = std::max<long>(max_noutput, blk.min_output_buffer() / 2);
    const int mult = std::max(1, blk.output_multiple());
    max_noutput = std::max(mult, max_noutput - max_noutput % mult);
    const int n_in = (int)inputs.size();
    const unsigned hist = blk.history() - 1;
    std::vector<std::vector<char>> in(n_in);
    size_t avail = (size_t)-1;
    for (int p = 0; p < n_in; p++) {
        const size_t isz = (size_t)blk.input_signature()->sizeof_stream_item(p);
        in[p].assign(hist * isz, 0);                                   // zero-filled history
        in[p].insert(in[p].end(), inputs[p].bytes.begin(), inputs[p].bytes.end());
        avail = std::min(avail, inputs[p].bytes.size() / isz);
    }
    std::vector<port_data> out(n_out_ports);
    for (int p = 0; p < n_out_ports; p++) out[p].item_size = (size_t)blk.output_signature()->sizeof_stream_item(p);
    size_t consumed = 0;
    std::vector<std::vector<char>> ring_in(n_in), ring_out(n_out_ports);
    for (;;) {
        // largest noutput whose forecast fits into what is left
        int n = 0;
        for (int cand = max_noutput; cand >= mult; cand -= mult) {
            gr_vector_int req(n_in, 0);
            blk.forecast(cand, req);
            bool ok = true;
            for (int p = 0; p < n_in; p++) {
                const size_t have = avail - consumed + hist;           // items visible behind the read pointer
                const size_t need = blk.fixed_rate() ? (size_t)cand + hist : (size_t)req[p] + hist;
                if (need > have) ok = false;
            }
            if (ok) { n = cand; break; }
        }
        if (n <= 0) break;
        gr_vector_int ninput(n_in, (int)(avail - consumed));
        // GNU Radio's buffers are fixed rings: a block sees the SAME memory call after call (which is what lets the
        // HIP runtime keep its pinning of a pageable buffer).  Model that: every call's input window is copied into a
        // per-port scratch buffer that lives across calls (the upstream block's write, not this block's work), and the
        // outputs are produced into persistent buffers and appended afterwards.
        gr_vector_const_void_star ip(n_in);
        for (int p = 0; p < n_in; p++) {
            const size_t isz = (size_t)blk.input_signature()->sizeof_stream_item(p);
            gr_vector_int req(n_in, 0);
            blk.forecast(n, req);
            const size_t span = ((blk.fixed_rate() ? (size_t)n : (size_t)req[p]) + hist) * isz;
            if (ring_in[p].size() < span) ring_in[p].resize(span);
            std::memcpy(ring_in[p].data(), in[p].data() + consumed * isz, span);
            ip[p] = ring_in[p].data();
        }
        gr_vector_void_star op(n_out_ports);
        for (int p = 0; p < n_out_ports; p++) {
            if (ring_out[p].size() < (size_t)n * out[p].item_size) ring_out[p].resize((size_t)n * out[p].item_size);
            op[p] = ring_out[p].data();
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int produced = blk.general_work(n, ninput, ip, op);
        if (work_seconds) *work_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (produced < 0) throw std::runtime_error(blk.name() + ": work() failed");
        for (int p = 0; p < n_out_ports; p++)
            out[p].bytes.insert(out[p].bytes.end(), ring_out[p].begin(), ring_out[p].begin() + (size_t)produced * out[p].item_size);
        if (blk.last_consumed() <= 0) break;
        consumed += (size_t)blk.last_consumed();
    }
    return out;
}

}  // namespace lite
}  // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
}  // namespace gnuradio
