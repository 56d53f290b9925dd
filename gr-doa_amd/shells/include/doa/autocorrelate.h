// gr::doa::autocorrelate — public block interface, same factory signature as the reference
// (reference include/doa/autocorrelate.h:43-57).  Implementation: HIP kernels behind libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// N complex streams in, one stream of column-major N x N sample-covariance matrices out
// (sliding window of snapshot_size samples advancing by snapshot_size - overlap_size;
// avg_method 1 = forward-backward averaging).
class DOA_API autocorrelate : virtual public gr::block
{
public:
    typedef DOA_SPTR<autocorrelate> sptr;
    static sptr make(int inputs, int snapshot_size, int overlap_size, int avg_method);
};

}  // namespace doa
}  // namespace gr
