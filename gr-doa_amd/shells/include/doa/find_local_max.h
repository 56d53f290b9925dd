// gr::doa::find_local_max — public block interface, same factory signature as the reference
// (reference include/doa/find_local_max.h:43-57).  Implementation: HIP kernel behind libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// vlen vector_len floats in; port 0: the num_max_vals largest local maxima, port 1: their
// locations on the x axis [x_min, x_max), each sorted descending.
class DOA_API find_local_max : virtual public gr::sync_block
{
public:
    typedef DOA_SPTR<find_local_max> sptr;
    static sptr make(int num_max_vals, int vector_len, float x_min, float x_max);
};

}  // namespace doa
}  // namespace gr
