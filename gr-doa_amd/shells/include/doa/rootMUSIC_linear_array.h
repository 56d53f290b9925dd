// gr::doa::rootMUSIC_linear_array — public block interface, same factory signature as the reference
// (reference include/doa/rootMUSIC_linear_array.h:41-55).  Implementation: HIP kernels behind
// libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// vlen num_ant_ele^2 complex covariance items in, vlen num_targets float angles (degrees,
// ascending) out.
class DOA_API rootMUSIC_linear_array : virtual public gr::sync_block
{
public:
    typedef DOA_SPTR<rootMUSIC_linear_array> sptr;
    static sptr make(float norm_spacing, int num_targets, int num_ant_ele);
};

}  // namespace doa
}  // namespace gr
