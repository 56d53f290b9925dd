// gr::doa::calibrate_lin_array — public block interface, same factory signature as the reference
// (reference include/doa/calibrate_lin_array.h).  Implementation: HIP kernel behind libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// vlen num_ant_ele^2 complex covariance items (one pilot source at pilot_angle degrees) in,
// vlen num_ant_ele complex antenna-response estimates out.
class DOA_API calibrate_lin_array : virtual public gr::sync_block
{
public:
    typedef DOA_SPTR<calibrate_lin_array> sptr;
    static sptr make(float norm_spacing, int num_ant_ele, float pilot_angle);
};

}  // namespace doa
}  // namespace gr
