// gr::doa::antenna_correction — public block interface, same factory signature as the reference
// (reference include/doa/antenna_correction.h:44-56).  Implementation: HIP kernel behind libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// num_ant_ele complex streams in and out; stream k is multiplied by (1/gain_k) * exp(-j phase_k) read
// from config_filename (one "gain phase" pair per line).
class DOA_API antenna_correction : virtual public gr::sync_block
{
public:
    typedef DOA_SPTR<antenna_correction> sptr;
    static sptr make(int num_ant_ele, char *config_filename);
};

}  // namespace doa
}  // namespace gr
