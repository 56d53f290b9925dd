// gr::doa::MUSIC_lin_array — public block interface, same factory signature as the reference
// (reference include/doa/MUSIC_lin_array.h:43-57).  Implementation: HIP kernels behind libdoa_hip.so.
#pragma once
#include <doa/api.h>

namespace gr {
namespace doa {

// vlen num_ant_ele^2 complex covariance items in, vlen pspectrum_len float pseudo-spectra (dB,
// peak = 0) out, for a uniform linear array with element spacing norm_spacing wavelengths.
class DOA_API MUSIC_lin_array : virtual public gr::sync_block
{
public:
    typedef DOA_SPTR<MUSIC_lin_array> sptr;
    static sptr make(float norm_spacing, int num_targets, int num_ant_ele, int pspectrum_len);
};

}  // namespace doa
}  // namespace gr
