// music_scan.hpp — interface between music.hip (dispatch, C ABI) and the per-polynomial-size
// translation units of the K4 scan kernels (music_scan_inst.hip compiled once per compiled size, so the
// ~250 kernel instantiations build in parallel).
#pragma once
#include "kernels.hpp"

namespace doa {

struct ScanPeakArgs {           // optional fused K5, and the lean kernel's record pointer
    const float *xaxis = nullptr;
    float *val = nullptr, *loc = nullptr;
    int M = 0;
    bool store = true;          // false: nobody wants the spectrum (angles-only pipeline call); d_spec is scratch then
    const void *cheb = nullptr; // N <= 4, double: the pre-transformed records of launch_music_evd (kChebRecord doubles per item)
};

// returns true when the fused peak pick ran (fast paths only)
template <int N> bool launch_scan_n(const MusicTables &t, int bits, int n_items, const void *d_coef, void *d_spec,
                                    void *d_q, const ScanPeakArgs &pk, hipStream_t st);

#define DOA_SCAN_SIZES(X) X(2) X(3) X(4) X(6) X(8) X(12) X(16)
#define DOA_SCAN_EXTERN(n)                                                                                         \
    extern template bool launch_scan_n<n>(const MusicTables &, int, int, const void *, void *, void *,             \
                                          const ScanPeakArgs &, hipStream_t);

}  // namespace doa
