// music_scan_inst.hip — one translation unit per compiled polynomial size of the K4 scan
// (hipcc -DDOA_SCAN_N=<2|3|4|6|8|12|16>, see the Makefile).
#include "music_scan_impl.hpp"

#ifndef DOA_SCAN_N
#error "compile with -DDOA_SCAN_N=<size>"
#endif

namespace doa {
template bool launch_scan_n<DOA_SCAN_N>(const MusicTables &, int, int, const void *, void *, void *, const ScanPeakArgs &,
                                        hipStream_t);
}  // namespace doa
