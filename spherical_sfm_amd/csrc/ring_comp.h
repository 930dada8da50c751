// spherical_sfm_amd -- one long camera ring laid out in its own circular order (ba_flatten.h: band_plan; band_sub.h: sub_build; band_ring.h).  Host-only, no HIP.
#pragma once
#include <vector>
namespace ssfm {
struct RingComp {
    int comp = 0;                                 // component index (comp_ptr)
    int b = 0;                                    // half-width of the periodic band = rows of a separator, in band blocks
    int rows = 0;                                 // block rows of the ring (without the copy slot)
    std::vector<int> arc_len;                     // block rows of A_0 .. A_{m-1}
};
}  // namespace ssfm
