// K1's work list (hip/celt_launch.hip: og_pvq_all_kernel): which leaves it takes and the key they are sorted by.  Shared by
// opusgpu_sort_pvq_leaves (leaf_sort.cpp) and the batch engine's threaded form of the same sort (engine.cpp).
#pragma once
#include <cstdint>
#include "../../../include/opusgpu.h"

namespace og {

// counting-sort key (dimension N, pulses K), largest first: a wavefront of the K1 kernel then holds leaves of (nearly) one
// size and pulse count, so its lanes run the same loop trip counts (cwrsi walks N positions and searches ~K/2 table
// entries; the rotation walks N)
enum { K1_KQ = 64, K1_NKEYS = 256 * K1_KQ };      // every value of the leaf record's n byte has a key
inline uint32_t k1_key(const OgCeltLeaf &lf) { return uint32_t(lf.n) * K1_KQ + (lf.k < K1_KQ ? lf.k : K1_KQ - 1); }
inline int k1_class(int n) { return n > 64 ? 0 : n > 24 ? 1 : n > 8 ? 2 : 3; }

// K1's leaves: every PVQ leaf (K > 0, N > 1), and the zero-pulse leaves that share a (band, buffer) with one: K1 stores
// zeros for those, so that the frame kernel can take the whole (band, buffer) from the scratch row and everything else as
// zero without the rows ever being cleared
inline bool k1_wanted(const OgCeltHdr &h, const OgCeltLeaf &lf)
{
    if (lf.meta & OG_L_N1) return false;
    if (lf.k) return true;
    if (lf.band >= OG_MAX_BANDS) return false;
    const OgCeltBand &bd = h.band[lf.band];
    return (lf.meta & OG_L_CH_Y) ? bd.ny > bd.fy : bd.nx > bd.fx;
}
inline OgLeafRef k1_ref(const OgCeltHdr &h, uint32_t leaf, uint32_t frame)
{
    OgLeafRef r;
    r.leaf = leaf;
    r.frame = frame | uint32_t(h.spread & 3) << OG_REF_SPREAD_SHIFT;
    return r;
}

}  // namespace og
