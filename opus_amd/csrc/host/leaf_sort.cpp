// C ABI, layer 3b host helper (include/opusgpu.h): K1's work list.  No GPU calls, no parser state: shared by the library and
// by the CPU emulation fixture of the kernels (tests/emu).
#include <cstdint>
#include <vector>
#include "../../../include/opusgpu.h"

extern "C" {

uint32_t opusgpu_sort_pvq_leaves(const OgCeltHdr *hdr, uint32_t nframes, const OgCeltLeaf *leaves, OgLeafRef *out, uint32_t cap,
                                 uint32_t class_counts[4])
{
    if (class_counts) class_counts[0] = class_counts[1] = class_counts[2] = class_counts[3] = 0;
    // counting sort by (dimension N, pulses K), largest first: a wavefront of the K1 kernel then holds leaves of (nearly)
    // one size and pulse count, so its lanes run the same loop trip counts (cwrsi walks N positions and searches ~K/2
    // table entries; the rotation walks N).  Key = N * 64 + min(K, 63).
    if (!hdr || !leaves || !out || nframes > OG_REF_FRAME_MASK) return 0;
    enum { KQ = 64, NKEYS = 178 * KQ };
    std::vector<uint32_t> count(NKEYS + 1, 0);
    auto key_of = [](const OgCeltLeaf &lf) { return (uint32_t)lf.n * KQ + (lf.k < KQ ? lf.k : KQ - 1); };
    // K1's leaves: every PVQ leaf (K > 0, N > 1), and the zero-pulse leaves that share a (band, buffer) with one: K1 stores
    // zeros for those, so that the frame kernel can take the whole (band, buffer) from the scratch row and everything
    // else as zero without the rows ever being cleared
    auto wanted = [](const OgCeltHdr &h, const OgCeltLeaf &lf) {
        if (lf.meta & OG_L_N1) return false;
        if (lf.k) return true;
        if (lf.band >= OG_MAX_BANDS) return false;
        const OgCeltBand &bd = h.band[lf.band];
        return (lf.meta & OG_L_CH_Y) ? bd.ny > bd.fy : bd.nx > bd.fx;
    };
    uint32_t per_n[178] = {0};
    for (uint32_t f = 0; f < nframes; f++)
        for (uint32_t l = 0; l < hdr[f].nleaves; l++) {
            const OgCeltLeaf &lf = leaves[hdr[f].leaf_begin + l];
            if (wanted(hdr[f], lf)) { count[key_of(lf)]++; per_n[lf.n]++; }
        }
    uint32_t total = 0;
    for (int k = NKEYS - 1; k >= 0; k--) { const uint32_t c = count[k]; count[k] = total; total += c; }   // count[] -> start[]
    if (total > cap) return 0;
    if (class_counts)
        for (int n = 0; n < 178; n++) class_counts[n > 64 ? 0 : n > 24 ? 1 : n > 8 ? 2 : 3] += per_n[n];
    for (uint32_t f = 0; f < nframes; f++)
        for (uint32_t l = 0; l < hdr[f].nleaves; l++) {
            const uint32_t idx = hdr[f].leaf_begin + l;
            const OgCeltLeaf &lf = leaves[idx];
            if (wanted(hdr[f], lf)) { OgLeafRef &r = out[count[key_of(lf)]++]; r.leaf = idx; r.frame = f | (uint32_t)(hdr[f].spread & 3) << OG_REF_SPREAD_SHIFT; }
        }
    return total;
}

}  // extern "C"
