// C ABI, layer 3b host helper (include/opusgpu.h): K1's work list.  No GPU calls, no parser state: shared by the library and
// by the CPU emulation fixture of the kernels (tests/emu).
#include <cstdint>
#include <vector>
#include "leaf_sort.hpp"

extern "C" {

uint32_t opusgpu_sort_pvq_leaves(const OgCeltHdr *hdr, uint32_t nframes, const OgCeltLeaf *leaves, OgLeafRef *out, uint32_t cap,
                                 uint32_t class_counts[4])
{
    using namespace og;
    if (class_counts) class_counts[0] = class_counts[1] = class_counts[2] = class_counts[3] = 0;
    if (!hdr || !leaves || !out || nframes > OG_REF_FRAME_MASK) return 0;
    std::vector<uint32_t> count(K1_NKEYS + 1, 0);           // counting sort by k1_key, largest first (leaf_sort.hpp)
    uint32_t per_n[256] = {0};
    for (uint32_t f = 0; f < nframes; f++)
        for (uint32_t l = 0; l < hdr[f].nleaves; l++) {
            const OgCeltLeaf &lf = leaves[hdr[f].leaf_begin + l];
            if (k1_wanted(hdr[f], lf)) { count[k1_key(lf)]++; per_n[lf.n]++; }
        }
    uint32_t total = 0;
    for (int k = K1_NKEYS - 1; k >= 0; k--) { const uint32_t c = count[k]; count[k] = total; total += c; }   // count[] -> start[]
    if (total > cap) return 0;
    if (class_counts)
        for (int n = 0; n < 256; n++) class_counts[k1_class(n)] += per_n[n];
    for (uint32_t f = 0; f < nframes; f++)
        for (uint32_t l = 0; l < hdr[f].nleaves; l++) {
            const uint32_t idx = hdr[f].leaf_begin + l;
            const OgCeltLeaf &lf = leaves[idx];
            if (k1_wanted(hdr[f], lf)) out[count[k1_key(lf)]++] = k1_ref(hdr[f], idx, f);
        }
    return total;
}

}  // extern "C"
