// TEST FIXTURE: g++ build of the CELT kernel body (opus_amd/csrc/hip/celt_frame.h) in which a workgroup's
// lanes run one after the other (see hip/wg.h).  It lets the CPU-only CI exercise the kernel logic against
// the oracle.  It is not part of libopusgpu.so and nothing under opus_amd/ loads it.
#include <cstring>
#include <vector>
#include "../../opus_amd/csrc/hip/celt_frame.h"
#include "../../opus_amd/csrc/host/celt_tables.hpp"

extern "C" int emu_launch_celt(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                               OgCeltState *state, float *pcm, size_t pcm_stride, const OgCeltDebug *dbg)
{
    static OgCeltTables T;
    static bool init = false;
    if (!init) { og_fill_celt_tables(&T); init = true; }
    static OgCeltLds L;
    (void)frames_per_group;
    for (uint32_t f = 0; f < nframes; f++) {
        std::memset(&L, 0xA5, sizeof L);               // LDS starts undefined on the device
        OgCeltDebug d = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (dbg) {
            d.pulses = dbg->pulses ? dbg->pulses + (size_t)f * 1920 : nullptr;
            d.norm = dbg->norm ? dbg->norm + (size_t)f * 1920 : nullptr;
            d.freq = dbg->freq ? dbg->freq + (size_t)f * 1920 : nullptr;
            d.collapse = dbg->collapse ? dbg->collapse + (size_t)f * 42 : nullptr;
        }
        og_celt_frame(hdr + f, leaves, &T, state, pcm, pcm_stride, L, d);
    }
    return 0;
}
extern "C" size_t emu_lds_bytes(void) { return sizeof(OgCeltLds); }
