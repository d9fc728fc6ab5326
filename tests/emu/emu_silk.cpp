// TEST FIXTURE: g++ build of the SILK kernel bodies (opus_amd/csrc/hip/silk_synth.h); see emu_celt.cpp.
#include <cstring>
#include <vector>
#include "../../opus_amd/csrc/hip/celt_frame_types.h"
#include "../../opus_amd/csrc/hip/silk_synth.h"

void og_fill_silk_tables(OgSilkTables *t);

extern "C" size_t emu_silk_state_floats(uint32_t nstreams)
{
    return size_t(OG_SILK_SLOT) * 2 * size_t(nstreams) + size_t(nstreams) * 64 + 4;
}

extern "C" int emu_launch_silk(const OgSilkFrame *frames, uint32_t nframes, float *state, uint32_t nstreams, float *pcm,
                               size_t pcm_stride, float *dbg_internal)
{
    static OgSilkTables T;
    static bool init = false;
    if (!init) { og_fill_silk_tables(&T); init = true; }
    OgSilkArrays A;
    A.slots = state;
    A.rs_hist = state + size_t(OG_SILK_SLOT) * 2 * size_t(nstreams);
    A.nslots = 2 * nstreams;
    A.nstreams = nstreams;
    static OgSilkLds Ls;
    for (uint32_t i = 0; i < 2 * nframes; i += OG_SK_SLOTS) {
        std::memset(&Ls, 0xA5, sizeof Ls);             // LDS starts undefined on the device
        og_silk_synth_wave(frames, nframes, A, i, Ls);
    }
    static OgSilkOutLds L;
    for (uint32_t f = 0; f < nframes; f++) {
        std::memset(&L, 0xA5, sizeof L);
        og_silk_output(frames + f, A, &T, pcm, pcm_stride, L, dbg_internal ? dbg_internal + (size_t)f * 640 : nullptr);
    }
    return 0;
}
