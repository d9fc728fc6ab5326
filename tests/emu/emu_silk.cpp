// TEST FIXTURE: g++ build of the SILK kernel bodies (opus_amd/csrc/hip/silk_synth.h); see emu_celt.cpp.
#include <cstring>
#include <vector>
#include "../../opus_amd/csrc/hip/celt_frame_types.h"
#include "../../opus_amd/csrc/hip/silk_synth.h"

void og_fill_silk_tables(OgSilkTables *t);

extern "C" size_t emu_silk_state_floats(uint32_t nstreams)
{
    const size_t ns = 2 * size_t(nstreams), nst = (ns + 63) / 64 * 64;
    return (size_t(OG_SILK_OUT_ROWS) + 16 + OG_SILK_RES_ROWS) * nst + OG_SILK_FRAME_STRIDE * ns + size_t(nstreams) * 64 + 4;
}

extern "C" int emu_launch_silk(const OgSilkFrame *frames, uint32_t nframes, float *state, uint32_t nstreams, float *pcm,
                               size_t pcm_stride, float *dbg_internal)
{
    static OgSilkTables T;
    static bool init = false;
    if (!init) { og_fill_silk_tables(&T); init = true; }
    OgSilkArrays A;
    const size_t ns = 2 * size_t(nstreams), nst = (ns + 63) / 64 * 64;
    float *p = state;
    A.out = p; p += size_t(OG_SILK_OUT_ROWS) * nst;
    A.lpc = p; p += 16 * nst;
    A.res = p; p += size_t(OG_SILK_RES_ROWS) * nst;
    A.rs_hist = p; p += size_t(nstreams) * 64;
    A.frame = (float *)(((uintptr_t)p + 15) & ~(uintptr_t)15);      // 16-byte aligned rows (the numpy buffer may not be)
    A.nslots = uint32_t(ns);
    A.nstreams = nstreams;
    for (uint32_t i = 0; i < 2 * nframes; i++) og_silk_channel(frames, nframes, A, i);
    static OgSilkOutLds L;
    for (uint32_t f = 0; f < nframes; f++) {
        std::memset(&L, 0xA5, sizeof L);
        og_silk_output(frames + f, A, &T, pcm, pcm_stride, L, dbg_internal ? dbg_internal + (size_t)f * 640 : nullptr);
    }
    return 0;
}
