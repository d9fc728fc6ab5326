// SILK synthesis, device side.
//
// K4 og_silk_channel: one lane = one (stream, channel).  The three recursions of the reference's
// SilkFrame::parse synthesis loop (lu-zero/opus src/silk/decoder.rs:1775-1898) -- LPC re-whitening of
// the output history (:1794-1828), the 5-tap long-term predictor (:1843-1860) and the order-10/16
// short-term LPC synthesis (:1864-1884) -- are sequential in time inside a channel and independent
// across channels, so streams map across lanes.  State and work arrays are struct-of-arrays in HBM,
// `[sample][slot]` with slot = 2*stream + channel: when the lanes of a wavefront walk the same sample
// index their accesses coalesce.  Arithmetic keeps the reference's operation order (no FMA
// contraction) so that the reference's golden PCM (silk/tests.rs) holds bit for bit.
//
// K5+K6 og_silk_output: one workgroup = one stream's frame: mid/side -> left/right
// (unmix_ms, silk/decoder.rs:2012-2061) or the mono copy with its 2-sample delay (:2104-2113), then the
// internal-rate -> 48 kHz polyphase FIR (absent from the reference; SURVEY a21, the build's own design)
// and the interleaved PCM store.
#pragma once
#include "wg.h"
#include "../../../include/opusgpu.h"

#ifndef OG_WG
#define OG_WG 256
#endif

#define OG_SILK_HIST 322              // silk/decoder.rs:692 LPC_HISTORY
#define OG_SILK_RES_HIST 290          // silk/decoder.rs:691: 288 + LTP_ORDER / 2
#define OG_SILK_OUT_ROWS (OG_SILK_HIST + 320)
#define OG_SILK_RES_ROWS (OG_SILK_RES_HIST + 320)
#define OG_RS_HALF 36
#define OG_RS_TAPS (2 * OG_RS_HALF + 1)

// Per-batch SILK state in HBM.  A slot = one channel of one stream: slot = channel * nstreams + stream (all mid channels,
// then all side channels: the 64 lanes of a wavefront -- consecutive streams, one channel -- are 64 consecutive slots).  The
// per-slot arrays are tiled by 64 slots: [tile][row][64], so a wavefront's rows are 256 bytes each and lie one behind the other
// (a wavefront walks ~1 300 rows per frame: with rows a whole batch apart every access was a TLB miss and an HBM page of its own).
struct OgSilkArrays {
    float *out;                       // [OG_SILK_OUT_ROWS][nslots]: rows 0..321 = clamped output history (silk/decoder.rs:1300
                                      // `output`), rows 322.. = the frame being synthesised
    float *lpc;                       // [16][nslots]: last `order` unclamped outputs (`lpc_history`, :1301), row 0 = oldest
    float *res;                       // [OG_SILK_RES_ROWS][nslots]: per-frame residual work array (:1713)
    float *rs_hist;                   // [nstreams][2][32]: resampler input history per output channel
    float *frame;                     // [nslots][328]: the slot's newest frame, contiguous (K4 -> K5 hand-over): [2], [3] = the two
                                      // samples before it (unmix_ms's look-back, silk/decoder.rs:2040), [4 ..] = the frame
    uint32_t nslots, nstreams;
};
#define OG_SILK_NS 64                 // row stride inside a tile
// column of `slot` in a tiled array of `rows` rows: element r at [r * OG_SILK_NS]
OG_DEV float *og_silk_col(float *arr, int rows, uint32_t slot) { return arr + ((size_t)(slot >> 6) * rows) * OG_SILK_NS + (slot & 63); }
#define OG_SILK_FRAME_STRIDE 328       // [0], [1] unused, [2], [3] look-back, [4 .. 324) frame, 16-byte aligned rows

struct alignas(16) OgI4 { int32_t x, y, z, w; };
struct alignas(16) OgF4v { float x, y, z, w; };
struct OgSilkTables {
    float rs_taps[3][OG_RS_TAPS + 3]; // 8/12/16 kHz -> 48 kHz low-pass prototypes
};

OG_DEV float og_clamp1(float v) { v = v > -1.f ? v : -1.f; return v < 1.f ? v : 1.f; }

// One channel's frame.  `ch` = this channel's record; `slot` = channel * nstreams + stream.
// Every loop runs in blocks: a block's loads (which do not depend on the block's own results) are issued together, then
// the samples are computed in the reference's order, then stored -- the recursions are unchanged, but a lane waits for
// HBM once per block instead of once per sample.  Block sizes divide every SILK subframe (40 / 60 / 80 samples).
// OG_SILK_BLK (LPC synthesis, excitation) and OG_SILK_LTP_BLK (long-term predictor and re-whitening: needs lag - 2 >= block,
// and SILK's shortest lag is 2 ms) are template parameters chosen per bandwidth (multiples of four that divide the subframe): NB 40 / 10, MB 20 / 20, WB 40 / 20.  The LPC
// loop also fetches its next block while it computes the current one (the computation of a block takes about as long as
// the fetch: with one wavefront per CU at 16 384 streams nothing else hides it).
template <int ORDER, int OG_SILK_BLK, int OG_SILK_LTP_BLK>
OG_DEV void og_silk_channel_t(const OgSilkChan *__restrict__ ch, const OgSilkArrays &A, uint32_t slot)
{
    constexpr size_t ns = OG_SILK_NS;
    float *__restrict__ out = og_silk_col(A.out, OG_SILK_OUT_ROWS, slot);
    float *__restrict__ res = og_silk_col(A.res, OG_SILK_RES_ROWS, slot);
    float *__restrict__ lpcst = og_silk_col(A.lpc, 16, slot);
    float *__restrict__ fr = A.frame + (size_t)slot * OG_SILK_FRAME_STRIDE;     // [2], [3] = look-back, [4 ..] = the frame (16-byte aligned)
    fr[2] = out[(size_t)(OG_SILK_HIST - 2) * ns];       // the two samples before this frame: the newest of the history
    fr[3] = out[(size_t)(OG_SILK_HIST - 1) * ns];
    const int subframes = ch->subframes, sf = ch->sf_size, f_size = subframes * sf;
    const bool voiced = (ch->flags & OG_S_VOICED) != 0, interp = (ch->flags & OG_S_INTERP) != 0;
    const bool factor4 = (ch->flags & OG_S_INTERP4) != 0;

    for (int i0 = 0; i0 < f_size; i0 += OG_SILK_BLK) {                       // :1637
        int32_t e[OG_SILK_BLK];
#pragma unroll
        for (int k = 0; k < OG_SILK_BLK; k += 4) {      // 16-byte loads: the record is 16-byte aligned (opusgpu_sideinfo.h)
            const OgI4 v = *(const OgI4 *)&ch->excitation[i0 + k];
            e[k] = v.x; e[k + 1] = v.y; e[k + 2] = v.z; e[k + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < OG_SILK_BLK; k++) res[(size_t)(OG_SILK_RES_HIST + i0 + k) * ns] = (float)e[k] / 8388608.0f;
    }

    float h[16];                      // h[0] = most recent unclamped output; always 16 deep: the order may change
#pragma unroll                        // between frames (bandwidth switch) and the reference keeps the full history
    for (int k = 0; k < 16; k++) h[k] = lpcst[(size_t)(15 - k) * ns];

    for (int i = 0; i < subframes; i++) {
        const float *cf = (i < 2 && interp) ? ch->lpc_interp : ch->lpc;
        float a[ORDER];
#pragma unroll
        for (int k = 0; k < ORDER; k++) a[k] = cf[k];
        const float gain = ch->gain[i];
        if (voiced) {
            const int lag = ch->pitch_lag[i];
            const int before = lag + 2;
            int end;
            float scale;
            if (i < 2 || factor4) { end = i * sf; scale = ch->ltpscale; }
            else { end = (i - 2) * sf; scale = 1.f; }
            if (before > end) {                                              // re-whiten the output history, :1794-1828
                const int start = OG_SILK_HIST + i * sf - before, cnt = before - end;
                const int start_res = OG_SILK_RES_HIST + i * sf - before;
                float w[ORDER];                                              // w[k] = out[start + n - 1 - k]
#pragma unroll
                for (int k = 0; k < ORDER; k++) w[k] = out[(size_t)(start - 1 - k) * ns];
                for (int n0 = 0; n0 < cnt; n0 += OG_SILK_LTP_BLK) {
                    float cur[OG_SILK_LTP_BLK], r[OG_SILK_LTP_BLK];
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++) cur[q] = n0 + q < cnt ? out[(size_t)(start + n0 + q) * ns] : 0.f;
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++) {
                        float sum = cur[q];
#pragma unroll
                        for (int k = 0; k < ORDER; k++) sum -= a[k] * w[k];
                        r[q] = og_clamp1(sum) * scale / gain;
#pragma unroll
                        for (int k = ORDER - 1; k > 0; k--) w[k] = w[k - 1];
                        w[0] = cur[q];
                    }
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++)
                        if (n0 + q < cnt) res[(size_t)(start_res + n0 + q) * ns] = r[q];
                }
            }
            if (end != 0) {                                                  // :1830-1841
                const float rescale = ch->gain[i - 1] / gain;
                const int lo = OG_SILK_RES_HIST + i * sf - end, hi = OG_SILK_RES_HIST + i * sf;     // end: a multiple of sf
                for (int n0 = lo; n0 < hi; n0 += OG_SILK_BLK) {
                    float v[OG_SILK_BLK];
#pragma unroll
                    for (int q = 0; q < OG_SILK_BLK; q++) v[q] = res[(size_t)(n0 + q) * ns];
#pragma unroll
                    for (int q = 0; q < OG_SILK_BLK; q++) res[(size_t)(n0 + q) * ns] = v[q] * rescale;
                }
            }
            const float t0 = ch->ltp_taps[i][0], t1 = ch->ltp_taps[i][1], t2 = ch->ltp_taps[i][2], t3 = ch->ltp_taps[i][3],
                        t4 = ch->ltp_taps[i][4];
            const int nlo = OG_SILK_RES_HIST + i * sf, nhi = OG_SILK_RES_HIST + (i + 1) * sf;
            if (lag - 2 >= OG_SILK_LTP_BLK) {                                // LTP, :1843-1860, a block at a time: sample n reads
                for (int n0 = nlo; n0 < nhi; n0 += OG_SILK_LTP_BLK) {        // res[n - lag - 2 .. n - lag + 2], all before the block
                    float x[OG_SILK_LTP_BLK], p[OG_SILK_LTP_BLK + 4];        // p[m] = res[n0 - lag - 2 + m]
#pragma unroll
                    for (int m = 0; m < OG_SILK_LTP_BLK + 4; m++) p[m] = res[(size_t)(n0 - lag - 2 + m) * ns];
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++) x[q] = res[(size_t)(n0 + q) * ns];
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++) {
                        float sum = x[q];
                        sum += t0 * p[q + 4];
                        sum += t1 * p[q + 3];
                        sum += t2 * p[q + 2];
                        sum += t3 * p[q + 1];
                        sum += t4 * p[q];
                        x[q] = sum;
                    }
#pragma unroll
                    for (int q = 0; q < OG_SILK_LTP_BLK; q++) res[(size_t)(n0 + q) * ns] = x[q];
                }
            } else {                                                         // (a lag the bitstream cannot code: sample by sample)
                for (int n = nlo; n < nhi; n++) {
                    const float *r = res + (size_t)(n - lag + 2) * ns;
                    float sum = res[(size_t)n * ns];
                    sum += t0 * r[0];
                    sum += t1 * *(r - ns);
                    sum += t2 * *(r - 2 * ns);
                    sum += t3 * *(r - 3 * ns);
                    sum += t4 * *(r - 4 * ns);
                    res[(size_t)n * ns] = sum;
                }
            }
        }
        float xn[OG_SILK_BLK];                                               // LPC synthesis, :1864-1884: the block in flight
#pragma unroll
        for (int q = 0; q < OG_SILK_BLK; q++) xn[q] = res[(size_t)(OG_SILK_RES_HIST + i * sf + q) * ns];
        for (int j0 = 0; j0 < sf; j0 += OG_SILK_BLK) {
            float x[OG_SILK_BLK];
#pragma unroll
            for (int q = 0; q < OG_SILK_BLK; q++) x[q] = xn[q];
            if (j0 + OG_SILK_BLK < sf) {
#pragma unroll
                for (int q = 0; q < OG_SILK_BLK; q++) xn[q] = res[(size_t)(OG_SILK_RES_HIST + i * sf + j0 + OG_SILK_BLK + q) * ns];
            }
#pragma unroll
            for (int q = 0; q < OG_SILK_BLK; q++) {
                float sum = x[q] * gain;
#pragma unroll
                for (int k = 0; k < ORDER; k++) sum += a[k] * h[k];
#pragma unroll
                for (int k = 15; k > 0; k--) h[k] = h[k - 1];
                h[0] = sum;
                x[q] = og_clamp1(sum);
            }
#pragma unroll
            for (int q = 0; q < OG_SILK_BLK; q++) out[(size_t)(OG_SILK_HIST + i * sf + j0 + q) * ns] = x[q];
#pragma unroll
            for (int q = 0; q < OG_SILK_BLK; q += 4)    // and, contiguous per slot, for the output kernel (16-byte aligned: subframe
                *(OgF4v *)&fr[4 + i * sf + j0 + q] = OgF4v{x[q], x[q + 1], x[q + 2], x[q + 3]};     // and block sizes are multiples of four)
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) lpcst[(size_t)(15 - k) * ns] = h[k];
    for (int i0 = 0; i0 < OG_SILK_HIST; i0 += 46) {                          // :1891-1898 (322 = 7 * 46; reads run ahead of writes)
        float v[46];
#pragma unroll
        for (int q = 0; q < 46; q++) v[q] = out[(size_t)(i0 + q + f_size) * ns];
#pragma unroll
        for (int q = 0; q < 46; q++) out[(size_t)(i0 + q) * ns] = v[q];
    }
}

// Dispatch for lane `idx` of a launch over 2*nframes (frame, channel) pairs: the mid channels of all frames first, then the
// side channels, so that the wavefronts of the second half leave at once where the streams are mono.
OG_DEV void og_silk_channel(const OgSilkFrame *frames, uint32_t nframes, const OgSilkArrays &A, uint32_t idx)
{
    if (idx >= 2 * nframes) return;
    const int c = idx >= nframes;
    const OgSilkFrame *f = frames + (c ? idx - nframes : idx);
    const OgSilkChan *ch = &f->ch[c];
    const uint32_t slot = c * A.nstreams + f->stream;
    constexpr size_t ns = OG_SILK_NS;
    if (ch->flags & OG_S_RESET) {                                            // SilkFrame::flush, silk/decoder.rs:1642-1662
        float *out = og_silk_col(A.out, OG_SILK_OUT_ROWS, slot), *lp = og_silk_col(A.lpc, 16, slot);
        for (int i = 0; i < OG_SILK_HIST; i++) out[(size_t)i * ns] = 0.f;
        for (int i = 0; i < 16; i++) lp[(size_t)i * ns] = 0.f;
    }
    if (!(ch->flags & OG_S_CODED)) {
        // no frame of this channel (mid-only, silk/decoder.rs:2093-2099): what unmix_ms reads of it is its history as it stands
        if (f->flags & OG_SF_STEREO) {
            const float *out = og_silk_col(A.out, OG_SILK_OUT_ROWS, slot);
            float *fr = A.frame + (size_t)slot * OG_SILK_FRAME_STRIDE;
            const int fs = f->f_size;
            for (int i = 0; i < fs + 2; i++) fr[2 + i] = out[(size_t)(OG_SILK_HIST - fs - 2 + i) * ns];
        }
        return;
    }
    if (ch->order == 16) og_silk_channel_t<16, 40, 20>(ch, A, slot);         // WB: subframes of 80 samples, lag >= 32
    else if (ch->sf_size == 60) og_silk_channel_t<10, 20, 20>(ch, A, slot);  // MB: 60, lag >= 24
    else og_silk_channel_t<10, 40, 10>(ch, A, slot);                         // NB: 40, lag >= 16
}

// ------------------------------------------------------------------------------------------------
// K5 + K6, one workgroup per frame.  LDS: x[2][32 + 320] input samples at the internal rate.
struct OgSilkOutLds {
    float x[2][32 + 320];
    float h[OG_RS_TAPS + 3];          // the frame's low-pass prototype
};

template <int LF>
OG_DEV void og_silk_fir(const float *hh, const OgSilkOutLds &L, float *pcm, int fs, int CC, int tid)
{
    const int nout = fs * LF;
    for (int q = tid; q < nout * CC; q += OG_WG) {
        const int m = CC == 2 ? q >> 1 : q, c = CC == 2 ? q & 1 : 0;
        const int k_hi = m / LF;
        const int k_lo = (m - 2 * OG_RS_HALF + LF - 1 + 64 * LF) / LF - 64;   // ceil((m - 72) / LF)
        const float *x = &L.x[c][32];
        float acc = 0.f;
        for (int k = k_lo; k <= k_hi; k++) acc += hh[m - k * LF] * x[k];
        pcm[q] = acc;
    }
}

OG_DEV void og_silk_output(const OgSilkFrame *f, const OgSilkArrays &A, const OgSilkTables *T, float *pcm_base,
                           size_t pcm_stride, OgSilkOutLds &L, float *dbg_internal /* [2][320] or null */)
{
    const int fs = f->f_size, bw = f->bandwidth;
    const bool stereo = (f->flags & OG_SF_STEREO) != 0, stereo_out = (f->flags & OG_SF_STEREO_OUT) != 0;
    const int CC = stereo_out ? 2 : 1;
    // the frame as K4 left it, contiguous per slot: mid[i] = two samples behind the frame's sample i, side[i] = one behind
    // (silk/decoder.rs:2040: output[in_start - 2 ..] / [in_start - 1 ..])
    const float *mid = A.frame + (size_t)f->stream * OG_SILK_FRAME_STRIDE + 2;
    const float *side = A.frame + (size_t)(A.nstreams + f->stream) * OG_SILK_FRAME_STRIDE + 3;
    float *hist = A.rs_hist + (size_t)f->stream * 64;
    const bool rs_reset = (f->flags & OG_SF_RESAMPLER_RESET) != 0;

    OG_LANES_BEGIN(tid)
    for (int q = tid; q < OG_RS_TAPS + 3; q += OG_WG) L.h[q] = T->rs_taps[bw][q];
    for (int q = tid; q < CC * 32; q += OG_WG) {
        const int c = q >> 5, i = q & 31;
        L.x[c][i] = rs_reset ? 0.f : hist[c * 32 + i];
    }
    if (stereo && stereo_out) {                                              // unmix_ms, silk/decoder.rs:2012-2061
        const float w0 = f->w0, w1 = f->w1, w0p = f->prev_w0, w1p = f->prev_w1;
        const int n1 = bw == 0 ? 64 : bw == 1 ? 96 : 128;
        const float w0d = (w0 - w0p) / (float)n1, w1d = (w1 - w1p) / (float)n1;
        for (int i = tid; i < fs; i += OG_WG) {
            float i0 = w0, i1 = w1;
            if (i < n1) { i0 = w0p + (float)i * w0d; i1 = w1p + (float)i * w1d; }
            const float m0 = mid[i], m1 = mid[i + 1], m2 = mid[i + 2];
            const float p0 = 0.25f * (m0 + 2.0f * m1 + m2);
            const float si0 = side[i] + i0 * p0;
            L.x[1][32 + i] = og_clamp1((1.0f + i1) * m1 + si0);              // right -> output channel 1
            L.x[0][32 + i] = og_clamp1((1.0f - i1) * m1 - si0);              // left  -> output channel 0
        }
    } else {                                                                 // mono copy, 2-sample delay, :2104-2113
        for (int i = tid; i < fs; i += OG_WG) {
            const float v = mid[i];
            L.x[0][32 + i] = v;
            L.x[1][32 + i] = v;
        }
    }
    OG_LANES_END

    if (dbg_internal) {
        OG_LANES_BEGIN(tid)
        for (int q = tid; q < 2 * fs; q += OG_WG) dbg_internal[(q >= fs) * 320 + (q >= fs ? q - fs : q)] = L.x[q >= fs][32 + (q >= fs ? q - fs : q)];
        OG_LANES_END
    }
    // polyphase FIR: y[m] = sum_k h[m - k*Lf] x[k], k ascending (k < 0 from the history); Lf = 48 kHz / internal rate as a
    // compile-time constant, so that the index divisions are multiplications
    float *pcm = pcm_base + (size_t)f->stream * pcm_stride + (size_t)f->pcm_off * CC;
    OG_LANES_BEGIN(tid)
    if (bw == 0) og_silk_fir<6>(L.h, L, pcm, fs, CC, tid);
    else if (bw == 1) og_silk_fir<4>(L.h, L, pcm, fs, CC, tid);
    else og_silk_fir<3>(L.h, L, pcm, fs, CC, tid);
    for (int q = tid; q < CC * 32; q += OG_WG) {
        const int c = q >> 5, i = q & 31;
        hist[c * 32 + i] = L.x[c][fs + i];                                  // last 32 inputs (fs >= 80 > 32)
    }
    OG_LANES_END
}
