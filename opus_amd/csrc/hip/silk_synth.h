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

// Per-batch SILK state in HBM.  All arrays are [row][nslots], nslots = 2 * nstreams.
struct OgSilkArrays {
    float *out;                       // [OG_SILK_OUT_ROWS][nslots]: rows 0..321 = clamped output history (silk/decoder.rs:1300
                                      // `output`), rows 322.. = the frame being synthesised
    float *lpc;                       // [16][nslots]: last `order` unclamped outputs (`lpc_history`, :1301), row 0 = oldest
    float *res;                       // [OG_SILK_RES_ROWS][nslots]: per-frame residual work array (:1713)
    float *rs_hist;                   // [nstreams][2][32]: resampler input history per output channel
    uint32_t nslots;
};

struct OgSilkTables {
    float rs_taps[3][OG_RS_TAPS + 3]; // 8/12/16 kHz -> 48 kHz low-pass prototypes
};

OG_DEV float og_clamp1(float v) { v = v > -1.f ? v : -1.f; return v < 1.f ? v : 1.f; }

// One channel's frame.  `ch` = this channel's record; `slot` = 2*stream + channel.
// Every loop runs in blocks: a block's loads (which do not depend on the block's own results) are issued together, then
// the samples are computed in the reference's order, then stored -- the recursions are unchanged, but a lane waits for
// HBM once per block instead of once per sample.  Block sizes divide every SILK subframe (40 / 60 / 80 samples).
#define OG_SILK_BLK 20                // LPC synthesis, excitation, history shift
#define OG_SILK_LTP_BLK 10            // long-term predictor: needs lag - 2 >= block (SILK's shortest lag is 2 ms = 16 samples)
template <int ORDER>
OG_DEV void og_silk_channel_t(const OgSilkChan *__restrict__ ch, const OgSilkArrays &A, uint32_t slot)
{
    const size_t ns = A.nslots;
    float *__restrict__ out = A.out + slot;
    float *__restrict__ res = A.res + slot;
    float *__restrict__ lpcst = A.lpc + slot;
    const int subframes = ch->subframes, sf = ch->sf_size, f_size = subframes * sf;
    const bool voiced = (ch->flags & OG_S_VOICED) != 0, interp = (ch->flags & OG_S_INTERP) != 0;
    const bool factor4 = (ch->flags & OG_S_INTERP4) != 0;

    for (int i0 = 0; i0 < f_size; i0 += OG_SILK_BLK) {                       // :1637
        int32_t e[OG_SILK_BLK];
#pragma unroll
        for (int k = 0; k < OG_SILK_BLK; k++) e[k] = ch->excitation[i0 + k];
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
        for (int j0 = 0; j0 < sf; j0 += OG_SILK_BLK) {                       // LPC synthesis, :1864-1884
            float x[OG_SILK_BLK];
#pragma unroll
            for (int q = 0; q < OG_SILK_BLK; q++) x[q] = res[(size_t)(OG_SILK_RES_HIST + i * sf + j0 + q) * ns];
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
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) lpcst[(size_t)(15 - k) * ns] = h[k];
    for (int i0 = 0; i0 < OG_SILK_HIST; i0 += 14) {                          // :1891-1898 (322 = 23 * 14; reads run ahead of writes)
        float v[14];
#pragma unroll
        for (int q = 0; q < 14; q++) v[q] = out[(size_t)(i0 + q + f_size) * ns];
#pragma unroll
        for (int q = 0; q < 14; q++) out[(size_t)(i0 + q) * ns] = v[q];
    }
}

// Dispatch for lane `idx` of a launch over 2*nframes (frame, channel) pairs.
OG_DEV void og_silk_channel(const OgSilkFrame *frames, uint32_t nframes, const OgSilkArrays &A, uint32_t idx)
{
    if (idx >= 2 * nframes) return;
    const OgSilkFrame *f = frames + (idx >> 1);
    const int c = idx & 1;
    const OgSilkChan *ch = &f->ch[c];
    const uint32_t slot = 2 * f->stream + c;
    const size_t ns = A.nslots;
    if (ch->flags & OG_S_RESET) {                                            // SilkFrame::flush, silk/decoder.rs:1642-1662
        float *out = A.out + slot, *lp = A.lpc + slot;
        for (int i = 0; i < OG_SILK_HIST; i++) out[(size_t)i * ns] = 0.f;
        for (int i = 0; i < 16; i++) lp[(size_t)i * ns] = 0.f;
    }
    if (!(ch->flags & OG_S_CODED)) return;
    if (ch->order == 16) og_silk_channel_t<16>(ch, A, slot);
    else og_silk_channel_t<10>(ch, A, slot);
}

// ------------------------------------------------------------------------------------------------
// K5 + K6, one workgroup per frame.  LDS: x[2][32 + 320] input samples at the internal rate.
struct OgSilkOutLds {
    float x[2][32 + 320];
};

OG_DEV void og_silk_output(const OgSilkFrame *f, const OgSilkArrays &A, const OgSilkTables *T, float *pcm_base,
                           size_t pcm_stride, OgSilkOutLds &L, float *dbg_internal /* [2][320] or null */)
{
    const size_t ns = A.nslots;
    const int fs = f->f_size, bw = f->bandwidth;
    const bool stereo = (f->flags & OG_SF_STEREO) != 0, stereo_out = (f->flags & OG_SF_STEREO_OUT) != 0;
    const int CC = stereo_out ? 2 : 1;
    const float *mid = A.out + 2 * f->stream + (size_t)(OG_SILK_HIST - fs - 2) * ns;    // silk/decoder.rs:2040
    const float *side = A.out + 2 * f->stream + 1 + (size_t)(OG_SILK_HIST - fs - 1) * ns;
    float *hist = A.rs_hist + (size_t)f->stream * 64;
    const bool rs_reset = (f->flags & OG_SF_RESAMPLER_RESET) != 0;

    OG_LANES_BEGIN(tid)
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
            const float m0 = mid[(size_t)i * ns], m1 = mid[(size_t)(i + 1) * ns], m2 = mid[(size_t)(i + 2) * ns];
            const float p0 = 0.25f * (m0 + 2.0f * m1 + m2);
            const float si0 = side[(size_t)i * ns] + i0 * p0;
            L.x[1][32 + i] = og_clamp1((1.0f + i1) * m1 + si0);              // right -> output channel 1
            L.x[0][32 + i] = og_clamp1((1.0f - i1) * m1 - si0);              // left  -> output channel 0
        }
    } else {                                                                 // mono copy, 2-sample delay, :2104-2113
        for (int i = tid; i < fs; i += OG_WG) {
            const float v = mid[(size_t)i * ns];
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
    // polyphase FIR: y[m] = sum_k h[m - k*Lf] x[k], k ascending (k < 0 from the history)
    const int Lf = bw == 0 ? 6 : bw == 1 ? 4 : 3;
    const float *hh = T->rs_taps[bw];
    float *pcm = pcm_base + (size_t)f->stream * pcm_stride + (size_t)f->pcm_off * CC;
    const int nout = fs * Lf;
    OG_LANES_BEGIN(tid)
    for (int q = tid; q < nout * CC; q += OG_WG) {
        const int m = q / CC, c = q - m * CC;
        const int k_hi = m / Lf;
        const int k_lo = (m - 2 * OG_RS_HALF + Lf - 1 + 64 * Lf) / Lf - 64;   // ceil((m - 72) / Lf)
        const float *x = &L.x[c][32];
        float acc = 0.f;
        for (int k = k_lo; k <= k_hi; k++) acc += hh[m - k * Lf] * x[k];
        pcm[q] = acc;
    }
    for (int q = tid; q < CC * 32; q += OG_WG) {
        const int c = q >> 5, i = q & 31;
        hist[c * 32 + i] = L.x[c][fs + i];                                  // last 32 inputs (fs >= 80 > 32)
    }
    OG_LANES_END
}
