// SILK synthesis, device side.
//
// K4 og_silk_synth_wave: the reference's SilkFrame::parse synthesis loop (lu-zero/opus src/silk/decoder.rs:1775-1898) --
// LPC re-whitening of the output history (:1794-1828), the 5-tap long-term predictor (:1843-1860) and the order-10/16
// short-term LPC synthesis (:1864-1884) -- for eight (stream, channel) pairs per wavefront, their working sets in LDS; only the
// LPC synthesis recursion runs on one lane per channel, the rest is dealt over eight lanes per channel.  Arithmetic keeps the
// reference's operation order per sample (no FMA contraction) so that the reference's golden PCM (silk/tests.rs) holds bit for bit.
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
#define OG_RS_HALF 36
#define OG_RS_TAPS (2 * OG_RS_HALF + 1)

// Per-batch SILK state in HBM.  A slot = one channel of one stream: slot = channel * nstreams + stream (all mid channels, then
// all side channels).  A slot's state is contiguous: hist[324] (the last 322 clamped outputs, silk/decoder.rs:1300 `output`,
// oldest first; two floats of padding) + lpc[16] (the last 16 unclamped outputs, `lpc_history` :1301, oldest first).  The newest
// frame is the end of hist: that is where the output kernel reads it (and its two samples of look-back, unmix_ms :2040).
struct OgSilkArrays {
    float *slots;                     // [nslots][OG_SILK_SLOT]
    float *rs_hist;                   // [nstreams][2][32]: resampler input history per output channel
    uint32_t nslots, nstreams;
};
#define OG_SILK_SLOT 340              /* floats per slot: 322 + 2 + 16; 16-byte aligned rows */
OG_DEV float *og_silk_slot(const OgSilkArrays &A, uint32_t slot) { return A.slots + (size_t)slot * OG_SILK_SLOT; }

struct alignas(16) OgI4 { int32_t x, y, z, w; };
struct alignas(16) OgF4v { float x, y, z, w; };
struct OgSilkTables {
    float rs_taps[3][OG_RS_TAPS + 3]; // 8/12/16 kHz -> 48 kHz low-pass prototypes
};

OG_DEV float og_clamp1(float v) { v = v > -1.f ? v : -1.f; return v < 1.f ? v : 1.f; }

// K4.  One wavefront synthesises OG_SK_SLOTS channels side by side, OG_SK_SUB lanes each, with the channels' working sets --
// output history + the frame (642 samples) and the residual work array (610) -- in LDS.  Of the reference's loops
// (silk/decoder.rs:1775-1898) only the LPC synthesis is a recursion over the samples (17 dependent operations per sample whose
// order cannot change without changing bits): one lane per channel runs it, eight channels per wavefront.  Everything else is
// dealt over the channel's eight lanes sample by sample, each sample's own operations in the reference's order (so the bits are
// the reference's): excitation conversion (:1637), LPC re-whitening of the output history (an FIR over known samples,
// :1794-1828), residual rescaling (:1830-1841), the 5-tap long-term predictor (:1843-1860; sample n reads
// res[n - lag - 2 .. n - lag + 2], so lag - 2 samples are independent at a time) and the history shift (:1891-1898: the
// newest 322 samples go back to HBM, nothing moves).  No barrier anywhere: a wavefront's LDS operations complete in order.
// (Round 2: one lane per channel for everything, state and work arrays in HBM: 230 us per 16 384 mono frames, three SIMDs in
// four idle and every lane waiting on a memory round trip per block of samples.)
#define OG_SK_SLOTS 8
#define OG_SK_SUB 8
#define OG_SK_OUTS 645                /* row strides: odd, so that the eight channels' rows start on different LDS banks */
#define OG_SK_RESS 611
struct OgSilkLds {
    float out[OG_SK_SLOTS][OG_SK_OUTS];       // [0, 322) history, [322, 322 + f_size) the frame
    float res[OG_SK_SLOTS][OG_SK_RESS];       // [0, 290) residual history (rebuilt by the re-whitening), [290, ..) the frame's residual
};
struct OgSkItem {                     // a lane's channel (the same in the eight lanes of a group)
    const OgSilkChan *ch;
    float *st;                        // the slot's state in HBM
    int on;                           // 0 nothing to do, 1 synthesise
    int reset;
    int subframes, sf, voiced, interp, factor4;
};
struct OgSkCoef { float a[16]; };
struct OgSkParam {                    // the channel's parameters, fetched once (every lane of the group holds them): read in place they
    float gain[4], ltpscale;          // cost a memory round trip per region -- with one wavefront per SIMD nothing hides those
    int lag[4];
    float taps[4][5];
    float lpc[16], lpci[16];
};

template <int ORDER>                  // 16, or 10 when no channel of the wavefront is wideband (coefficients past a channel's order are zero)
OG_DEV void og_silk_subframes(OgSilkLds &L, OG_WAVE_REF(OgSkItem, itv), OG_WAVE_REF(OgSkCoef, hv), OG_WAVE_REF(OgSkParam, pv))
{
    OG_WAVE_VAR(OgSkCoef, av);
    for (int i = 0; i < 4; i++) {
        // ---- the subframe's coefficients; re-whitening and rescaling (voiced channels)
        OG_WAVE_BEGIN(lane)
        const OgSkItem it = OG_WV(itv);
        const int g = lane >> 3, sub = lane & 7;
        if (it.on && i < it.subframes) {
            const OgSkParam &pm = OG_WV(pv);
            const int sf = it.sf;
            const bool ip = i < 2 && it.interp;
#pragma unroll
            for (int k = 0; k < ORDER; k++) OG_WV(av).a[k] = ip ? pm.lpci[k] : pm.lpc[k];     // (zero past the channel's order)
            if (it.voiced) {
                float *out = L.out[g], *res = L.res[g];
                const float gain = pm.gain[i];
                const int lag = pm.lag[i], before = lag + 2;
                int end;
                float scale;
                if (i < 2 || it.factor4) { end = i * sf; scale = pm.ltpscale; }
                else { end = (i - 2) * sf; scale = 1.f; }
                if (before > end) {                                          // re-whiten the output history, :1794-1828
                    const int start = OG_SILK_HIST + i * sf - before, cnt = before - end;
                    const int start_res = OG_SILK_RES_HIST + i * sf - before;
                    const int per = (cnt + OG_SK_SUB - 1) / OG_SK_SUB;       // a lane takes `per` consecutive samples: the window slides in registers
                    const int n0 = sub * per, n1 = n0 + per < cnt ? n0 + per : cnt;
                    if (n0 < n1) {
                        float w[ORDER];                                      // w[k] = out[start + n - 1 - k]
#pragma unroll
                        for (int k = 0; k < ORDER; k++) w[k] = out[start + n0 - 1 - k];
                        for (int n = n0; n < n1; n++) {
                            const float cur = out[start + n];
                            float sum = cur;
#pragma unroll
                            for (int k = 0; k < ORDER; k++) sum -= OG_WV(av).a[k] * w[k];
                            res[start_res + n] = og_clamp1(sum) * scale / gain;
#pragma unroll
                            for (int k = ORDER - 1; k > 0; k--) w[k] = w[k - 1];
                            w[0] = cur;
                        }
                    }
                }
                if (end != 0) {                                              // :1830-1841 (disjoint from the samples above)
                    const float rescale = pm.gain[i > 0 ? i - 1 : 0] / gain;
                    const int lo = OG_SILK_RES_HIST + i * sf - end, hi = OG_SILK_RES_HIST + i * sf;
                    for (int n = lo + sub; n < hi; n += OG_SK_SUB) res[n] = res[n] * rescale;
                }
            }
        }
        OG_WAVE_END
        // ---- long-term predictor, :1843-1860: lag - 2 samples at a time
        for (int b = 0; b < 6; b++) {                                        // (sf <= 80, lag - 2 >= 14: at most six steps)
            OG_WAVE_BEGIN(lane)
            const OgSkItem it = OG_WV(itv);
            const int g = lane >> 3, sub = lane & 7;
            if (it.on && i < it.subframes && it.voiced) {
                const OgSkParam &pm = OG_WV(pv);
                float *res = L.res[g];
                const int sf = it.sf, lag = pm.lag[i];
                const int B = lag - 2 > 0 ? lag - 2 : 1;
                const int nlo = OG_SILK_RES_HIST + i * sf + b * B, nhi0 = nlo + B, nend = OG_SILK_RES_HIST + (i + 1) * sf;
                const int nhi = nhi0 < nend ? nhi0 : nend;
                const float t0 = pm.taps[i][0], t1 = pm.taps[i][1], t2 = pm.taps[i][2], t3 = pm.taps[i][3], t4 = pm.taps[i][4];
                for (int n = nlo + sub; n < nhi; n += OG_SK_SUB) {
                    const float *r = res + (n - lag + 2);
                    float sum = res[n];
                    sum += t0 * r[0];
                    sum += t1 * r[-1];
                    sum += t2 * r[-2];
                    sum += t3 * r[-3];
                    sum += t4 * r[-4];
                    res[n] = sum;
                }
            }
            OG_WAVE_END
        }
        // ---- the remaining steps of channels whose lag is shorter than the bitstream can code (never with RFC 6716 tables)
        OG_WAVE_BEGIN(lane)
        {
            const OgSkItem it = OG_WV(itv);
            const int g = lane >> 3, sub = lane & 7;
            if (it.on && i < it.subframes && it.voiced && sub == 0) {
                const OgSkParam &pm = OG_WV(pv);
                float *res = L.res[g];
                const int sf = it.sf, lag = pm.lag[i];
                const int B = lag - 2 > 0 ? lag - 2 : 1;
                const float t0 = pm.taps[i][0], t1 = pm.taps[i][1], t2 = pm.taps[i][2], t3 = pm.taps[i][3], t4 = pm.taps[i][4];
                for (int n = OG_SILK_RES_HIST + i * sf + 6 * B; n < OG_SILK_RES_HIST + (i + 1) * sf; n++) {
                    const float *r = res + (n - lag + 2);
                    float sum = res[n];
                    sum += t0 * r[0];
                    sum += t1 * r[-1];
                    sum += t2 * r[-2];
                    sum += t3 * r[-3];
                    sum += t4 * r[-4];
                    res[n] = sum;
                }
            }
        }
        OG_WAVE_END
        // ---- LPC synthesis, :1864-1884: the recursion, one lane per channel; blocks of 20 samples (20 divides every subframe
        //      size) unrolled so that the 16-deep history h is renamed, not moved
        OG_WAVE_BEGIN(lane)
        {
            const OgSkItem it = OG_WV(itv);
            const int g = lane >> 3, sub = lane & 7;
            if (it.on && i < it.subframes && sub == 0) {
                const float gain = OG_WV(pv).gain[i];
                const int sf = it.sf;
                const float *x = L.res[g] + OG_SILK_RES_HIST + i * sf;
                float *y = L.out[g] + OG_SILK_HIST + i * sf;
                float h[16];
#pragma unroll
                for (int k = 0; k < 16; k++) h[k] = OG_WV(hv).a[k];          // h[0] = most recent unclamped output
                for (int j0 = 0; j0 < sf; j0 += 20) {
                    float xb[20];
#pragma unroll
                    for (int q = 0; q < 20; q++) xb[q] = x[j0 + q];
#pragma unroll
                    for (int q = 0; q < 20; q++) {
                        float sum = xb[q] * gain;
#pragma unroll
                        for (int k = 0; k < ORDER; k++) sum += OG_WV(av).a[k] * h[k];
#pragma unroll
                        for (int k = 15; k > 0; k--) h[k] = h[k - 1];
                        h[0] = sum;
                        xb[q] = og_clamp1(sum);
                    }
#pragma unroll
                    for (int q = 0; q < 20; q++) y[j0 + q] = xb[q];
                }
#pragma unroll
                for (int k = 0; k < 16; k++) OG_WV(hv).a[k] = h[k];
            }
        }
        OG_WAVE_END
    }
}

// The wavefront's channels are (frame, channel) pairs first .. first + 7 of a launch over 2 * nframes pairs: the mid channels of
// all frames first, then the side channels, so that the wavefronts of the second half leave at once where the streams are mono.
OG_DEV void og_silk_synth_wave(const OgSilkFrame *frames, uint32_t nframes, const OgSilkArrays &A, uint32_t first, OgSilkLds &L)
{
    OG_WAVE_VAR(OgSkItem, itv);
    OG_WAVE_VAR(OgSkCoef, hv);                           // lane 0 of a group: the 16 last unclamped outputs, newest first
    OG_WAVE_VAR(int, o16);
    OG_WAVE_VAR(OgSkParam, pv);
    // ---- the channels' records; state in (a reset channel starts from zeros, SilkFrame::flush, silk/decoder.rs:1642-1662);
    //      excitation -> residual (:1637)
    OG_WAVE_BEGIN(lane)
    const int g = lane >> 3, sub = lane & 7;
    const uint32_t idx = first + (uint32_t)g;
    OgSkItem it;
    it.ch = nullptr; it.st = nullptr; it.on = it.reset = it.subframes = it.sf = it.voiced = it.interp = it.factor4 = 0;
    OG_WV(o16) = 0;
    if (idx < 2 * nframes) {
        const int c = idx >= nframes;
        const OgSilkFrame *f = frames + (c ? idx - nframes : idx);
        const OgSilkChan *ch = &f->ch[c];
        it.ch = ch;
        it.st = og_silk_slot(A, (uint32_t)c * A.nstreams + f->stream);
        it.reset = (ch->flags & OG_S_RESET) != 0;
        it.on = (ch->flags & OG_S_CODED) != 0;           // (no frame of this channel -- mid-only, :2093-2099: its history stands)
        it.subframes = ch->subframes; it.sf = ch->sf_size;
        it.voiced = (ch->flags & OG_S_VOICED) != 0; it.interp = (ch->flags & OG_S_INTERP) != 0; it.factor4 = (ch->flags & OG_S_INTERP4) != 0;
        OG_WV(o16) = it.on && ch->order > 10;
        float *out = L.out[g], *res = L.res[g];
        if (it.on) {
            OgSkParam &pm = OG_WV(pv);
            const int order = ch->order;
            pm.ltpscale = ch->ltpscale;
#pragma unroll
            for (int k = 0; k < 4; k++) { pm.gain[k] = ch->gain[k]; pm.lag[k] = ch->pitch_lag[k]; }
#pragma unroll
            for (int k = 0; k < 20; k++) pm.taps[k / 5][k % 5] = ch->ltp_taps[k / 5][k % 5];
#pragma unroll
            for (int k = 0; k < 16; k++) { pm.lpc[k] = k < order ? ch->lpc[k] : 0.f; pm.lpci[k] = k < order ? ch->lpc_interp[k] : 0.f; }
            // every fetch first, in 16-byte pieces dealt over the channel's eight lanes (the slot and the record are 16-byte
            // aligned), then the LDS stores: one memory round trip for the lot -- with one wavefront per SIMD nothing else hides it
            const int f_size = it.subframes * it.sf;
            OgF4v hq[11];                                // hist[324] = 81 pieces
            OgI4 eq[10];                                 // excitation[320] = 80 pieces
            OgF4v lq[4];
#pragma unroll
            for (int m = 0; m < 11; m++) {
                const int q = sub + OG_SK_SUB * m;
                hq[m] = (q < 81 && !it.reset) ? *(const OgF4v *)&it.st[4 * q] : OgF4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int m = 0; m < 10; m++) {
                const int n = 4 * (sub + OG_SK_SUB * m);
                eq[m] = n < f_size ? *(const OgI4 *)&ch->excitation[n] : OgI4{0, 0, 0, 0};
            }
#pragma unroll
            for (int m = 0; m < 4; m++) lq[m] = (sub == 0 && !it.reset) ? *(const OgF4v *)&it.st[OG_SILK_HIST + 2 + 4 * m] : OgF4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 11; m++) {
                const int n = 4 * (sub + OG_SK_SUB * m);
                if (n < OG_SILK_HIST) out[n] = hq[m].x;
                if (n + 1 < OG_SILK_HIST) out[n + 1] = hq[m].y;
                if (n + 2 < OG_SILK_HIST) out[n + 2] = hq[m].z;
                if (n + 3 < OG_SILK_HIST) out[n + 3] = hq[m].w;
            }
#pragma unroll
            for (int m = 0; m < 10; m++) {
                const int n = 4 * (sub + OG_SK_SUB * m);
                if (n < f_size) {                        // (f_size is a multiple of four)
                    res[OG_SILK_RES_HIST + n] = (float)eq[m].x / 8388608.0f;
                    res[OG_SILK_RES_HIST + n + 1] = (float)eq[m].y / 8388608.0f;
                    res[OG_SILK_RES_HIST + n + 2] = (float)eq[m].z / 8388608.0f;
                    res[OG_SILK_RES_HIST + n + 3] = (float)eq[m].w / 8388608.0f;
                }
            }
            if (sub == 0) {                              // lpc[16], oldest first -> h[0] = newest
                const float lv[16] = {lq[0].x, lq[0].y, lq[0].z, lq[0].w, lq[1].x, lq[1].y, lq[1].z, lq[1].w,
                                      lq[2].x, lq[2].y, lq[2].z, lq[2].w, lq[3].x, lq[3].y, lq[3].z, lq[3].w};
#pragma unroll
                for (int k = 0; k < 16; k++) OG_WV(hv).a[k] = lv[15 - k];
            }
        } else if (it.reset) {
            for (int n = sub; n < OG_SILK_SLOT; n += OG_SK_SUB) it.st[n] = 0.f;
        }
    }
    OG_WV(itv) = it;
    OG_WAVE_END
    int any16 = 0;
#if defined(__HIP__)
    any16 = __builtin_amdgcn_readfirstlane(__any(o16 != 0));
#else
    for (int l = 0; l < 64; l++) any16 |= o16[l];
#endif
    if (any16) og_silk_subframes<16>(L, itv, hv, pv);
    else og_silk_subframes<10>(L, itv, hv, pv);
    // ---- state out: the newest 322 samples (the history shift of :1891-1898) and the 16 unclamped ones
    OG_WAVE_BEGIN(lane)
    const OgSkItem it = OG_WV(itv);
    const int g = lane >> 3, sub = lane & 7;
    if (it.on) {
        const int f_size = it.subframes * it.sf;
        const float *out = L.out[g] + f_size;
#pragma unroll
        for (int m = 0; m < 11; m++) {                   // 16-byte pieces; hist[322], [323] are padding
            const int q = sub + OG_SK_SUB * m, n = 4 * q;
            if (q < 81) *(OgF4v *)&it.st[n] = OgF4v{out[n], out[n + 1], n + 2 < OG_SILK_HIST ? out[n + 2] : 0.f, n + 3 < OG_SILK_HIST ? out[n + 3] : 0.f};
        }
        if (sub == 0) {
#pragma unroll
            for (int m = 0; m < 4; m++)
                *(OgF4v *)&it.st[OG_SILK_HIST + 2 + 4 * m] = OgF4v{OG_WV(hv).a[15 - 4 * m], OG_WV(hv).a[14 - 4 * m], OG_WV(hv).a[13 - 4 * m], OG_WV(hv).a[12 - 4 * m]};
        }
    }
    OG_WAVE_END
}

// ------------------------------------------------------------------------------------------------
// K5 + K6, one workgroup per frame.  LDS: x[2][32 + 320] input samples at the internal rate.
struct alignas(16) OgSilkOutLds {
    float x[2][32 + 320];
    float y[2 * 960];                 // the frame's 48 kHz output, interleaved, on its way to a coalesced store
};

// Polyphase FIR: y[m] = sum_k h[m - k Lf] x[k], k ascending (k < 0 from the history), Lf = 48 kHz / internal rate.  A work item is
// one input position k' of one channel and produces the Lf outputs m = Lf k' + p: they share the 73 / Lf input samples, which
// are read once, and the taps -- compile-time indices into a uniform table -- come in scalar registers, so LDS sees 8 reads per
// output instead of the 50 of one work item per output (round 2: 75 us per 16 384 frames, bound by those reads).  Per output the
// products are added in the same order as before (k ascending): the same bits.
template <int LF>
OG_DEV void og_silk_fir(const float *__restrict__ hh /* uniform: the bandwidth's prototype in global memory */, OgSilkOutLds &L, int fs, int CC, int tid)
{
    constexpr int J = (2 * OG_RS_HALF) / LF;            // taps per phase: j = 0 .. J (p + LF j <= 72)
    for (int q = tid; q < fs * CC; q += OG_WG) {
        const int c = q >= fs, kp = c ? q - fs : q;
        const float *x = &L.x[c][32 + kp];
        float xv[J + 1];
#pragma unroll
        for (int j = 0; j <= J; j++) xv[j] = x[-j];     // x[k' - j]; k' - j >= -24 > -32: inside the history
#pragma unroll
        for (int p = 0; p < LF; p++) {
            float acc = 0.f;
#pragma unroll
            for (int j = J; j >= 0; j--)                // k = k' - j ascending
                if (p + LF * j <= 2 * OG_RS_HALF) acc += hh[p + LF * j] * xv[j];
            L.y[(LF * kp + p) * CC + c] = acc;
        }
    }
}

OG_DEV void og_silk_output(const OgSilkFrame *f, const OgSilkArrays &A, const OgSilkTables *T, float *pcm_base,
                           size_t pcm_stride, OgSilkOutLds &L, float *dbg_internal /* [2][320] or null */)
{
    const int fs = f->f_size, bw = f->bandwidth;
    const bool stereo = (f->flags & OG_SF_STEREO) != 0, stereo_out = (f->flags & OG_SF_STEREO_OUT) != 0;
    const int CC = stereo_out ? 2 : 1;
    // the frame as K4 left it, at the end of the slot's history: mid[i] = two samples behind the frame's sample i, side[i] = one
    // behind (silk/decoder.rs:2040: output[in_start - 2 ..] / [in_start - 1 ..])
    const float *mid = og_silk_slot(A, f->stream) + (OG_SILK_HIST - fs - 2);
    const float *side = og_silk_slot(A, A.nstreams + f->stream) + (OG_SILK_HIST - fs - 1);
    float *hist = A.rs_hist + (size_t)f->stream * 64;
    const bool rs_reset = (f->flags & OG_SF_RESAMPLER_RESET) != 0;

    OG_LANES_BEGIN(tid)
    for (int q = tid; q < CC * 32; q += OG_WG) {
        const int c = q >> 5, i = q & 31;
        L.x[c][i] = rs_reset ? 0.f : hist[c * 32 + i];
    }
    if (f->flags & OG_SF_CONCEAL) {                                          // nothing to decode for this frame: silence in
        for (int i = tid; i < fs; i += OG_WG) { L.x[0][32 + i] = 0.f; L.x[1][32 + i] = 0.f; }
    } else if (stereo && stereo_out) {                                       // unmix_ms, silk/decoder.rs:2012-2061
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
    // 48 kHz samples into LDS, then out in order; the last 32 inputs become the next frame's history (fs >= 80 > 32)
    float *pcm = pcm_base + (size_t)f->stream * pcm_stride + (size_t)f->pcm_off * CC;
    OG_LANES_BEGIN(tid)
    if (bw == 0) og_silk_fir<6>(T->rs_taps[0], L, fs, CC, tid);
    else if (bw == 1) og_silk_fir<4>(T->rs_taps[1], L, fs, CC, tid);
    else og_silk_fir<3>(T->rs_taps[2], L, fs, CC, tid);
    for (int q = tid; q < CC * 32; q += OG_WG) {
        const int c = q >> 5, i = q & 31;
        hist[c * 32 + i] = L.x[c][fs + i];
    }
    OG_LANES_END
    const int nout = fs * (bw == 0 ? 6 : bw == 1 ? 4 : 3) * CC;
    OG_LANES_BEGIN(tid)
    if ((((uintptr_t)pcm) & 15) == 0) {                 // (nout is a multiple of four)
        for (int q = tid; q < nout / 4; q += OG_WG) ((OgF4v *)pcm)[q] = ((const OgF4v *)L.y)[q];
    } else {
        for (int q = tid; q < nout; q += OG_WG) pcm[q] = L.y[q];
    }
    OG_LANES_END
}
