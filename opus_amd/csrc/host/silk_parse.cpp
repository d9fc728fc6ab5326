// SILK frame parser: the entropy-decoding half of the reference's Silk::decode / SilkFrame::parse
// (lu-zero/opus src/silk/decoder.rs:1664-1770, :2063-2127).  Every range-decoder symbol is read here
// and turned into the numeric parameters of one OgSilkFrame (gains, LPC, pitch lags, LTP taps,
// excitation as the pre-division integer); the recursive filters run on the GPU (hip/silk_synth.h).
//
// Integer (Q-format) arithmetic follows the reference line by line because the resulting f32
// parameters must be bit-identical to the reference's for its golden PCM (silk/tests.rs) to hold.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "parse.hpp"
#include "../../../include/opusgpu.h"

namespace og {

void SilkChanHost::flush() { *this = SilkChanHost{}; }                       // silk/decoder.rs:1642-1662
void SilkHostState::flush()                                                  // silk/decoder.rs:1934-1940
{
    if (mid.coded) mid.flush();
    if (side.coded) side.flush();
    stereo_reset = true;
    prev_w0 = prev_w1 = 0.f;
}

// The whole state, whether a channel has been coded or not: the side channel's `flushed` can have been spent on a reset record
// that never reached the device (its packet was refused further on), and a later flush() would then skip the channel.
void SilkHostState::reset()
{
    mid.flush();
    side.flush();
    stereo_reset = true;
    w0 = w1 = prev_w0 = prev_w1 = 0.f;
}

namespace {

inline int32_t mul_shift(int32_t a, int32_t b, int bits) { return int32_t((int64_t(a) * b) >> bits); }                // :104-110
inline int32_t mul_round(int32_t a, int64_t b, int bits) { return int32_t((int64_t(a) * b + (int64_t(1) << (bits - 1))) >> bits); }   // :112-118

inline int log2lin(int x)                                                    // silk/decoder.rs:95-101
{
    const int i = 1 << (x >> 7), f = x & 127;
    return i + ((-174 * f * (128 - f) >> 16) + f) * (i >> 7);
}

struct BandTables {                                                          // silk/decoder.rs:121-135 `trait Band`
    int order, step;
    const uint16_t *stage1, *stage2;
    const uint8_t *pred_weight, *pred_index, *codebook, *ordering;
    const uint16_t *weight;
    const int16_t *min_spacing;
};
const BandTables kNbMb = {10, 11796, ICDF_SILK_LSF_STAGE1_NB_MB, &ICDF_SILK_LSF_STAGE2_NB_MB[0][0],
                          &SILK_LSF_PRED_WEIGHT_NB_MB[0][0], &SILK_LSF_PRED_WEIGHT_INDEX_NB_MB[0][0], &SILK_LSF_CODEBOOK_NB_MB[0][0],
                          SILK_LSF_ORDERING_NB_MB, &SILK_LSF_WEIGHT_NB_MB[0][0], SILK_LSF_MIN_SPACING_NB_MB};
const BandTables kWb = {16, 9830, ICDF_SILK_LSF_STAGE1_WB, &ICDF_SILK_LSF_STAGE2_WB[0][0],
                        &SILK_LSF_PRED_WEIGHT_WB[0][0], &SILK_LSF_PRED_WEIGHT_INDEX_WB[0][0], &SILK_LSF_CODEBOOK_WB[0][0],
                        SILK_LSF_ORDERING_WB, &SILK_LSF_WEIGHT_WB[0][0], SILK_LSF_MIN_SPACING_WB};

// NLSF stabilisation, silk/decoder.rs:137-204 (RFC 6716 4.2.7.5.4)
void stabilize(const BandTables &t, int16_t *nlsf)
{
    const int order = t.order;
    const int16_t *ms = t.min_spacing;
    for (int iter = 0; iter < 20; iter++) {
        int k = 0, worst = 0;
        for (int i = 0; i <= order; i++) {
            const int lo = i ? nlsf[i - 1] : 0, hi = i == order ? 32768 : nlsf[i];
            const int d = hi - lo - ms[i];
            if (d < worst) { worst = d; k = i; }
        }
        if (worst == 0) return;
        if (k == 0) {
            nlsf[0] = ms[0];
        } else if (k == order) {
            nlsf[order - 1] = int16_t(32768 - ms[order]);
        } else {
            const int half = ms[k] >> 1;
            int below = 0, above = 0;
            for (int i = 0; i < k; i++) below += ms[i];
            for (int i = k + 1; i <= order; i++) above += ms[i];
            const int cmin = below + half, cmax = 32768 - above - half;
            const int sum = nlsf[k - 1] + nlsf[k];
            const int center = std::min(cmax, std::max(cmin, (sum >> 1) + (sum & 1)));
            nlsf[k - 1] = int16_t(center - half);
            nlsf[k] = int16_t(std::min(32767, nlsf[k - 1] + ms[k]));
        }
    }
    // fallback: sort and clamp both ways.  The way up saturates at 32767 (libopus' silk_NLSF_stabilize; the reference adds in
    // i16, decoder.rs:189, which overflows -- a panic in a checked build, and a negative table index at :335-339 otherwise --
    // when a packet crowds the coefficients at the top; the way down then brings them into range).  Found by
    // tools/parse_fuzz_asan.cpp.
    std::sort(nlsf, nlsf + order);
    int prev = 0;
    for (int i = 0; i < order; i++) {
        nlsf[i] = int16_t(std::min(32767, std::max<int>(nlsf[i], prev + ms[i])));
        prev = nlsf[i];
    }
    int next = 32768;
    for (int i = order - 1; i >= 0; i--) {
        nlsf[i] = int16_t(std::min<int>(nlsf[i], next - ms[i + 1]));
        next = nlsf[i];
    }
}

// LPC stability test by reflection coefficients, silk/decoder.rs:206-260 (RFC 6716 4.2.7.5.8)
bool is_stable(const int16_t *lpc, int order)
{
    int32_t buf[2][16], dc = 0;
    for (int i = 0; i < order; i++) { dc += lpc[i]; buf[0][i] = lpc[i] * 4096; buf[1][i] = 0; }
    if (dc > 4096) return false;
    int32_t invgain = 1 << 30;
    int cur = 0;                                                             // row holding the order-(k+1) polynomial
    for (int k = order - 1;; k--) {
        const int32_t a = buf[cur][k];
        if (std::abs(a) > 16773022) return false;
        const int32_t rc = -a * 128;
        const int32_t div = (1 << 30) - mul_shift(rc, rc, 32);
        invgain = mul_shift(invgain, div, 32) << 2;
        if (k == 0) return invgain >= 107374;
        const int b1 = ilog(uint32_t(div)), b2 = b1 - 16;
        const int32_t inv = ((1 << 29) - 1) / (div >> (b2 + 1));
        const int32_t err = (1 << 29) - mul_shift(int32_t(uint32_t(div) << (15 - b2)), inv, 16);
        const int32_t gain = int32_t(uint32_t(inv) << 16) + (err * inv >> 13);
        const int32_t *src = buf[cur];
        int32_t *dst = buf[cur ^ 1];
        // (the difference wraps for coefficients far outside a stable filter's, as in the reference's release build: unsigned here)
        for (int j = 0; j < k; j++) dst[j] = mul_shift(int32_t(uint32_t(src[j]) - uint32_t(mul_shift(src[k - j - 1], rc, 31))), gain, b1);
        cur ^= 1;
    }
}

// bandwidth expansion until the Q12 coefficients fit and the filter is stable, silk/decoder.rs:262-324
void range_limit(float *out, int32_t *a, int order)
{
    int16_t q12[16];
    bool fits = false;
    for (int iter = 0; iter < 10 && !fits; iter++) {
        int k = 0;
        int32_t maxabs = -1;
        for (int i = 0; i < order; i++) { const int32_t v = std::abs(a[i]); if (v > maxabs) { maxabs = v; k = i; } }
        const uint32_t m = uint32_t((maxabs + 16) >> 5);
        if (m <= 32767) { fits = true; break; }
        const uint32_t mc = std::min<uint32_t>(m, 163838);
        const uint32_t start = 65470 - ((mc - 32767) << 14) / ((mc * uint32_t(k + 1)) >> 2);
        uint32_t chirp = start;
        for (int i = 0; i < order; i++) {
            a[i] = mul_round(a[i], chirp, 16);
            chirp = (start * chirp + 32768) >> 16;
        }
    }
    for (int i = 0; i < order; i++) {
        int32_t v = (a[i] + 16) >> 5;
        if (!fits) { v = std::min(32767, std::max(-32768, v)); a[i] = v << 5; }   // saturate after 10 rounds
        q12[i] = int16_t(v);
    }
    for (int i = 1; i <= 16 && !is_stable(q12, order); i++) {
        const uint32_t start = 65536u - (1u << i);
        uint32_t chirp = start;
        for (int j = 0; j < order; j++) {
            a[j] = mul_round(a[j], chirp, 16);
            q12[j] = int16_t((a[j] + 16) >> 5);
            chirp = (start * chirp + 32768) >> 16;
        }
    }
    for (int i = 0; i < order; i++) out[i] = float(q12[i]) / 4096.f;
}

// NLSF -> LPC, silk/decoder.rs:326-397 (RFC 6716 4.2.7.5.6)
void lsf_to_lpc(const BandTables &t, float *lpc, const int16_t *nlsf)
{
    const int order = t.order, half = order / 2;
    int32_t c[16], p[9] = {0}, q[9] = {0}, a[16];
    for (int i = 0; i < order; i++) {
        const int idx = std::min(127, std::max(0, nlsf[i] >> 8)), frac = nlsf[i] & 255;       // (in range by construction: stabilize)
        const int c0 = SILK_COSINE[idx], c1 = SILK_COSINE[idx + 1];
        c[t.ordering[i]] = (c0 * 256 + (c1 - c0) * frac + 4) >> 3;
    }
    p[0] = q[0] = 65536;
    p[1] = -c[0];
    q[1] = -c[1];
    for (int i = 0; i < half - 1; i++) {
        const int32_t cp = c[2 + 2 * i], cq = c[3 + 2 * i];
        p[i + 2] = p[i] * 2 - mul_round(cp, p[i + 1], 16);
        q[i + 2] = q[i] * 2 - mul_round(cq, q[i + 1], 16);
        for (int j = i + 1; j >= 2; j--) {
            p[j] += p[j - 2] - mul_round(cp, p[j - 1], 16);
            q[j] += q[j - 2] - mul_round(cq, q[j - 1], 16);
        }
        p[1] -= cp;
        q[1] -= cq;
    }
    for (int i = 0; i < half; i++) {
        const int32_t ps = p[i] + p[i + 1], qd = q[i + 1] - q[i];
        a[i] = -ps - qd;
        a[order - 1 - i] = -ps + qd;
    }
    range_limit(lpc, a, order);
}

struct ChanParser {
    SilkChanHost &st;
    RangeDec &rd;
    OgSilkChan &out;
    int active = 0, voiced = 0, high = 0;

    float subframe_gain(bool independent)                                    // silk/decoder.rs:1326-1344
    {
        if (independent) {
            const int msb = rd.icdf(ICDF_SILK_MSB_SUBFRAME_GAIN[voiced + active]);
            const int lsb = rd.icdf(ICDF_SILK_LSB_SUBFRAME_GAIN);
            st.log_gain = std::max((msb << 3) | lsb, st.log_gain - 16);
        } else {
            const int delta = rd.icdf(ICDF_SILK_DELTA_SUBFRAME_GAIN);
            st.log_gain = std::min(63, std::max(0, std::max(delta * 2 - 16, st.log_gain + delta - 4)));
        }
        return float(log2lin((st.log_gain * 0x1D1C71 >> 16) + 2090)) / 65536.0f;
    }

    void lpc(const BandTables &t, bool long_frame)                           // silk/decoder.rs:1347-1463
    {
        const int order = t.order;
        const int s1 = rd.icdf(t.stage1[voiced]);
        int s2[16];
        for (int i = 0; i < order; i++) {
            int v = rd.icdf(t.stage2[s1 * order + i]) - 4;
            if (v == -4) v -= rd.icdf(ICDF_SILK_LSF_STAGE2_EXTENSION);
            else if (v == 4) v += rd.icdf(ICDF_SILK_LSF_STAGE2_EXTENSION);
            s2[i] = v;
        }
        int16_t res[16], nlsf[16];
        int16_t prev = 0;
        for (int i = order - 1; i >= 0; i--) {                               // backwards prediction of the residual
            const int fix = s2[i] < 0 ? 102 : s2[i] > 0 ? -102 : 0;
            int16_t r = int16_t(((s2[i] * 1024 + fix) * t.step) >> 16);
            if (i != order - 1) {
                const int w = t.pred_weight[t.pred_index[s1 * (order - 1) + i] * (order - 1) + i];
                r = int16_t(r + int16_t((int(prev) * w) >> 8));
            }
            prev = res[i] = r;
        }
        for (int i = 0; i < order; i++) {
            const int v = (int(t.codebook[s1 * order + i]) << 7) + (int(res[i]) << 14) / int(t.weight[s1 * order + i]);
            nlsf[i] = int16_t(std::min(32767, std::max(0, v)));
        }
        stabilize(t, nlsf);
        bool interpolated = false, factor4 = true;
        if (long_frame) {
            const int w = rd.icdf(ICDF_SILK_LSF_INTERPOLATION_INDEX);
            if (w != 4 && st.coded) {
                interpolated = true;
                factor4 = false;
                if (w != 0) {
                    int16_t mix[16];
                    for (int i = 0; i < order; i++) mix[i] = int16_t(st.nlsfs[i] + ((int16_t(nlsf[i] - st.nlsfs[i]) * w) >> 2));
                    lsf_to_lpc(t, out.lpc_interp, mix);
                } else {
                    std::memcpy(out.lpc_interp, st.lpc, order * sizeof(float));   // previous frame's filter
                }
            }
        }
        std::memcpy(st.nlsfs, nlsf, order * sizeof(int16_t));
        lsf_to_lpc(t, st.lpc, nlsf);
        std::memcpy(out.lpc, st.lpc, order * sizeof(float));
        if (interpolated) out.flags |= OG_S_INTERP;
        if (factor4) out.flags |= OG_S_INTERP4;
    }

    void pitch(int nsf, int bw, bool absolute)                               // silk/decoder.rs:1465-1507
    {
        static const int kMin[3] = {16, 24, 32}, kMax[3] = {144, 216, 288}, kScale[3] = {4, 6, 8};
        int lag = 0;
        bool have = false;
        if (!absolute) {
            const int delta = rd.icdf(ICDF_SILK_PITCH_DELTA);
            if (delta) { lag = st.previous_lag + delta - 9; have = true; }
        }
        if (!have) {
            const int hi = rd.icdf(ICDF_SILK_PITCH_HIGH_PART);
            const int lo = rd.icdf(ICDF_SILK_PITCH_LOW_PART[bw]);
            lag = hi * kScale[bw] + lo + kMin[bw];
        }
        st.previous_lag = lag;
        const int8_t *offs;
        if (nsf == 2) {
            const int idx = rd.icdf(bw == 0 ? ICDF_SILK_PITCH_CONTOUR_NB[0] : ICDF_SILK_PITCH_CONTOUR_MB_WB[0]);
            offs = bw == 0 ? SILK_PITCH_OFFSET_NB_10MS[idx] : SILK_PITCH_OFFSET_MB_WB_10MS[idx];
        } else {
            const int idx = rd.icdf(bw == 0 ? ICDF_SILK_PITCH_CONTOUR_NB[1] : ICDF_SILK_PITCH_CONTOUR_MB_WB[1]);
            offs = bw == 0 ? SILK_PITCH_OFFSET_NB_20MS[idx] : SILK_PITCH_OFFSET_MB_WB_20MS[idx];
        }
        for (int i = 0; i < nsf; i++) out.pitch_lag[i] = int16_t(std::min(kMax[bw], std::max(kMin[bw], lag + offs[i])));
    }

    void pulse_split(int level, int avail, int *halves)                      // silk/decoder.rs:1564-1578
    {
        halves[0] = avail ? rd.icdf(ICDF_SILK_PULSE_LOCATION[level][avail - 1]) : 0;
        halves[1] = avail - halves[0];
    }

    void excitation(int bw, bool long_frame)                                 // silk/decoder.rs:1522-1640
    {
        static const int kShell[3][2] = {{5, 10}, {8, 15}, {10, 20}};
        const int blocks = kShell[bw][long_frame];
        int count[20], lsbs[20], e[320];
        uint32_t seed = uint32_t(rd.icdf(ICDF_SILK_LCG_SEED));
        const int rate = rd.icdf(ICDF_SILK_EXC_RATE[voiced]);
        for (int b = 0; b < blocks; b++) {
            int p = rd.icdf(ICDF_SILK_PULSE_COUNT[rate]), l = 0;
            if (p == 17) {
                while (p == 17 && ++l != 10) p = rd.icdf(ICDF_SILK_PULSE_COUNT[9]);
                if (l == 10) p = rd.icdf(ICDF_SILK_PULSE_COUNT[10]);
            }
            count[b] = p;
            lsbs[b] = l;
        }
        for (int b = 0; b < blocks; b++) {                                   // 16 -> 8 -> 4 -> 2 -> 1 binary splits
            int *loc = e + 16 * b;
            if (!count[b]) { std::memset(loc, 0, 16 * sizeof(int)); continue; }
            int h8[2], h4[2], h2[2], h1[2];
            pulse_split(0, count[b], h8);
            for (int i8 = 0; i8 < 2; i8++) {
                pulse_split(1, h8[i8], h4);
                for (int i4 = 0; i4 < 2; i4++) {
                    pulse_split(2, h4[i4], h2);
                    for (int i2 = 0; i2 < 2; i2++) {
                        pulse_split(3, h2[i2], h1);
                        loc[8 * i8 + 4 * i4 + 2 * i2] = h1[0];
                        loc[8 * i8 + 4 * i4 + 2 * i2 + 1] = h1[1];
                    }
                }
            }
        }
        for (int b = 0; b < blocks; b++)
            for (int j = 0; j < 16; j++)
                for (int k = 0; k < lsbs[b]; k++) e[16 * b + j] = (e[16 * b + j] << 1) | rd.icdf(ICDF_SILK_EXC_LSB);
        for (int b = 0; b < blocks; b++)
            for (int j = 0; j < 16; j++)
                if (e[16 * b + j] && rd.icdf(ICDF_SILK_EXC_SIGN[voiced + active][high][std::min(count[b], 6)]) == 0)
                    e[16 * b + j] = -e[16 * b + j];
        const int offset = SILK_QUANT_OFFSET[voiced][high];
        for (int i = 0; i < blocks * 16; i++) {                              // silk/decoder.rs:1620-1638
            const int l = e[i];
            int ex = (l * 256 | offset) - 20 * ((l > 0) - (l < 0));
            seed = seed * 196314165u + 907633515u;
            if (seed & 0x80000000u) ex = -ex;
            seed += uint32_t(l);
            out.excitation[i] = ex;                                          // f32 residual = ex / 2^23 on the device
        }
        for (int i = blocks * 16; i < 320; i++) out.excitation[i] = 0;
    }

    // SilkFrame::parse up to the synthesis loop (silk/decoder.rs:1664-1770)
    void frame(int bw, int subframes, int sf_size, bool vad, bool first)
    {
        std::memset(&out, 0, sizeof out);
        out.flags = OG_S_CODED;
        if (st.flushed) out.flags |= OG_S_RESET;             // history was cleared on the host (:1642-1662): clear it in HBM too
        if (vad) {
            const int t = rd.icdf(ICDF_SILK_FRAME_TYPE_ACTIVE);
            active = 1; voiced = t >> 1; high = t & 1;
        } else {
            active = 0; voiced = 0; high = rd.icdf(ICDF_SILK_FRAME_TYPE_INACTIVE) != 0;
        }
        const BandTables &t = bw == 2 ? kWb : kNbMb;
        out.order = uint8_t(t.order);
        out.subframes = uint8_t(subframes);
        out.sf_size = uint8_t(sf_size);
        for (int i = 0; i < subframes; i++) out.gain[i] = subframe_gain(i == 0 && (first || !st.coded));
        const bool long_frame = subframes == 4;
        lpc(t, long_frame);
        if (voiced) {
            out.flags |= OG_S_VOICED;
            pitch(subframes, bw, first || !st.prev_voiced);
            const int per = rd.icdf(ICDF_SILK_LTP_PERIODICITY);              // silk/decoder.rs:1509-1520
            for (int i = 0; i < subframes; i++) {
                const int8_t *taps = SILK_LTP_TAPS + SILK_LTP_TAPS_ROW[per] + 5 * rd.icdf(ICDF_SILK_LTP_FILTER[per]);
                for (int k = 0; k < 5; k++) out.ltp_taps[i][k] = float(taps[k]) / 128.f;
            }
        }
        out.ltpscale = float(voiced && first ? SILK_LTP_SCALE[rd.icdf(ICDF_SILK_LTP_SCALE_INDEX)] : 15565) / 16384.f;   // :1752-1756
        excitation(bw, long_frame);
        st.prev_voiced = voiced;
        st.coded = 1;
        st.flushed = false;
        st.last_gain = out.gain[subframes - 1];
        st.last_lag = out.pitch_lag[subframes - 1];
        std::memcpy(st.last_taps, out.ltp_taps[subframes - 1], sizeof st.last_taps);
        std::memcpy(st.last_lpc, out.lpc, sizeof st.last_lpc);
        st.last_n = sf_size * subframes;
        std::memcpy(st.last_exc, out.excitation, size_t(st.last_n) * sizeof(int32_t));
    }
};

// One concealed channel frame: see silk_conceal (oracle: orc_silk.c frame_conceal, operation for operation).
void conceal_chan(SilkChanHost &st, OgSilkChan &out, int bw, int subframes, int sf_size, int lost)
{
    const int f_size = sf_size * subframes, order = bw == 2 ? 16 : 10;
    std::memset(&out, 0, sizeof out);
    const bool voiced = st.prev_voiced != 0;
    out.flags = uint8_t(OG_S_CODED | (voiced ? OG_S_VOICED : 0) | (st.flushed ? OG_S_RESET : 0));
    out.order = uint8_t(order);
    out.subframes = uint8_t(subframes);
    out.sf_size = uint8_t(sf_size);
    out.ltpscale = 1.f;
    const float att = lost == 0 ? 0.99f : voiced ? 0.95f : 0.9f;
    float chirp = 0.99f;
    for (int k = 0; k < order; k++) { st.last_lpc[k] *= chirp; chirp *= 0.99f; }
    if (voiced) {
        if (lost == 0) {
            float sum = 0;
            for (int k = 0; k < 5; k++) sum += st.last_taps[k];
            const float scale = sum > 0.95f ? 0.95f / sum : sum < 0.7f && sum > 0.01f ? 0.7f / sum : 1.f;
            for (int k = 0; k < 5; k++) st.last_taps[k] *= scale;
        }
        for (int k = 0; k < 5; k++) st.last_taps[k] *= att;
        std::memset(st.last_exc, 0, sizeof st.last_exc);
    } else {
        int32_t pick[320];
        st.last_gain *= att;
        for (int i = 0; i < f_size; i++) {
            st.plc_seed = st.plc_seed * 196314165u + 907633515u;
            pick[i] = st.last_exc[st.last_n - 64 + int(st.plc_seed >> 26)];
        }
        std::memcpy(st.last_exc, pick, size_t(f_size) * sizeof(int32_t));
        std::memcpy(out.excitation, pick, size_t(f_size) * sizeof(int32_t));
    }
    st.last_n = f_size;
    for (int i = 0; i < subframes; i++) {
        out.gain[i] = st.last_gain;
        out.pitch_lag[i] = int16_t(st.last_lag);
        std::memcpy(out.ltp_taps[i], st.last_taps, sizeof st.last_taps);
    }
    std::memcpy(out.lpc, st.last_lpc, sizeof out.lpc);
    std::memcpy(out.lpc_interp, st.last_lpc, sizeof out.lpc_interp);
    st.flushed = false;
}

}  // namespace

int silk_parse(SilkHostState &st, RangeDec &rd, int bw, int duration_ms, int stereo, std::vector<OgSilkFrame> &out, bool fec)
{
    // Silk::setup (silk/decoder.rs:1942-1977)
    const int frames = duration_ms <= 20 ? 1 : duration_ms / 20;
    const int subframes = duration_ms == 10 ? 2 : 4;
    const int sf_size = bw == 0 ? 40 : bw == 1 ? 60 : 80;
    bool mid_vad[3] = {false, false, false}, side_vad[3] = {false, false, false};
    bool lbrr[2][3] = {{false, false, false}, {false, false, false}}, any_lbrr[2] = {false, false};
    for (int i = 0; i < frames; i++) mid_vad[i] = rd.bit_logp(1);            // silk/decoder.rs:2066-2081
    any_lbrr[0] = rd.bit_logp(1);                                            // the reference stops here: Error::Unsupported (:2070-2072)
    if (stereo) {
        for (int i = 0; i < frames; i++) side_vad[i] = rd.bit_logp(1);
        any_lbrr[1] = rd.bit_logp(1);
    }
    // RFC 6716 4.2.4: per-frame LBRR flags (Table 4: the symbol is the bit mask of the frames, frame 0 lowest, never 0), then
    // 4.2.5-4.2.7 the LBRR (in-band FEC) frames, in front of the regular ones and interleaved mid / side per 20 ms.  An LBRR
    // frame is coded like a regular frame with the VAD flag set, independently unless its channel had one 20 ms earlier.  A
    // decoder that is not recovering a loss reads past them: they are parsed into scratch state and records and dropped
    // (with the stereo weights they carry), so nothing of them reaches the device.
    static const uint16_t kLbrr40[] = {256, 4, 0, 53, 106, 256}, kLbrr60[] = {256, 8, 0, 41, 61, 90, 131, 146, 174, 256};
    for (int ch = 0; ch < (stereo ? 2 : 1); ch++) {
        if (!any_lbrr[ch]) continue;
        const int sym = frames == 1 ? 1 : rd.icdf_table(frames == 2 ? kLbrr40 : kLbrr60);
        for (int i = 0; i < frames; i++) lbrr[ch][i] = (sym >> i) & 1;
    }
    auto new_frame = [&]() -> OgSilkFrame & {
        out.emplace_back();
        OgSilkFrame &f = out.back();
        std::memset(&f, 0, sizeof f);
        f.bandwidth = uint8_t(bw);
        f.f_size = uint16_t(sf_size * subframes);
        f.flags = uint8_t((stereo ? OG_SF_STEREO : 0) | (st.stereo_out ? OG_SF_STEREO_OUT : 0));
        if (st.resampler_reset) { f.flags |= OG_SF_RESAMPLER_RESET; st.resampler_reset = false; }
        f.prev_w0 = st.prev_w0;
        f.prev_w1 = st.prev_w1;
        return f;
    };
    auto stereo_weights = [&]() {                        // parse_stereo_weight, :1979-2010
        static const int kW[16] = {-13732, -10050, -8266, -7526, -6500, -5000, -2950, -820,
                                   820, 2950, 5000, 6500, 7526, 8266, 10050, 13732};
        const int n = rd.icdf(ICDF_SILK_STEREO_STAGE1);
        const int i0 = rd.icdf(ICDF_SILK_STEREO_STAGE2) + 3 * (n / 5);
        const int i1 = rd.icdf(ICDF_SILK_STEREO_STAGE3) * 2 + 1;
        const int i2 = rd.icdf(ICDF_SILK_STEREO_STAGE2) + 3 * (n % 5);
        const int i3 = rd.icdf(ICDF_SILK_STEREO_STAGE3) * 2 + 1;
        const int w0 = kW[i0] + (((kW[i0 + 1] - kW[i0]) * 6554) >> 16) * i1;
        const int w1 = kW[i2] + (((kW[i2 + 1] - kW[i2]) * 6554) >> 16) * i3;
        st.w0 = float(w0 - w1) / 8192.f;
        st.w1 = float(w1) / 8192.f;
    };
    auto side_absent = [&](OgSilkFrame &f, bool midonly) {
        // side absent: the reference flushes it when mid-only (silk/decoder.rs:2097-2099); the device clears the
        // side history on OG_S_RESET without OG_S_CODED
        std::memset(&f.ch[1], 0, sizeof f.ch[1]);
        f.ch[1].order = f.ch[0].order; f.ch[1].subframes = uint8_t(subframes); f.ch[1].sf_size = uint8_t(sf_size);
        if (midonly) {
            f.flags |= OG_SF_MID_ONLY;
            if (st.side.coded) st.side.flush();
            f.ch[1].flags = OG_S_RESET;                 // unmix_ms of this very frame reads the cleared side history
            st.side.flushed = false;
        } else if (st.side.flushed) {
            // (only an in-band FEC recovery gets here: a mid copy without a side copy and without the mid-only flag.)  The side
            // was flushed on the host -- a CELT packet in between, say -- but no side frame has reached the device since: its
            // history there is still the old one, and unmix_ms of this frame reads it
            f.ch[1].flags = OG_S_RESET;
            st.side.flushed = false;
        }
    };
    if (fec) {
        // In-band FEC recovery (RFC 6716 4.2.4-4.2.7; the reference stops at the LBRR flag, silk/decoder.rs:2070-2072): the LBRR
        // frames of this packet are low-rate copies of the PREVIOUS packet's frames.  They are decoded into the running state in
        // place of the lost packet; the caller sends the packet again, plainly, for its own audio.
        SilkChanHost scratch;
        OgSilkChan dropped;
        for (int i = 0; i < frames; i++) {
            OgSilkFrame &f = new_frame();
            if (lbrr[0][i]) {
                bool midonly = false;
                if (stereo) {
                    stereo_weights();
                    if (!lbrr[1][i]) midonly = rd.icdf(ICDF_SILK_MID_ONLY) != 0;
                }
                f.w0 = st.w0;
                f.w1 = st.w1;
                ChanParser{st.mid, rd, f.ch[0]}.frame(bw, subframes, sf_size, true, i == 0 || !lbrr[0][i - 1]);
                if (stereo && lbrr[1][i]) ChanParser{st.side, rd, f.ch[1]}.frame(bw, subframes, sf_size, true, i == 0 || !lbrr[1][i - 1]);
                else side_absent(f, midonly);
                if (stereo && st.stereo_out) { st.prev_w0 = st.w0; st.prev_w1 = st.w1; }
            } else {
                // no copy of this frame: silence, and both channels start over (a side-only copy is read past)
                if (stereo && lbrr[1][i]) ChanParser{scratch, rd, dropped}.frame(bw, subframes, sf_size, true, i == 0 || !lbrr[1][i - 1]);
                f.flags |= OG_SF_CONCEAL;
                f.w0 = st.w0;
                f.w1 = st.w1;
                st.mid.flush(); st.side.flush();
                for (int c = 0; c < 2; c++) {
                    f.ch[c].order = uint8_t(bw == 2 ? 16 : 10); f.ch[c].subframes = uint8_t(subframes); f.ch[c].sf_size = uint8_t(sf_size);
                    f.ch[c].flags = OG_S_RESET;
                }
                st.mid.flushed = st.side.flushed = false;
            }
        }
        return frames;
    }
    if (any_lbrr[0] || any_lbrr[1]) {
        SilkChanHost scratch[2];
        OgSilkChan dropped;
        for (int i = 0; i < frames; i++)
            for (int ch = 0; ch < (stereo ? 2 : 1); ch++) {
                if (!lbrr[ch][i]) continue;
                if (stereo && ch == 0) {                                     // weights, and the mid-only flag unless the side has an LBRR frame too
                    rd.icdf(ICDF_SILK_STEREO_STAGE1);
                    rd.icdf(ICDF_SILK_STEREO_STAGE2); rd.icdf(ICDF_SILK_STEREO_STAGE3);
                    rd.icdf(ICDF_SILK_STEREO_STAGE2); rd.icdf(ICDF_SILK_STEREO_STAGE3);
                    if (!lbrr[1][i]) rd.icdf(ICDF_SILK_MID_ONLY);
                }
                ChanParser{scratch[ch], rd, dropped}.frame(bw, subframes, sf_size, true, i == 0 || !lbrr[ch][i - 1]);
            }
    }
    for (int i = 0; i < frames; i++) {
        OgSilkFrame &f = new_frame();
        const bool first = i == 0;
        bool midonly = false;
        if (stereo) {
            stereo_weights();
            midonly = side_vad[i] ? false : rd.icdf(ICDF_SILK_MID_ONLY) != 0;
        }
        f.w0 = st.w0;
        f.w1 = st.w1;
        ChanParser{st.mid, rd, f.ch[0]}.frame(bw, subframes, sf_size, mid_vad[i], first);
        if (stereo && !midonly) ChanParser{st.side, rd, f.ch[1]}.frame(bw, subframes, sf_size, side_vad[i], first);
        else side_absent(f, midonly);
        if (stereo && st.stereo_out) {                                       // unmix_ms consumes then updates prev (:2058-2059)
            st.prev_w0 = st.w0;
            st.prev_w1 = st.w1;
        }
    }
    return frames;
}

int silk_conceal(SilkHostState &st, int bw, int duration_ms, int stereo, int lost, std::vector<OgSilkFrame> &out)
{
    const int frames = duration_ms <= 20 ? 1 : duration_ms / 20;
    const int subframes = duration_ms == 10 ? 2 : 4;
    const int sf_size = bw == 0 ? 40 : bw == 1 ? 60 : 80;
    if (!st.mid.coded) return 0;
    for (int i = 0; i < frames; i++) {
        out.emplace_back();
        OgSilkFrame &f = out.back();
        std::memset(&f, 0, sizeof f);
        f.bandwidth = uint8_t(bw);
        f.f_size = uint16_t(sf_size * subframes);
        f.flags = uint8_t((stereo ? OG_SF_STEREO : 0) | (st.stereo_out ? OG_SF_STEREO_OUT : 0));
        if (st.resampler_reset) { f.flags |= OG_SF_RESAMPLER_RESET; st.resampler_reset = false; }
        f.prev_w0 = st.prev_w0;
        f.prev_w1 = st.prev_w1;
        f.w0 = st.w0;
        f.w1 = st.w1;
        conceal_chan(st.mid, f.ch[0], bw, subframes, sf_size, lost + i);
        if (stereo && st.side.coded) conceal_chan(st.side, f.ch[1], bw, subframes, sf_size, lost + i);
        else {
            f.ch[1].order = f.ch[0].order; f.ch[1].subframes = uint8_t(subframes); f.ch[1].sf_size = uint8_t(sf_size);
            if (stereo) f.flags |= OG_SF_MID_ONLY;
            if (stereo && st.side.flushed) { f.ch[1].flags = OG_S_RESET; st.side.flushed = false; }      // (as in side_absent)
        }
        if (stereo && st.stereo_out) { st.prev_w0 = st.w0; st.prev_w1 = st.w1; }
    }
    return frames;
}

}  // namespace og
