// CELT frame parser: the sequential (range-decoder) half of the reference's Celt::decode
// (lu-zero/opus src/celt/decoder.rs:2222-2323) run "dry": every symbol is decoded and all integer
// control is evaluated, but no coefficient is touched.  Output = one OgCeltHdr + its leaf records,
// which the HIP kernels (hip/celt_launch.hip: celt_pvq.h, celt_plan.h, celt_frame.h) turn into PCM.
//
// Where the reference is defective or stops (SURVEY.md section 8 notes) the behaviour is RFC 6716's.
#include <algorithm>
#include <cmath>
#include <cstring>
#include "parse.hpp"

namespace og {

void CeltHostState::reset()
{
    std::memset(oldE, 0, sizeof oldE);
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 21; i++) logE1[c][i] = logE2[c][i] = -28.f;
    pf_period = pf_period_old = 0;
    pf_gain_q = pf_gain_q_old = pf_tapset = pf_tapset_old = 0;
    rng = 0;
    pending_reset = true;
}

namespace {

constexpr int BITRES = 3;
const float kAlpha[4] = {29440.f / 32768.f, 26112.f / 32768.f, 21248.f / 32768.f, 16384.f / 32768.f};   // :101-106
const float kBeta[4] = {1.f - 30147.f / 32768.f, 1.f - 22282.f / 32768.f, 1.f - 12124.f / 32768.f, 1.f - 6554.f / 32768.f};

inline int fmul16(int a, int b) { return (16384 + int(int16_t(a)) * int(int16_t(b))) >> 15; }   // bitexact.rs:11-15
int bx_cos(int x)                                                                                  // bitexact.rs:18-22
{
    int x2 = int16_t((4096 + int(int16_t(x)) * int(int16_t(x))) >> 13);
    x2 = (32767 - x2) + fmul16(x2, -7651 + fmul16(x2, 8277 + fmul16(-626, x2)));
    return int16_t(1 + x2);
}
int bx_log2tan(int isin, int icos)                                                                 // bitexact.rs:25-36
{
    int lc = ilog(icos), ls = ilog(isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * 2048 + fmul16(isin, fmul16(isin, -2597) + 7932) - fmul16(icos, fmul16(icos, -2597) + 7932);
}
inline uint32_t isqrt32(uint32_t v) { return uint32_t(std::sqrt(double(v))); }   // exact: the double square root of v < 2^32 never rounds up to the next integer

// bits2pulses (celt/decoder.rs:738-764) tabulated: for every row of the pulse cache, the pulse-count index chosen for a budget
// of `bits` eighth-bits.  Cache entries are bytes, so every budget above 255 chooses as 256 does and every negative one as -1.
struct PulseLut {
    uint8_t q[24][258];
    uint8_t row[105];
    PulseLut()
    {
        int nrows = 0, off_of[24];
        for (int i = 0; i < 105; i++) {
            const int off = CELT_CACHE_INDEX[i];
            row[i] = 0;
            if (off < 0) continue;
            int r = 0;
            while (r < nrows && off_of[r] != off) r++;
            if (r == nrows) {
                off_of[nrows++] = off;
                const uint8_t *cache = CELT_CACHE_BITS + off;
                for (int bits = -1; bits <= 256; bits++) {
                    int lo = 0, hi = cache[0];
                    for (int it = 0; it < 6; it++) {
                        int mid = (lo + hi + 1) >> 1;
                        if (int(cache[mid]) >= bits) hi = mid; else lo = mid;
                    }
                    q[r][bits + 1] = uint8_t((bits - (lo == 0 ? -1 : int(cache[lo])) <= int(cache[hi]) - bits) ? lo : hi);
                }
            }
            row[i] = uint8_t(r);
        }
    }
};
const PulseLut kPulseLut;

inline uint32_t U(int n, int k) { return CELT_PVQ_U[CELT_PVQ_U_ROW[std::min(n, k)] + std::max(n, k)]; }   // :412, :663
inline uint32_t V(int n, int k) { return U(n, k) + U(n, k + 1); }                                          // :666

struct Theta { int itheta, inv, imid, iside, delta, qalloc; };

class FrameParser {
public:
    FrameParser(RangeDec &rd, std::vector<OgCeltLeaf> &leaves, OgCeltHdr *hdr) : rd_(rd), leaves_(leaves), hdr_(hdr) {}

    RangeDec &rd_;
    std::vector<OgCeltLeaf> &leaves_;
    OgCeltHdr *hdr_;
    int band_ = 0, intensity_ = 0, tf_change_ = 0, remaining_ = 0;

    // A band's leaves are grouped for the device walk as they are emitted: X then Y, fill leaves in front of coded ones, decode
    // order inside each group.  A band holds at most 16 leaves per channel (four levels of splits).
    OgCeltLeaf group_[4][32];
    int cnt_[4] = {0, 0, 0, 0};
    bool fold_[4] = {false, false, false, false}, overflow_ = false;

    void emit(int off, int n, int k, int B, uint32_t index, bool dead, float gain, int cmshift, int chy, bool lowband, bool n1)
    {
        const int g = (chy ? 2 : 0) + (k == 0 && !n1 ? 0 : 1);
        if (cnt_[g] == 32) { overflow_ = true; return; }
        OgCeltLeaf &l = group_[g][cnt_[g]++];
        fold_[g] |= lowband;
        l.index = index; l.gain = gain;
        l.off = uint8_t(off); l.n = uint8_t(n); l.k = uint8_t(k);
        l.cmshift = uint8_t(cmshift); l.band = uint8_t(band_);
        l.pos = uint16_t((int(CELT_EBANDS[band_]) << hdr_->lm) + off);
        const int logb = 31 - __builtin_clz(unsigned(B));
        l.meta = uint8_t(logb | (dead ? OG_L_FILL_DEAD : 0) | (chy ? OG_L_CH_Y : 0) | (n1 ? OG_L_N1 : 0) | (lowband ? OG_L_HAS_LOWBAND : 0));
    }

    void close_band(OgCeltBand &bd)
    {
        for (int g = 0; g < 4; g++) leaves_.insert(leaves_.end(), group_[g], group_[g] + cnt_[g]);
        if (fold_[0]) bd.flags |= OG_B_FOLD_X;
        if (fold_[2]) bd.flags |= OG_B_FOLD_Y;
        bd.fx = uint8_t(cnt_[0]); bd.nx = uint8_t(cnt_[0] + cnt_[1]);
        bd.fy = uint8_t(cnt_[2]); bd.ny = uint8_t(cnt_[2] + cnt_[3]);
        for (int g = 0; g < 4; g++) { cnt_[g] = 0; fold_[g] = false; }
    }

    void band_n1(bool stereo, int chy)                         // celt/decoder.rs:1466-1495
    {
        // stereo: both channels' signs in one call; otherwise the one channel the call is for (a dual-stereo band comes here
        // once per channel: the second time for Y)
        for (int c = 0; c < (stereo ? 2 : 1); c++) {
            uint32_t sign = 0;
            if (remaining_ >= 1 << BITRES) { sign = rd_.bits(1); remaining_ -= 1 << BITRES; }
            emit(0, 1, 0, 1, sign, false, 1.f, 0, stereo ? c : chy, false, true);
        }
    }

    Theta theta(int N, int &b, int B, int B0, int LM, bool stereo)   // celt/decoder.rs:1497-1593
    {
        Theta t{0, 0, 0, 0, 0, 0};
        int pulse_cap = CELT_LOG_N[band_] + LM * (1 << BITRES);
        int offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
        int qn;
        {
            int N2 = 2 * N - 1;
            if (stereo && N == 2) N2--;
            int qb = std::min(b - pulse_cap - (4 << BITRES), (b + N2 * offset) / N2);
            qb = std::min(8 << BITRES, qb);
            if (qb < (1 << BITRES >> 1)) qn = 1;
            else { qn = CELT_QN_EXP2[qb & 7] >> (14 - (qb >> BITRES)); qn = (qn + 1) >> 1 << 1; }
        }
        if (stereo && band_ >= intensity_) qn = 1;
        int tell = rd_.tell_frac();
        if (qn != 1) {
            int x;
            if (stereo && N > 2) {                              // step pdf (entropy.rs:487-512, RFC form)
                int x0 = qn / 2, ft = 3 * (x0 + 1) + x0;
                int fs = rd_.decode(ft);
                x = fs < (x0 + 1) * 3 ? fs / 3 : x0 + 1 + (fs - (x0 + 1) * 3);
                rd_.update(x <= x0 ? 3 * x : (x - 1 - x0) + (x0 + 1) * 3, x <= x0 ? 3 * (x + 1) : (x - x0) + (x0 + 1) * 3, ft);
            } else if (B0 > 1 || stereo) {
                x = rd_.uint(qn + 1);
            } else {                                            // triangular pdf (entropy.rs:514-538)
                int h = qn >> 1, ft = (h + 1) * (h + 1), fs, fl;
                int fm = rd_.decode(ft);
                if (fm < (h * (h + 1) >> 1)) {
                    x = (isqrt32(8 * uint32_t(fm) + 1) - 1) >> 1;
                    fs = x + 1;
                    fl = x * (x + 1) >> 1;
                } else {
                    x = (2 * (qn + 1) - isqrt32(8 * uint32_t(ft - fm - 1) + 1)) >> 1;
                    fs = qn + 1 - x;
                    fl = ft - ((qn + 1 - x) * (qn + 2 - x) >> 1);
                }
                rd_.update(fl, fl + fs, ft);
            }
            t.itheta = x * 16384 / qn;
        } else if (stereo) {
            if (b > 2 << BITRES && remaining_ > 2 << BITRES) t.inv = rd_.bit_logp(2);
        }
        t.qalloc = rd_.tell_frac() - tell;
        b -= t.qalloc;
        if (t.itheta == 0) { t.imid = 32767; t.iside = 0; t.delta = -16384; }          // side fill dies (:1567-1571)
        else if (t.itheta == 16384) { t.imid = 0; t.iside = 32767; t.delta = 16384; }   // mid fill dies (:1572-1577)
        else {
            t.imid = bx_cos(t.itheta);
            t.iside = bx_cos(16384 - t.itheta);
            t.delta = fmul16((N - 1) << 7, bx_log2tan(t.iside, t.imid));
        }
        return t;
    }

    // celt/decoder.rs:1772-1790 + :1863-1975 (split) + :1601-1663 (leaf)
    void partition(int off, int N, int b, int B, int LM, float gain, bool dead, int cmshift, int chy, bool lowband)
    {
        const uint8_t *cache = CELT_CACHE_BITS + CELT_CACHE_INDEX[(LM + 1) * 21 + band_];
        if (LM != -1 && b > cache[cache[0]] + 12 && N > 2) {
            int B0 = B;
            N >>= 1;
            LM -= 1;
            B = (B + 1) >> 1;
            Theta t = theta(N, b, B, B0, LM, false);
            const bool mid_dead = dead || t.itheta == 16384, side_dead = dead || t.itheta == 0;
            float mid = (1.f / 32768) * t.imid, side = (1.f / 32768) * t.iside;
            int delta = t.delta;
            if (B0 > 1 && (t.itheta & 0x3fff)) {
                if (t.itheta > 8192) delta -= delta >> (4 - LM);
                else delta = std::min(0, delta + (N << BITRES >> (5 - LM)));
            }
            int mbits = std::max(0, std::min(b, (b - delta) / 2));
            int sbits = b - mbits;
            remaining_ -= t.qalloc;
            int rebalance = remaining_;
            if (mbits >= sbits) {
                partition(off, N, mbits, B, LM, gain * mid, mid_dead, cmshift, chy, lowband);
                rebalance = mbits - (rebalance - remaining_);
                if (rebalance > 3 << BITRES && t.itheta != 0) sbits += rebalance - (3 << BITRES);
                partition(off + N, N, sbits, B, LM, gain * side, side_dead, cmshift + (B0 >> 1), chy, lowband);
            } else {
                partition(off + N, N, sbits, B, LM, gain * side, side_dead, cmshift + (B0 >> 1), chy, lowband);
                rebalance = sbits - (rebalance - remaining_);
                if (rebalance > 3 << BITRES && t.itheta != 16384) mbits += rebalance - (3 << BITRES);
                partition(off, N, mbits, B, LM, gain * mid, mid_dead, cmshift, chy, lowband);
            }
            return;
        }
        // bits2pulses / pulses2bits (celt/decoder.rs:738-764; RFC 6716 rate.h offsets)
        int q = kPulseLut.q[kPulseLut.row[(LM + 1) * 21 + band_]][std::max(-1, std::min(256, b - 1)) + 1];
        int curr = q ? cache[q] + 1 : 0;
        remaining_ -= curr;
        while (remaining_ < 0 && q > 0) {
            remaining_ += curr;
            q--;
            curr = q ? cache[q] + 1 : 0;
            remaining_ -= curr;
        }
        if (q != 0) {
            int K = q < 8 ? q : (8 + (q & 7)) << ((q >> 3) - 1);                 // :1627-1631
            uint32_t idx = rd_.uint(V(N, K));                                    // :670
            emit(off, N, K, B, idx, dead, gain, cmshift, chy, lowband, false);
        } else {
            emit(off, N, 0, B, 0, dead, gain, cmshift, chy, lowband, false);     // fill path runs on the device
        }
    }

    // level-0 wrapper (celt/decoder.rs:1708-1768): only the integer block bookkeeping matters here
    void quant_band(int N, int b, int B, int LM, float gain, int chy, bool lowband)
    {
        if (N == 1) { band_n1(false, chy); return; }
        int N_B = N / B, tf = tf_change_, recombine = tf > 0 ? tf : 0;
        B >>= recombine;
        N_B <<= recombine;
        while ((N_B & 1) == 0 && tf < 0) { B <<= 1; N_B >>= 1; tf++; }
        partition(0, N, b, B, LM, gain, false, 0, chy, lowband);
    }

    void quant_band_stereo(int N, int b, int B, int LM, bool lowband)       // celt/decoder.rs:1795-1862 + RFC bands.c
    {
        OgCeltBand &bd = hdr_->band[band_];
        if (N == 1) { band_n1(true, 0); return; }
        bd.flags |= OG_B_JOINT;
        Theta t = theta(N, b, B, B, LM, true);
        bd.imid = uint16_t(t.imid);
        bd.iside = uint16_t(t.iside);
        if (t.inv) bd.flags |= OG_B_INV;
        float side = (1.f / 32768) * t.iside;
        if (N == 2) {
            int mbits = b, sbits = 0;
            if (t.itheta != 0 && t.itheta != 16384) sbits = 1 << BITRES;
            mbits -= sbits;
            int c = t.itheta > 8192;
            remaining_ -= t.qalloc + sbits;
            int sign = 0;
            if (sbits) sign = rd_.bits(1);
            if (c) bd.flags |= OG_B_N2SWAP;
            if (sign) bd.flags |= OG_B_N2NEG;
            quant_band(N, mbits, B, LM, 1.f, c, lowband);        // orig_fill: OG_B_MIDFILL0 stays clear
        } else {
            if (t.itheta == 16384) bd.flags |= OG_B_MIDFILL0;
            int mbits = std::max(0, std::min(b, (b - t.delta) / 2));
            int sbits = b - mbits;
            remaining_ -= t.qalloc;
            int rebalance = remaining_;
            if (mbits >= sbits) {
                quant_band(N, mbits, B, LM, 1.f, 0, lowband);
                rebalance = mbits - (rebalance - remaining_);
                if (rebalance > 3 << BITRES && t.itheta != 0) sbits += rebalance - (3 << BITRES);
                quant_band(N, sbits, B, LM, side, 1, false);
            } else {
                quant_band(N, sbits, B, LM, side, 1, false);
                rebalance = sbits - (rebalance - remaining_);
                if (rebalance > 3 << BITRES && t.itheta != 16384) mbits += rebalance - (3 << BITRES);
                quant_band(N, mbits, B, LM, 1.f, 0, lowband);
            }
        }
    }
};

}  // namespace

int celt_parse_frame(CeltHostState &st, RangeDec &rd, int N, int start, int end, int C, OgCeltHdr *hdr,
                     std::vector<OgCeltLeaf> &leaves, float conceal_decay)
{
    const bool conceal = conceal_decay >= 0.f;
    const int LM = ilog(N / 120) - 1, M = 1 << LM, CC = st.channels;
    int total_bits = rd.storage_bits(), tell = rd.tell();
    std::memset(hdr, 0, sizeof *hdr);
    hdr->leaf_begin = uint32_t(leaves.size());
    hdr->seed = st.rng;
    hdr->lm = uint8_t(LM);
    hdr->start = uint8_t(start);
    hdr->end = uint8_t(end);
    if (C == 2) hdr->flags |= OG_F_STEREO;
    if (CC == 2) hdr->flags |= OG_F_OUT_STEREO;
    if (st.pending_reset) { hdr->flags |= OG_F_RESET; st.pending_reset = false; }

    if (C == 1) for (int i = 0; i < 21; i++) st.oldE[0][i] = std::max(st.oldE[0][i], st.oldE[1][i]);   // :2266-2274

    bool silence = false;                                          // :2237-2248
    if (conceal) silence = false;
    else if (tell >= total_bits) silence = true;
    else if (tell == 1) silence = rd.bit_logp(15);
    if (silence) { rd.to_end(); tell = total_bits; hdr->flags |= OG_F_SILENCE; }

    int pf_pitch = 0, pf_qg1 = 0, pf_tapset = 0;                   // :876-901
    if (start == 0 && tell + 16 <= total_bits) {
        if (rd.bit_logp(1)) {
            int octave = rd.uint(6);
            pf_pitch = (16 << octave) + rd.bits(4 + octave) - 1;
            pf_qg1 = rd.bits(3) + 1;
            if (rd.tell() + 2 <= total_bits) pf_tapset = rd.icdf(ICDF_CELT_TAPSET);
        }
        tell = rd.tell();
    }
    bool transient = false;                                        // :2255-2264
    if (LM > 0 && tell + 3 <= total_bits) { transient = rd.bit_logp(3); tell = rd.tell(); }
    if (transient) hdr->flags |= OG_F_TRANSIENT;
    const int B = transient ? M : 1;
    bool intra = tell + 3 <= total_bits ? rd.bit_logp(3) : false;

    {   // coarse energy, :903-992
        const uint8_t *model = intra ? CELT_E_PROB_INTRA[LM] : CELT_E_PROB_INTER[LM];
        const float coef = intra ? 0.f : kAlpha[LM], beta = intra ? 1.f - 4915.f / 32768.f : kBeta[LM];
        float prev[2] = {0.f, 0.f};
        for (int i = 0; i < 21; i++)
            for (int c = 0; c < C; c++) {
                if (i < start || i >= end) { st.oldE[c][i] = 0.f; continue; }
                if (conceal) { st.oldE[c][i] = std::max(st.oldE[c][i] - conceal_decay, -28.f); continue; }
                int left = total_bits - rd.tell(), qi;
                if (left >= 15) { int pi = 2 * std::min(i, 20); qi = rd.laplace(model[pi] << 7, model[pi + 1] << 6); }
                else if (left >= 2) { qi = rd.icdf(ICDF_CELT_ENERGY_SMALL); qi = (qi >> 1) ^ -(qi & 1); }
                else if (left >= 1) qi = -rd.bit_logp(1);
                else qi = -1;
                float q = float(qi);
                st.oldE[c][i] = std::max(-9.f, st.oldE[c][i]) * coef + prev[c] + q;
                prev[c] += beta * q;
            }
    }
    int tf_res[21] = {0};
    bool overflow = false;
    {   // tf_decode, :994-1033
        int budget = total_bits, logp = transient ? 2 : 4, curr = 0, changed = 0;
        tell = rd.tell();
        int rsv = LM > 0 && tell + logp + 1 <= budget;
        budget -= rsv;
        for (int i = start; i < end; i++) {
            if (tell + logp <= budget) { curr ^= rd.bit_logp(logp); tell = rd.tell(); changed |= curr; }
            tf_res[i] = curr;
            logp = transient ? 4 : 5;
        }
        int sel = 0;
        if (rsv && CELT_TF_SELECT[LM][transient][0][changed] != CELT_TF_SELECT[LM][transient][1][changed]) sel = rd.bit_logp(1);
        for (int i = start; i < end; i++) tf_res[i] = CELT_TF_SELECT[LM][transient][sel][tf_res[i]];
    }
    // ---- allocation, :1035-1447 (RFC 6716 rate.c arithmetic) ----
    int spread = 2;
    if (rd.tell() + 4 <= total_bits) spread = rd.icdf(ICDF_CELT_SPREAD);
    hdr->spread = uint8_t(spread);
    int cap[21], offsets[21] = {0}, pulses[21] = {0}, ebits[21] = {0}, eprio[21] = {0};
    for (int i = 0; i < 21; i++) cap[i] = (CELT_CAPS[LM][C - 1][i] + 64) * C * (CELT_BAND_WIDTH[i] << LM) >> 2;
    {
        int dynalloc_logp = 6, tb = total_bits << BITRES, tf = rd.tell_frac();
        for (int i = start; i < end; i++) {
            int width = C * CELT_BAND_WIDTH[i] << LM;
            int quanta = std::min(width << BITRES, std::max(6 << BITRES, width));
            int logp = dynalloc_logp, boost = 0;
            while (tf + (logp << BITRES) < tb && boost < cap[i]) {
                int flag = rd.bit_logp(logp);
                tf = rd.tell_frac();
                if (!flag) break;
                boost += quanta; tb -= quanta; logp = 1;
            }
            offsets[i] = boost;
            if (boost > 0) dynalloc_logp = std::max(2, dynalloc_logp - 1);
        }
        int alloc_trim = tf + (6 << BITRES) <= tb ? rd.icdf(ICDF_CELT_ALLOC_TRIM) : 5;
        int bits = (total_bits << BITRES) - rd.tell_frac() - 1;
        int anti_rsv = transient && LM >= 2 && bits >= (LM + 2) << BITRES ? 1 << BITRES : 0;
        bits -= anti_rsv;

        // clt_compute_allocation
        int total = std::max(bits, 0), skip_start = start;
        int skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
        total -= skip_rsv;
        int intensity_rsv = 0, dual_rsv = 0;
        if (C == 2) {
            intensity_rsv = CELT_LOG2_FRAC[end - start];
            if (intensity_rsv > total) intensity_rsv = 0;
            else { total -= intensity_rsv; dual_rsv = total >= 1 << BITRES ? 1 << BITRES : 0; total -= dual_rsv; }
        }
        int thresh[21], trim_off[21], b1[21], b2[21];
        for (int j = start; j < end; j++) {
            int w = CELT_BAND_WIDTH[j];
            thresh[j] = std::max(C << BITRES, (3 * w << LM << BITRES) >> 4);
            trim_off[j] = C * w * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES)) >> 6;
            if ((w << LM) == 1) trim_off[j] -= C << BITRES;
        }
        auto vec_bits = [&](int row, int j) {
            int v = C * CELT_BAND_WIDTH[j] * CELT_ALLOC_VECTORS[row][j] << LM >> 2;
            if (v > 0) v = std::max(0, v + trim_off[j]);
            return v;
        };
        int lo = 1, hi = 10;
        do {
            int mid = (lo + hi) >> 1, psum = 0;
            bool done = false;
            for (int j = end; j-- > start;) {
                const int bj = vec_bits(mid, j) + offsets[j];                 // selects, not branches: the outcome is data
                done |= bj >= thresh[j];
                psum += done ? std::min(bj, cap[j]) : bj >= C << BITRES ? C << BITRES : 0;
            }
            if (psum > total) hi = mid - 1; else lo = mid + 1;
        } while (lo <= hi);
        hi = lo--;
        for (int j = start; j < end; j++) {
            int v1 = vec_bits(lo, j);
            int v2 = hi >= 11 ? cap[j] : vec_bits(hi, j);
            if (hi >= 11 && v2 > 0) v2 = std::max(0, v2 + trim_off[j]);
            if (lo > 0) v1 += offsets[j];
            v2 += offsets[j];
            if (offsets[j] > 0) skip_start = j;
            b1[j] = v1;
            b2[j] = std::max(0, v2 - v1);
        }
        // interp_bits2pulses
        const int alloc_floor = C << BITRES, stereo = C > 1, logM = LM << BITRES;
        auto interp = [&](int m, int j) { return b1[j] + (m * b2[j] >> 6); };
        lo = 0; hi = 64;
        for (int it = 0; it < 6; it++) {
            int mid = (lo + hi) >> 1, psum = 0;
            bool done = false;
            for (int j = end; j-- > start;) {
                const int t = interp(mid, j);
                done |= t >= thresh[j];
                psum += done ? std::min(t, cap[j]) : t >= alloc_floor ? alloc_floor : 0;
            }
            if (psum > total) hi = mid; else lo = mid;
        }
        int psum = 0;
        {
            bool done = false;
            for (int j = end; j-- > start;) {
                int t = interp(lo, j);
                done |= t >= thresh[j];
                t = done ? t : t >= alloc_floor ? alloc_floor : 0;
                t = std::min(t, cap[j]);
                pulses[j] = t;
                psum += t;
            }
        }
        int coded;
        for (coded = end;; coded--) {
            int j = coded - 1;
            if (j <= skip_start) { total += skip_rsv; break; }
            int left = total - psum, span = CELT_EBANDS[coded] - CELT_EBANDS[start];
            int percoeff = left / span;
            left -= span * percoeff;
            int rem = std::max(left - (CELT_EBANDS[j] - CELT_EBANDS[start]), 0);
            int band_bits = pulses[j] + percoeff * (CELT_EBANDS[coded] - CELT_EBANDS[j]) + rem;
            if (band_bits >= std::max(thresh[j], alloc_floor + (1 << BITRES))) {
                if (rd.bit_logp(1)) break;
                psum += 1 << BITRES;
                band_bits -= 1 << BITRES;
            }
            psum -= pulses[j] + intensity_rsv;
            if (intensity_rsv > 0) intensity_rsv = CELT_LOG2_FRAC[j - start];
            psum += intensity_rsv;
            if (band_bits >= alloc_floor) { psum += alloc_floor; pulses[j] = alloc_floor; }
            else pulses[j] = 0;
        }
        int intensity = 0, dual_stereo = 0;
        if (intensity_rsv > 0) intensity = start + int(rd.uint(coded + 1 - start));
        if (intensity <= start) { total += dual_rsv; dual_rsv = 0; }
        if (dual_rsv > 0) dual_stereo = rd.bit_logp(1);
        {
            int left = total - psum, span = CELT_EBANDS[coded] - CELT_EBANDS[start];
            int percoeff = left / span;
            left -= span * percoeff;
            for (int j = start; j < coded; j++) pulses[j] += percoeff * CELT_BAND_WIDTH[j];
            for (int j = start; j < coded; j++) {
                int t = std::min(left, int(CELT_BAND_WIDTH[j]));
                pulses[j] += t;
                left -= t;
            }
        }
        int balance = 0, j;
        for (j = start; j < coded; j++) {
            int n = CELT_BAND_WIDTH[j] << LM, bit = pulses[j] + balance, excess;
            if (n > 1) {
                excess = std::max(bit - cap[j], 0);
                pulses[j] = bit - excess;
                int den = C * n + ((C == 2 && n > 2 && !dual_stereo && j < intensity) ? 1 : 0);
                int nclogn = den * (CELT_LOG_N[j] + logM);
                int off = (nclogn >> 1) - den * 21;
                if (n == 2) off += den << BITRES >> 2;
                if (pulses[j] + off < den * 2 << BITRES) off += nclogn >> 2;
                else if (pulses[j] + off < den * 3 << BITRES) off += nclogn >> 3;
                int e = std::max(0, pulses[j] + off + (den << (BITRES - 1)));
                e = (e / den) >> BITRES;
                if (C * e > (pulses[j] >> BITRES)) e = pulses[j] >> stereo >> BITRES;
                e = std::min(e, 8);
                ebits[j] = e;
                eprio[j] = e * (den << BITRES) >= pulses[j] + off;
                pulses[j] -= C * e << BITRES;
            } else {
                excess = std::max(0, bit - (C << BITRES));
                pulses[j] = bit - excess;
                ebits[j] = 0;
                eprio[j] = 1;
            }
            if (excess > 0) {
                int extra_fine = std::min(excess >> (stereo + BITRES), 8 - ebits[j]);
                ebits[j] += extra_fine;
                int extra_bits = extra_fine * C << BITRES;
                eprio[j] = extra_bits >= excess - balance;
                excess -= extra_bits;
            }
            balance = excess;
        }
        for (; j < end; j++) {
            ebits[j] = pulses[j] >> stereo >> BITRES;
            pulses[j] = 0;
            eprio[j] = ebits[j] < 1;
        }
        hdr->intensity = uint8_t(intensity);

        // fine energy, :1449-1464
        for (int i = start; i < end; i++) {
            if (ebits[i] <= 0) continue;
            for (int c = 0; c < C; c++) {
                int q2 = rd.bits(ebits[i]);
                st.oldE[c][i] += (float(q2) + .5f) * float(1 << (14 - ebits[i])) * (1.f / 16384) - .5f;
            }
        }

        // ---- band loop, :2037-2220: budgets, folding sources, leaf emission ----
        FrameParser fp(rd, leaves, hdr);
        fp.intensity_ = intensity;
        const int band_total = (total_bits << BITRES) - anti_rsv;
        bool update_lowband = true;
        int lowband_offset = 0;
        for (int i = start; i < end; i++) {
            OgCeltBand &bd = hdr->band[i];
            const int n = CELT_BAND_WIDTH[i] << LM;
            int tellf = rd.tell_frac(), b;
            fp.band_ = i;
            if (i != start) balance -= tellf;
            int remaining_bits = band_total - tellf - 1;
            fp.remaining_ = remaining_bits;
            if (i <= coded - 1) {
                int curr_balance = balance / std::min(3, coded - i);
                b = std::max(0, std::min(16383, std::min(remaining_bits + 1, pulses[i] + curr_balance)));
            } else b = 0;
            if (M * CELT_EBANDS[i] - n >= M * CELT_EBANDS[start] && (update_lowband || lowband_offset == 0)) lowband_offset = i;
            fp.tf_change_ = tf_res[i];
            bd.tf_change = int8_t(tf_res[i]);
            bool lowband = false;
            bd.lowband = 0xFF;
            bd.fold_lo = bd.fold_hi = 0xFF;
            bd.leaf0 = uint16_t(leaves.size() - hdr->leaf_begin);
            if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_res[i] < 0)) {
                const int norm_offset = M * CELT_EBANDS[start];
                int eff = std::max(0, M * CELT_EBANDS[lowband_offset] - norm_offset - n);
                bd.lowband = uint8_t(eff / M);
                lowband = true;
                int fold_start = lowband_offset, fold_end = lowband_offset - 1;            // :2097-2120
                while (M * CELT_EBANDS[--fold_start] > eff + norm_offset) {}
                while (M * CELT_EBANDS[++fold_end] < eff + norm_offset + n) {}
                bd.fold_lo = uint8_t(fold_start);
                bd.fold_hi = uint8_t(fold_end);
            }
            if (i == end - 1) bd.flags |= OG_B_LAST;
            if (dual_stereo && i == intensity) { dual_stereo = 0; bd.flags |= OG_B_MERGE_NORM; }
            if (dual_stereo) {
                bd.flags |= OG_B_DUAL;
                fp.quant_band(n, b / 2, B, LM, 1.f, 0, lowband);
                fp.quant_band(n, b / 2, B, LM, 1.f, 1, lowband);
            } else if (C == 2) {
                fp.quant_band_stereo(n, b, B, LM, lowband);
            } else {
                fp.quant_band(n, b, B, LM, 1.f, 0, lowband);
            }
            fp.close_band(bd);
            balance += pulses[i] + tellf;
            update_lowband = b > (n << BITRES);
        }
        overflow = fp.overflow_;
        bool anti_on = false;
        if (anti_rsv > 0) anti_on = rd.bits(1);
        if (anti_on) hdr->flags |= OG_F_ANTICOLLAPSE;
        {   // energy finalise (RFC 6716 4.3.2.2)
            int bits_left = total_bits - rd.tell();
            for (int prio = 0; prio < 2; prio++)
                for (int i = start; i < end && bits_left >= C; i++) {
                    if (ebits[i] >= 8 || eprio[i] != prio) continue;
                    for (int c = 0; c < C; c++) {
                        int q2 = rd.bits(1);
                        st.oldE[c][i] += (float(q2) - .5f) * float(1 << (14 - ebits[i] - 1)) * (1.f / 16384);
                        bits_left--;
                    }
                }
        }
        for (int i = start; i < end; i++) hdr->pulses_depth[i] = uint16_t(((1 + pulses[i]) / CELT_BAND_WIDTH[i]) >> LM);
    }
    hdr->nleaves = uint16_t(leaves.size() - hdr->leaf_begin);

    // what the device needs of the energy state
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 21; i++) {
            float p1 = st.logE1[c][i], p2 = st.logE2[c][i];
            if (C == 1) { p1 = std::max(p1, st.logE1[1][i]); p2 = std::max(p2, st.logE2[1][i]); }
            hdr->prev_min[c][i] = std::min(p1, p2);
            hdr->energy[c][i] = st.oldE[c < C ? c : 0][i];
        }
    if (silence) for (int c = 0; c < C; c++) for (int i = 0; i < 21; i++) st.oldE[c][i] = -28.f;

    // post-filter parameter hand-over (RFC 6716 celt_decoder.c; state fields celt/decoder.rs:22-31)
    hdr->pf_period[0] = st.pf_period_old; hdr->pf_gain_q[0] = st.pf_gain_q_old; hdr->pf_tapset[0] = st.pf_tapset_old;
    hdr->pf_period[1] = st.pf_period;     hdr->pf_gain_q[1] = st.pf_gain_q;     hdr->pf_tapset[1] = st.pf_tapset;
    hdr->pf_period[2] = uint16_t(pf_pitch); hdr->pf_gain_q[2] = uint8_t(pf_qg1); hdr->pf_tapset[2] = uint8_t(pf_tapset);
    st.pf_period_old = st.pf_period; st.pf_gain_q_old = st.pf_gain_q; st.pf_tapset_old = st.pf_tapset;
    st.pf_period = uint16_t(pf_pitch); st.pf_gain_q = uint8_t(pf_qg1); st.pf_tapset = uint8_t(pf_tapset);
    if (LM != 0) { st.pf_period_old = st.pf_period; st.pf_gain_q_old = st.pf_gain_q; st.pf_tapset_old = st.pf_tapset; }

    if (C == 1) std::memcpy(st.oldE[1], st.oldE[0], sizeof st.oldE[0]);
    if (!transient) {
        std::memcpy(st.logE2, st.logE1, sizeof st.logE1);
        std::memcpy(st.logE1, st.oldE, sizeof st.oldE);
    } else {
        for (int c = 0; c < 2; c++) for (int i = 0; i < 21; i++) st.logE1[c][i] = std::min(st.logE1[c][i], st.oldE[c][i]);
    }
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 21; i++)
            if (i < start || i >= end) { st.oldE[c][i] = 0.f; st.logE1[c][i] = st.logE2[c][i] = -28.f; }
    if (conceal) { st.rng = st.rng * 1664525u + 1013904223u; return 0; }
    st.rng = rd.rng();                                             // :2322
    return rd.tell() > total_bits || overflow ? -1 : 0;
}

}  // namespace og
