/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under opus_amd/ may include, link or call this.
 *
 * CPU restatement (plain C, f32) of the reference's CELT decoder, lu-zero/opus src/celt/decoder.rs,
 * completed from RFC 6716 section 4.3 where the reference stops or is defective (SURVEY.md section 8
 * notes).  Pinning status:
 *   pinned by reference KATs : cwrsi (celt/decoder.rs:2414-2434), extract_collapse_mask (:2398-2411),
 *                              stereo_merge (:2330-2396), haar1 (:2436-2481), range decoder (entropy.rs)
 *   no reference test, pinned : tf / allocation / theta / bits2pulses / exp_rotation, and everything after PVQ band decoding
 *   from the signal side        (anti-collapse, denormalise, IMDCT, window, post-filter, de-emphasis: the reference has no such
 *                              code, celt/decoder.rs:2320-2322 ends at decode_bands) are written from RFC 6716 4.3.3-4.3.7.
 *                              What pins them: the RFC's own test vectors through opus_compare (tests/test_opus_compare.py);
 *                              the fast IMDCT against an f64 evaluation of the transform's definition and the window / overlap
 *                              against a textbook f64 MDCT (tests/test_oracle_kats.py: test_imdct_fast_against_f64_direct,
 *                              test_imdct_and_window_reconstruct_a_textbook_mdct); and the encode -> decode loopback of
 *                              tests/test_loopback.py, where this decoder, run forced, ENCODES a known signal and every
 *                              decoder must return the signal (22 dB SNR at 64 kb/s per channel, 28 dB at 128), at every
 *                              frame size, with spreading, joint stereo, short blocks and the post-filter.
 */
#include <math.h>
#include <stdlib.h>
#include "orc_rangedec.h"
#include "orc_api.h"

static const float EMEANS[21] = {   /* RFC 6716 4.3.2 mean energy (log2 units), not in the reference */
    6.437500f, 6.250000f, 5.750000f, 5.312500f, 5.062500f, 4.812500f, 4.500000f, 4.375000f, 4.875000f,
    4.687500f, 4.562500f, 4.437500f, 4.875000f, 4.625000f, 4.312500f, 4.500000f, 4.375000f, 4.625000f,
    4.750000f, 4.437500f, 3.750000f};

static const float ALPHA_COEF[4] = {29440.f / 32768.f, 26112.f / 32768.f, 21248.f / 32768.f, 16384.f / 32768.f};
static const float BETA_COEF[4] = {1.f - 30147.f / 32768.f, 1.f - 22282.f / 32768.f, 1.f - 12124.f / 32768.f,
                                   1.f - 6554.f / 32768.f};   /* celt/decoder.rs:101-113 */

/* ------------------------------------------------------------------ integer helpers */
static inline int frac_mul16(int a, int b) { return (16384 + (int)(int16_t)a * (int)(int16_t)b) >> 15; } /* bitexact.rs:11 */

static int bitexact_cos(int x)   /* celt/bitexact.rs:18-22 */
{
    int v = (int16_t)(((int)(int16_t)x * (int16_t)x + 4096) >> 13);
    return (int16_t)(1 + (32767 - v) + frac_mul16(v, -7651 + frac_mul16(v, 8277 + frac_mul16(-626, v))));
}

static int bitexact_log2tan(int isin, int icos)   /* celt/bitexact.rs:25-36 */
{
    int ls = orc_ilog(isin), lc = orc_ilog(icos);
    isin <<= 15 - ls;
    icos <<= 15 - lc;
    return (ls - lc) * (1 << 11) + frac_mul16(isin, frac_mul16(isin, -2597) + 7932)
           - frac_mul16(icos, frac_mul16(icos, -2597) + 7932);
}

static inline uint32_t pvq_u(int n, int k)   /* celt/decoder.rs:412-414, :663-665 */
{
    int lo = n < k ? n : k, hi = n < k ? k : n;
    return CELT_PVQ_U[CELT_PVQ_U_ROW[lo] + hi];
}

/* celt/decoder.rs:541-660: CWRS index -> signed pulse vector; returns ||y||^2 */
uint32_t orc_cwrsi(int n, int k, uint32_t i, int *y)
{
    uint32_t norm = 0, p, q;
    int s, k0, val;
    while (n > 2) {
        if (k >= n) {                       /* lots of pulses */
            p = pvq_u(n, k + 1);
            s = -(i >= p);
            i -= p & (uint32_t)s;
            k0 = k;
            q = pvq_u(n, n);
            if (q > i) {
                k = n;
                do p = pvq_u(--k, n); while (p > i);
            } else {
                for (p = pvq_u(n, k); p > i; p = pvq_u(n, k)) k--;
            }
            i -= p;
            val = (k0 - k + s) ^ s;
            *y++ = val;
            norm += (uint32_t)(val * val);
        } else {                            /* lots of dimensions */
            p = pvq_u(k, n);
            q = pvq_u(k + 1, n);
            if (p <= i && i < q) {
                i -= p;
                *y++ = 0;
            } else {
                s = -(i >= q);
                i -= q & (uint32_t)s;
                k0 = k;
                do p = pvq_u(--k, n); while (p > i);
                i -= p;
                val = (k0 - k + s) ^ s;
                *y++ = val;
                norm += (uint32_t)(val * val);
            }
        }
        n--;
    }
    p = 2 * (uint32_t)k + 1;                /* n == 2 */
    s = -(i >= p);
    i -= p & (uint32_t)s;
    k0 = k;
    k = (int)((i + 1) >> 1);
    if (k) i -= 2 * (uint32_t)k - 1;
    val = (k0 - k + s) ^ s;
    *y++ = val;
    norm += (uint32_t)(val * val);
    s = -(int)i;                            /* n == 1 */
    val = (k + s) ^ s;
    *y = val;
    norm += (uint32_t)(val * val);
    return norm;
}

uint32_t orc_pvq_v(int n, int k) { return pvq_u(n, k) + pvq_u(n, k + 1); }   /* celt/decoder.rs:666-668 */

/* The inverse of orc_cwrsi (RFC 6716 cwrs.c icwrs; no counterpart in the reference, a decoder): pulse vector -> index. */
uint32_t orc_icwrs(int n, const int *y)
{
    int j = n - 1, k = abs(y[j]);
    uint32_t i = y[j] < 0;
    do {
        j--;
        i += pvq_u(n - j, k);
        k += abs(y[j]);
        if (y[j] < 0) i += pvq_u(n - j, k + 1);
    } while (j > 0);
    return i;
}

/* The decoder run as an encoder (orc_rangedec.h, forced mode): at every value a frame codes, `hook` is asked what to code.
 * kind: 1 coarse energy (a = alphabet: 0 Laplace, 1 {0,-1,+1}, 2 {0,-1}; f = the prediction the value adds to) -> qi
 *       2 fine energy (a = bits; f = the band's energy so far) -> 0 .. 2^bits - 1      3 final bit (a = fine bits so far) -> 0 | 1
 *       4 split angle (a = first bin of the vector in the frame, b = bins per half, c = qn) -> 0 .. qn
 *       5 pulses (a = first bin, b = bins, c = pulses, f = blocks + 16 * spread; y: out, |y|_1 = c)
 *       6 sign of a one-bin band (a = bin) -> 0 | 1       7 stereo angle of a jointly coded band (a = first bin, b = bins, c = qn) -> 0 .. qn
 *       8 a jointly coded band of two bins (a = first bin, c = 1: the side is the vector that is coded, 0: the mid) -> the sign
 *         bit that orients the other one
 * Everything else takes symbol 0 (no silence, no post-filter, no transient, inter energy, no tf change, no spreading rotation, no
 * boosts), except the allocation trim (5), the skip flags (1: code every band) and, for two channels, intensity = none and
 * dual stereo = 1 (each channel's bands coded like a mono band). */
int orc_force_on = 0, orc_force_have = 0;
int64_t orc_force_val = 0;
static OrcEncHook enc_hook = 0;
static int enc_transient = 0, enc_dual = 1, enc_spread = 0;
void orc_set_encoder(OrcEncHook h) { enc_hook = h; orc_force_on = h != 0; orc_force_have = 0; }
/* what the forced run codes for the frame-level tools (defaults: long blocks, dual stereo, no spreading rotation): transient = 1
 * codes short blocks with every band's tf_change brought to 0 (first tf bit 1, the others 0: RFC 6716 table 60 then gives 0 at
 * every LM); dual = 0 codes the bands jointly (kind 7 asks for the stereo angle); spread = 0 .. 3 */
void orc_set_encoder_opts(int transient, int dual_stereo, int spread) { enc_transient = transient; enc_dual = dual_stereo; enc_spread = spread; }
/* ... and for the post-filter: period 15 .. 1022 (0 = off), gain index 0 .. 7 (gain = 0.09375 (index + 1)), tap set 0 .. 2 */
static int enc_pf_period = 0, enc_pf_gain = 0, enc_pf_tapset = 0;
/* ... and, for long blocks, the time-frequency change of every band: 1 = every tf bit's band flagged, tf_select 0 (at 20 ms: -2,
 * two Haar levels more time resolution); 2 = the same with tf_select 1 (-3) */
static int enc_tf = 0;
void orc_set_encoder_tf(int mode) { enc_tf = mode; }
void orc_set_encoder_postfilter(int period, int gain_index, int tapset) { enc_pf_period = period; enc_pf_gain = gain_index; enc_pf_tapset = tapset; }

/* celt/decoder.rs:723-736.  Divergence: the reference ORs element e into bit (e mod B); RFC 6716
 * (vq.c extract_collapse_mask) uses contiguous blocks of N/B, which is what the de-interleaved
 * ("time order") layout of a short-block band means.  Both satisfy the reference's two KATs. */
int orc_extract_collapse_mask(const int *y, int n, int b)
{
    if (b <= 1) return 1;
    int n0 = n / b, mask = 0;
    for (int i = 0; i < b; i++) {
        int t = 0;
        for (int j = 0; j < n0; j++) t |= y[i * n0 + j];
        mask |= (t != 0) << i;
    }
    return mask;
}

void orc_haar1(float *x, int n0, int stride)   /* celt/decoder.rs:453-466 */
{
    const float r = 0.70710678118654752440f;
    n0 >>= 1;
    for (int i = 0; i < stride; i++)
        for (int j = 0; j < n0; j++) {
            float a = x[stride * 2 * j + i], b = x[stride * (2 * j + 1) + i];
            x[stride * 2 * j + i] = (a + b) * r;
            x[stride * (2 * j + 1) + i] = (a - b) * r;
        }
}

static void deinterleave_hadamard(float *x, int n0, int stride, int hadamard)   /* :503-538 */
{
    float tmp[176];
    int n = n0 * stride;
    const uint8_t *ord = CELT_HADAMARD_ORDERY + stride - 2;
    for (int i = 0; i < stride; i++)
        for (int j = 0; j < n0; j++) tmp[(hadamard ? ord[i] : i) * n0 + j] = x[j * stride + i];
    memcpy(x, tmp, n * sizeof(float));
}

static void interleave_hadamard(float *x, int n0, int stride, int hadamard)   /* :472-501 */
{
    float tmp[176];
    int n = n0 * stride;
    const uint8_t *ord = CELT_HADAMARD_ORDERY + stride - 2;
    for (int i = 0; i < stride; i++)
        for (int j = 0; j < n0; j++) tmp[j * stride + i] = x[(hadamard ? ord[i] : i) * n0 + j];
    memcpy(x, tmp, n * sizeof(float));
}

/* celt/decoder.rs:676-721.  The reference's reverse loop and per-block slicing are defective
 * (SURVEY section 8 notes); this is RFC 6716 vq.c exp_rotation1/exp_rotation with dir = -1. */
static void exp_rotation1(float *x, int len, int stride, float c, float s)
{
    float *p = x;
    for (int i = 0; i < len - stride; i++) {
        float x1 = p[0], x2 = p[stride];
        p[stride] = c * x2 + s * x1;
        *p++ = c * x1 - s * x2;
    }
    p = &x[len - 2 * stride - 1];
    for (int i = len - 2 * stride - 1; i >= 0; i--) {
        float x1 = p[0], x2 = p[stride];
        p[stride] = c * x2 + s * x1;
        *p-- = c * x1 - s * x2;
    }
}

void orc_exp_rotation(float *x, int len, int stride, int k, int spread)
{
    if (2 * k >= len || spread == 0) return;
    float gain = (float)len / (float)(len + (20 - 5 * spread) * k);
    float theta = 0.5f * gain * gain;
    float c = cosf(0.5f * (float)M_PI * theta), s = cosf(0.5f * (float)M_PI * (1.f - theta));
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    len /= stride;
    for (int i = 0; i < stride; i++) {
        if (stride2) exp_rotation1(x + i * len, len, stride2, s, c);
        exp_rotation1(x + i * len, len, 1, c, s);
    }
}

static void renormalize_vector(float *x, int n, float gain)   /* celt/decoder.rs:788-794 */
{
    float e = 1e-15f;
    for (int i = 0; i < n; i++) e += x[i] * x[i];
    float g = gain / sqrtf(e);
    for (int i = 0; i < n; i++) x[i] *= g;
}

void orc_stereo_merge(float *x, float *y, float mid, int n)   /* celt/decoder.rs:796-827 */
{
    float xp = 0, side = 0;
    for (int i = 0; i < n; i++) { xp += x[i] * y[i]; side += y[i] * y[i]; }
    xp *= mid;
    float e = mid * mid + side, e0 = e - 2.f * xp, e1 = e + 2.f * xp;
    if (e0 < 6e-4f || e1 < 6e-4f) {      /* RFC 6716 bands.c: copy and return (the reference falls through) */
        memcpy(y, x, n * sizeof(float));
        return;
    }
    float g0 = 1.f / sqrtf(e0), g1 = 1.f / sqrtf(e1);
    for (int i = 0; i < n; i++) {
        float v0 = mid * x[i], v1 = y[i];
        x[i] = g0 * (v0 - v1);
        y[i] = g1 * (v0 + v1);
    }
}

/* ------------------------------------------------------------------ band decoding */
typedef struct {
    OrcRd *rd;
    int band, intensity, spread, tf_change, remaining_bits;
    uint32_t seed;
    float *base[2];         /* channel coefficient arrays, for trace offsets */
    OrcCeltTrace *tr;
    float *hx;              /* the vector being split (for the encoder hook) */
} BandCtx;

static inline uint32_t lcg(uint32_t s) { return 1664525u * s + 1013904223u; }   /* :1595-1599 */

static int bits2pulses(const uint8_t *cache, int bits)   /* celt/decoder.rs:738-756 */
{
    int lo = 0, hi = cache[0];
    bits--;                                   /* RFC 6716 rate.h: bits-- before the search */
    for (int i = 0; i < 6; i++) {
        int mid = (lo + hi + 1) >> 1;
        if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
    }
    return (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) ? lo : hi;
}
static inline int pulses2bits(const uint8_t *cache, int q) { return q == 0 ? 0 : cache[q] + 1; }   /* :758 */

static void trace_leaf(BandCtx *c, float *x, int n, int k, int blocks, uint32_t idx, float gain, const int *y, int kind)
{
    OrcCeltTrace *t = c->tr;
    if (!t) return;
    int ch = (x >= c->base[1] && c->base[1] && x < c->base[1] + 960) ? 1 : 0;
    int off = (int)(x - c->base[ch]);
    if (t->nleaves < ORC_MAX_LEAVES) {
        OrcLeaf *l = &t->leaf[t->nleaves];
        l->band = c->band; l->ch = ch; l->off = off; l->n = n; l->k = k; l->blocks = blocks;
        l->idx = idx; l->gain = gain; l->kind = kind;
    }
    t->nleaves++;
    if (y) for (int i = 0; i < n; i++) t->pulses[ch][off + i] = (int16_t)y[i];
}

static int quant_band_n1(BandCtx *c, float *x, float *y, float *lowband_out)   /* celt/decoder.rs:1466-1495 */
{
    float *p = x;
    for (int ch = 0; ch < (y ? 2 : 1); ch++) {
        int sign = 0;
        if (c->remaining_bits >= 8) {
            if (orc_force_on) {
                int pc = p >= c->base[1] && c->base[1] ? 1 : 0;     /* (a dual-stereo band comes here once per channel, as x) */
                orc_force(enc_hook(6, c->band, pc, (int)(p - c->base[pc]), 1, 0, 0., NULL));
            }
            sign = orc_rd_rawbits(c->rd, 1);
            c->remaining_bits -= 8;
        }
        p[0] = sign ? -1.f : 1.f;
        trace_leaf(c, p, 1, 0, 1, sign, 1.f, NULL, ORC_LEAF_N1);
        p = y;
    }
    if (lowband_out) lowband_out[0] = x[0];
    return 1;
}

typedef struct { int inv, imid, iside, delta, itheta, qalloc; } SplitCtx;

static void compute_theta(BandCtx *c, SplitCtx *s, int n, int *b, int B, int B0, int lm, int stereo, int *fill)
{   /* celt/decoder.rs:1497-1593, integer arithmetic signed as in RFC 6716 bands.c */
    int pulse_cap = CELT_LOG_N[c->band] + lm * 8;
    int offset = (pulse_cap >> 1) - (stereo && n == 2 ? 16 : 4);
    int qn, itheta = 0, inv = 0;
    {
        int n2 = 2 * n - 1;
        if (stereo && n == 2) n2--;
        int qb = (*b + n2 * offset) / n2;
        if (*b - pulse_cap - 32 < qb) qb = *b - pulse_cap - 32;
        if (qb > 64) qb = 64;
        if (qb < 4) qn = 1;
        else { qn = CELT_QN_EXP2[qb & 7] >> (14 - (qb >> 3)); qn = (qn + 1) >> 1 << 1; }
    }
    if (stereo && c->band >= c->intensity) qn = 1;
    int tell = orc_rd_tell_frac(c->rd);
    if (qn != 1) {
        if (orc_force_on) orc_force(stereo ? enc_hook(7, c->band, 0, (int)(c->hx - c->base[0]), n, qn, 0., NULL)
                                           : enc_hook(4, c->band, c->hx >= c->base[1] && c->base[1] ? 1 : 0,
                                                      (int)(c->hx - c->base[c->hx >= c->base[1] && c->base[1] ? 1 : 0]), n, qn, 0., NULL));
        if (stereo && n > 2) itheta = orc_rd_step(c->rd, qn / 2);
        else if (B0 > 1 || stereo) itheta = orc_rd_uniform(c->rd, qn + 1);
        else itheta = orc_rd_triangular(c->rd, qn);
        itheta = itheta * 16384 / qn;
    } else if (stereo) {
        if (*b > 16 && c->remaining_bits > 16) inv = orc_rd_logp(c->rd, 2);
        itheta = 0;
    }
    int qalloc = orc_rd_tell_frac(c->rd) - tell;
    *b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) { imid = 32767; iside = 0; *fill &= (1 << B) - 1; delta = -16384; }
    else if (itheta == 16384) { imid = 0; iside = 32767; *fill &= ((1 << B) - 1) << B; delta = 16384; }
    else {
        imid = bitexact_cos(itheta);
        iside = bitexact_cos(16384 - itheta);
        delta = frac_mul16((n - 1) << 7, bitexact_log2tan(iside, imid));
    }
    s->inv = inv; s->imid = imid; s->iside = iside; s->delta = delta; s->itheta = itheta; s->qalloc = qalloc;
}

static int quant_partition(BandCtx *c, float *x, int n, int b, int B, float *lowband, int lm, float gain, int fill)
{   /* celt/decoder.rs:1772-1790 (split decision), :1863-1975 (split), :1601-1663 (leaf) */
    const uint8_t *cache = CELT_CACHE_BITS + CELT_CACHE_INDEX[(lm + 1) * 21 + c->band];
    int cm = 0, B0 = B;
    if (lm != -1 && b > cache[cache[0]] + 12 && n > 2) {
        SplitCtx s;
        n >>= 1;
        float *y = x + n;
        lm -= 1;
        if (B == 1) fill = (fill & 1) | (fill << 1);
        B = (B + 1) >> 1;
        c->hx = x;
        compute_theta(c, &s, n, &b, B, B0, lm, 0, &fill);
        float mid = (1.f / 32768) * s.imid, side = (1.f / 32768) * s.iside;
        int delta = s.delta;
        if (B0 > 1 && (s.itheta & 0x3fff)) {
            if (s.itheta > 8192) delta -= delta >> (4 - lm);
            else { delta = delta + (n << 3 >> (5 - lm)); if (delta > 0) delta = 0; }
        }
        int mbits = (b - delta) / 2;
        if (mbits > b) mbits = b;
        if (mbits < 0) mbits = 0;
        int sbits = b - mbits;
        c->remaining_bits -= s.qalloc;
        float *lowband2 = lowband ? lowband + n : NULL;
        int rebalance = c->remaining_bits;
        if (mbits >= sbits) {
            cm = quant_partition(c, x, n, mbits, B, lowband, lm, gain * mid, fill);
            rebalance = mbits - (rebalance - c->remaining_bits);
            if (rebalance > 24 && s.itheta != 0) sbits += rebalance - 24;
            cm |= quant_partition(c, y, n, sbits, B, lowband2, lm, gain * side, fill >> B) << (B0 >> 1);
        } else {
            cm = quant_partition(c, y, n, sbits, B, lowband2, lm, gain * side, fill >> B) << (B0 >> 1);
            rebalance = sbits - (rebalance - c->remaining_bits);
            if (rebalance > 24 && s.itheta != 16384) mbits += rebalance - 24;
            cm |= quant_partition(c, x, n, mbits, B, lowband, lm, gain * mid, fill);
        }
        return cm;
    }
    int q = bits2pulses(cache, b);
    int curr = pulses2bits(cache, q);
    c->remaining_bits -= curr;
    while (c->remaining_bits < 0 && q > 0) {
        c->remaining_bits += curr;
        q--;
        curr = pulses2bits(cache, q);
        c->remaining_bits -= curr;
    }
    if (q != 0) {
        int k = q < 8 ? q : (8 + (q & 7)) << ((q >> 3) - 1);
        int y[176];
        if (orc_force_on) {
            int ch = x >= c->base[1] && c->base[1] ? 1 : 0;
            memset(y, 0, sizeof y);
            enc_hook(5, c->band, ch, (int)(x - c->base[ch]), n, k, (double)(B + 16 * c->spread), y);
            orc_force(orc_icwrs(n, y));
        }
        uint32_t idx = orc_rd_uniform(c->rd, orc_pvq_v(n, k));   /* celt/decoder.rs:662-673 */
        float ryy = (float)orc_cwrsi(n, k, idx, y);
        float g = gain / sqrtf(ryy);                              /* celt/decoder.rs:766-786 */
        for (int i = 0; i < n; i++) x[i] = g * (float)y[i];
        orc_exp_rotation(x, n, B, k, c->spread);
        cm = orc_extract_collapse_mask(y, n, B);
        trace_leaf(c, x, n, k, B, idx, gain, y, ORC_LEAF_PVQ);
    } else {
        int cm_mask = (1 << B) - 1;
        fill &= cm_mask;
        if (!fill) {
            memset(x, 0, n * sizeof(float));
            trace_leaf(c, x, n, 0, B, 0, gain, NULL, ORC_LEAF_ZERO);
        } else {
            if (!lowband) {
                for (int i = 0; i < n; i++) { c->seed = lcg(c->seed); x[i] = (float)((int32_t)c->seed >> 20); }
                cm = cm_mask;
                trace_leaf(c, x, n, 0, B, 0, gain, NULL, ORC_LEAF_NOISE);
            } else {
                for (int i = 0; i < n; i++) {
                    c->seed = lcg(c->seed);
                    x[i] = lowband[i] + ((c->seed & 0x8000) ? 1.f / 256 : -1.f / 256);
                }
                cm = fill;
                trace_leaf(c, x, n, 0, B, 0, gain, NULL, ORC_LEAF_FOLD);
            }
            renormalize_vector(x, n, gain);
        }
    }
    return cm;
}

static int quant_band(BandCtx *c, float *x, int n, int b, int B, float *lowband, int lm, float *lowband_out,
                      float gain, float *scratch, int fill)
{   /* celt/decoder.rs:1665-1768 (pre), :1994-2032 (post) */
    int n0 = n, n_b = n / B, n_b0, B0 = B, time_divide = 0, recombine = 0, long_blocks = B == 1;
    int tf_change = c->tf_change;
    if (n == 1) return quant_band_n1(c, x, NULL, lowband_out);
    if (tf_change > 0) recombine = tf_change;
    if (scratch && lowband && (recombine || ((n_b & 1) == 0 && tf_change < 0) || B0 > 1)) {
        memcpy(scratch, lowband, n * sizeof(float));
        lowband = scratch;
    }
    for (int k = 0; k < recombine; k++) {
        if (lowband) orc_haar1(lowband, n >> k, 1 << k);
        fill = CELT_BIT_INTERLEAVE[fill & 0xF] | CELT_BIT_INTERLEAVE[fill >> 4] << 2;
    }
    B >>= recombine;
    n_b <<= recombine;
    while ((n_b & 1) == 0 && tf_change < 0) {
        if (lowband) orc_haar1(lowband, n_b, B);
        fill |= fill << B;
        B <<= 1;
        n_b >>= 1;
        time_divide++;
        tf_change++;
    }
    B0 = B;
    n_b0 = n_b;
    if (B0 > 1 && lowband) deinterleave_hadamard(lowband, n_b >> recombine, B0 << recombine, long_blocks);
    int cm = quant_partition(c, x, n, b, B, lowband, lm, gain, fill);
    if (B0 > 1) interleave_hadamard(x, n_b >> recombine, B0 << recombine, long_blocks);
    n_b = n_b0;
    B = B0;
    for (int k = 0; k < time_divide; k++) {
        B >>= 1;
        n_b <<= 1;
        cm |= cm >> B;
        orc_haar1(x, n_b, B);
    }
    for (int k = 0; k < recombine; k++) {
        cm = CELT_BIT_DEINTERLEAVE[cm];
        orc_haar1(x, n0 >> k, 1 << k);
    }
    B <<= recombine;
    if (lowband_out) {
        float s = sqrtf((float)n0);
        for (int j = 0; j < n0; j++) lowband_out[j] = s * x[j];
    }
    return cm & ((1 << B) - 1);
}

static int quant_band_stereo(BandCtx *c, float *x, float *y, int n, int b, int B, float *lowband, int lm,
                             float *lowband_out, float *scratch, int fill)
{   /* celt/decoder.rs:1795-1862 (stereo half of decode_band); merge + inv from RFC 6716 bands.c */
    if (n == 1) return quant_band_n1(c, x, y, lowband_out);
    int orig_fill = fill, cm;
    SplitCtx s;
    c->hx = x;
    compute_theta(c, &s, n, &b, B, B, lm, 1, &fill);
    float mid = (1.f / 32768) * s.imid, side = (1.f / 32768) * s.iside;
    if (c->tr && c->band < 21) { c->tr->itheta[c->band] = s.itheta; c->tr->inv[c->band] = s.inv; }
    if (n == 2) {
        int mbits = b, sbits = 0, sign = 0;
        if (s.itheta != 0 && s.itheta != 16384) sbits = 8;
        mbits -= sbits;
        int sw = s.itheta > 8192;
        c->remaining_bits -= s.qalloc + sbits;
        float *x2 = sw ? y : x, *y2 = sw ? x : y;
        if (sbits && orc_force_on) orc_force(enc_hook(8, c->band, 0, (int)(x - c->base[0]), 2, sw, 0., NULL));
        if (sbits) sign = orc_rd_rawbits(c->rd, 1);
        sign = 1 - 2 * sign;
        cm = quant_band(c, x2, n, mbits, B, lowband, lm, lowband_out, 1.f, scratch, orig_fill);
        y2[0] = -sign * x2[1];
        y2[1] = sign * x2[0];
        x[0] *= mid; x[1] *= mid; y[0] *= side; y[1] *= side;
        float t = x[0]; x[0] = t - y[0]; y[0] = t + y[0];
        t = x[1]; x[1] = t - y[1]; y[1] = t + y[1];
    } else {
        int mbits = (b - s.delta) / 2;
        if (mbits > b) mbits = b;
        if (mbits < 0) mbits = 0;
        int sbits = b - mbits;
        c->remaining_bits -= s.qalloc;
        int rebalance = c->remaining_bits;
        if (mbits >= sbits) {
            cm = quant_band(c, x, n, mbits, B, lowband, lm, lowband_out, 1.f, scratch, fill);
            rebalance = mbits - (rebalance - c->remaining_bits);
            if (rebalance > 24 && s.itheta != 0) sbits += rebalance - 24;
            cm |= quant_band(c, y, n, sbits, B, NULL, lm, NULL, side, NULL, fill >> B);
        } else {
            cm = quant_band(c, y, n, sbits, B, NULL, lm, NULL, side, NULL, fill >> B);
            rebalance = sbits - (rebalance - c->remaining_bits);
            if (rebalance > 24 && s.itheta != 16384) mbits += rebalance - 24;
            cm |= quant_band(c, x, n, mbits, B, lowband, lm, lowband_out, 1.f, scratch, fill);
        }
        orc_stereo_merge(x, y, mid, n);
    }
    if (s.inv) for (int j = 0; j < n; j++) y[j] = -y[j];
    return cm;
}

static void quant_all_bands(OrcCelt *st, OrcRd *rd, int start, int end, float *X, float *Y, uint8_t *collapse,
                            const int *pulses, int short_blocks, int spread, int dual_stereo, int intensity,
                            const int *tf_res, int total_bits, int balance, int lm, int coded_bands, OrcCeltTrace *tr)
{   /* celt/decoder.rs:2037-2220 */
    int M = 1 << lm, B = short_blocks ? M : 1, C = Y ? 2 : 1;
    float norm[2][8 * 100], scratch[8 * 22];
    int norm_offset = M * CELT_EBANDS[start];
    int update_lowband = 1, lowband_offset = 0;
    BandCtx c = {rd, 0, intensity, spread, 0, 0, st->rng, {X, Y}, tr};
    memset(norm, 0, sizeof norm);
    for (int i = start; i < end; i++) {
        int last = i == end - 1;
        float *x = X + M * CELT_EBANDS[i], *y = Y ? Y + M * CELT_EBANDS[i] : NULL;
        int n = M * CELT_EBANDS[i + 1] - M * CELT_EBANDS[i];
        int tell = orc_rd_tell_frac(rd), b;
        c.band = i;
        if (i != start) balance -= tell;
        int remaining_bits = total_bits - tell - 1;
        c.remaining_bits = remaining_bits;
        if (i <= coded_bands - 1) {
            int d = coded_bands - i; if (d > 3) d = 3;
            int curr_balance = balance / d;          /* SURVEY note :2068-2070 -> RFC divisor min(3, coded-i) */
            b = pulses[i] + curr_balance;
            if (b > remaining_bits + 1) b = remaining_bits + 1;
            if (b > 16383) b = 16383;
            if (b < 0) b = 0;
        } else b = 0;
        if (M * CELT_EBANDS[i] - n >= M * CELT_EBANDS[start] && (update_lowband || lowband_offset == 0))
            lowband_offset = i;
        c.tf_change = tf_res[i];
        int effective_lowband = -1, x_cm, y_cm;
        if (lowband_offset != 0 && (spread != 3 || B > 1 || c.tf_change < 0)) {
            effective_lowband = M * CELT_EBANDS[lowband_offset] - norm_offset - n;
            if (effective_lowband < 0) effective_lowband = 0;
            int fold_start = lowband_offset, fold_end = lowband_offset - 1;
            while (M * CELT_EBANDS[--fold_start] > effective_lowband + norm_offset) ;
            while (M * CELT_EBANDS[++fold_end] < effective_lowband + norm_offset + n) ;
            x_cm = y_cm = 0;
            for (int f = fold_start; f < fold_end; f++) { x_cm |= collapse[f * C]; y_cm |= collapse[f * C + C - 1]; }
        } else x_cm = y_cm = (1 << B) - 1;
        if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            for (int j = 0; j < M * CELT_EBANDS[i] - norm_offset; j++) norm[0][j] = 0.5f * (norm[0][j] + norm[1][j]);
        }
        float *lb0 = effective_lowband != -1 ? norm[0] + effective_lowband : NULL;
        float *lb1 = effective_lowband != -1 ? norm[1] + effective_lowband : NULL;
        float *out0 = last ? NULL : norm[0] + M * CELT_EBANDS[i] - norm_offset;
        float *out1 = last ? NULL : norm[1] + M * CELT_EBANDS[i] - norm_offset;
        if (tr) { tr->band_b[i] = b; tr->band_lowband[i] = effective_lowband; }
        if (dual_stereo) {
            x_cm = quant_band(&c, x, n, b / 2, B, lb0, lm, out0, 1.f, scratch, x_cm);
            y_cm = quant_band(&c, y, n, b / 2, B, lb1, lm, out1, 1.f, scratch, y_cm);
        } else {
            if (y) x_cm = quant_band_stereo(&c, x, y, n, b, B, lb0, lm, out0, scratch, x_cm | y_cm);
            else x_cm = quant_band(&c, x, n, b, B, lb0, lm, out0, 1.f, scratch, x_cm | y_cm);
            y_cm = x_cm;
        }
        collapse[i * C] = (uint8_t)x_cm;
        collapse[i * C + C - 1] = (uint8_t)y_cm;
        balance += pulses[i] + tell;
        update_lowband = b > (n << 3);
    }
    st->rng = c.seed;
}

/* ------------------------------------------------------------------ synthesis (RFC 6716 4.3.5-4.3.7; no reference code) */
static float g_window[120];
static int g_init = 0;
static void init_tables(void)
{
    if (g_init) return;
    for (int i = 0; i < 120; i++) {
        double s = sin(0.5 * M_PI * (i + 0.5) / 120);
        g_window[i] = (float)sin(0.5 * M_PI * s * s);
    }
    g_init = 1;
}
const float *orc_celt_window(void) { init_tables(); return g_window; }

/* Recursive mixed-radix (any factorisation, here 2^a*3*5) forward DFT, out-of-place, f32 data, twiddles
 * evaluated in double.  O(n * sum of factors); clarity over speed. */
typedef struct { float r, i; } cpx;
static void fft_rec(const cpx *in, cpx *out, int n, int stride, int nfull)
{
    if (n == 1) { out[0] = in[0]; return; }
    int p = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : (n % 5 == 0) ? 5 : n;
    int m = n / p;
    for (int q = 0; q < p; q++) fft_rec(in + q * stride, out + q * m, m, stride * p, nfull);
    cpx t[5 > 4 ? 5 : 4];
    for (int k = 0; k < m; k++) {
        for (int q = 0; q < p; q++) {
            double a = -2.0 * M_PI * (double)(q * k) / (double)n;
            float wr = (float)cos(a), wi = (float)sin(a);
            cpx v = out[q * m + k];
            t[q].r = v.r * wr - v.i * wi;
            t[q].i = v.r * wi + v.i * wr;
        }
        for (int r = 0; r < p; r++) {
            float sr = 0, si = 0;
            for (int q = 0; q < p; q++) {
                double a = -2.0 * M_PI * (double)((q * r) % p) / (double)p;
                float wr = (float)cos(a), wi = (float)sin(a);
                sr += t[q].r * wr - t[q].i * wi;
                si += t[q].r * wi + t[q].i * wr;
            }
            out[r * m + k].r = sr;
            out[r * m + k].i = si;
        }
    }
}

/* the forward DFT above on interleaved (re, im) pairs taken `stride` complex elements apart: lets the tests hold it against
 * the reference's one FFT known-answer vector (celt/imdct15.rs:213-251, the radix-5 butterfly on every third of 15 inputs) */
void orc_fft(const float *in, int stride, int n, float *out)
{
    cpx a[480], b[480];
    if (n < 1 || n > 480) return;
    for (int i = 0; i < n; i++) { a[i].r = in[2 * i * stride]; a[i].i = in[2 * i * stride + 1]; }
    fft_rec(a, b, n, 1, n);
    for (int i = 0; i < n; i++) { out[2 * i] = b[i].r; out[2 * i + 1] = b[i].i; }
}

/* Inverse MDCT of n2 coefficients (stride `stride`) -> n2 raw samples D[0..n2).  Structure of RFC 6716
 * mdct.c clt_mdct_backward: pre-twiddle, n2/2-point complex FFT, post-twiddle, no scaling.  Closed form
 * (tests check it):  D[n] = -sum_m X[m] cos(pi/n2 (m+1/2)(n2-n-1/2)). */
void orc_imdct_raw(const float *in, int stride, int n2, float *D)
{
    int n4 = n2 >> 1, N = 2 * n2;
    cpx z[480], Z[480];
    for (int i = 0; i < n4; i++) {
        double th = 2.0 * M_PI * (i + 0.125) / N;
        float c = (float)cos(th), s = (float)sin(th);
        float x1 = in[stride * 2 * i], x2 = in[stride * (n2 - 1 - 2 * i)];
        z[i].r = x1 * c + x2 * s;      /* (x1 + j x2) e^{-j th} */
        z[i].i = x2 * c - x1 * s;
    }
    fft_rec(z, Z, n4, 1, n4);
    for (int k = 0; k < n4; k++) {
        double th = 2.0 * M_PI * (k + 0.125) / N;
        float c = (float)cos(th), s = (float)sin(th);
        float vr = Z[k].r * c + Z[k].i * s, vi = Z[k].i * c - Z[k].r * s;   /* Z e^{-j th} */
        D[2 * k] = vi;
        D[n2 - 1 - 2 * k] = -vr;
    }
}

/* f64 direct evaluation of the same transform (ground truth for tests) */
void orc_imdct_raw_direct(const float *in, int stride, int n2, double *D)
{
    for (int n = 0; n < n2; n++) {
        double acc = 0;
        for (int m = 0; m < n2; m++) acc += (double)in[m * stride] * cos(M_PI / n2 * (m + 0.5) * (n2 - n - 0.5));
        D[n] = -acc;
    }
}

/* one block: raw IMDCT into out+60, then the TDAC mirror over out[0..120) (RFC 6716 mdct.c) */
static void imdct_block(const float *in, int stride, int n2, float *out)
{
    float D[960];
    const float *w = g_window;
    orc_imdct_raw(in, stride, n2, D);
    memcpy(out + 60, D, n2 * sizeof(float));
    for (int i = 0; i < 60; i++) {
        float x1 = out[119 - i], x2 = out[i];
        out[i] = w[119 - i] * x2 - w[i] * x1;
        out[119 - i] = w[i] * x2 + w[119 - i] * x1;
    }
}

static void comb_filter(float *y, int t0, int t1, int n, float g0, float g1, int tap0, int tap1)
{   /* RFC 6716 celt.c comb_filter, in place (x == y), overlap 120; taps celt/decoder.rs:90-94 */
    if (g0 == 0 && g1 == 0) return;
    if (t0 < 15) t0 = 15;
    if (t1 < 15) t1 = 15;
    float g00 = g0 * CELT_POSTFILTER_TAPS[tap0][0], g01 = g0 * CELT_POSTFILTER_TAPS[tap0][1], g02 = g0 * CELT_POSTFILTER_TAPS[tap0][2];
    float g10 = g1 * CELT_POSTFILTER_TAPS[tap1][0], g11 = g1 * CELT_POSTFILTER_TAPS[tap1][1], g12 = g1 * CELT_POSTFILTER_TAPS[tap1][2];
    int overlap = (g0 == g1 && t0 == t1 && tap0 == tap1) ? 0 : 120;
    int i;
    for (i = 0; i < overlap; i++) {
        float f = g_window[i] * g_window[i];
        y[i] = y[i] + ((1.f - f) * g00) * y[i - t0]
             + ((1.f - f) * g01) * (y[i - t0 + 1] + y[i - t0 - 1])
             + ((1.f - f) * g02) * (y[i - t0 + 2] + y[i - t0 - 2])
             + (f * g10) * y[i - t1]
             + (f * g11) * (y[i - t1 + 1] + y[i - t1 - 1])
             + (f * g12) * (y[i - t1 + 2] + y[i - t1 - 2]);
    }
    if (g1 == 0) return;
    for (; i < n; i++)
        y[i] = y[i] + g10 * y[i - t1] + g11 * (y[i - t1 + 1] + y[i - t1 - 1]) + g12 * (y[i - t1 + 2] + y[i - t1 - 2]);
}

static void anti_collapse(float *X[2], const uint8_t *collapse, int lm, int C, int start, int end,
                          float logE[2][21], float prev1[2][21], float prev2[2][21], const int *pulses, uint32_t seed)
{   /* RFC 6716 bands.c anti_collapse; the reference only reserves the bit (celt/decoder.rs:1092) */
    for (int i = start; i < end; i++) {
        int n0 = CELT_BAND_WIDTH[i];
        int depth = ((1 + pulses[i]) / n0) >> lm;
        float thresh = 0.5f * exp2f(-0.125f * depth);
        float sqrt_1 = 1.f / sqrtf((float)(n0 << lm));
        for (int c = 0; c < C; c++) {
            float p1 = prev1[c][i], p2 = prev2[c][i];
            if (C == 1) { if (prev1[1][i] > p1) p1 = prev1[1][i]; if (prev2[1][i] > p2) p2 = prev2[1][i]; }
            float ediff = logE[c][i] - (p1 < p2 ? p1 : p2);
            if (ediff < 0) ediff = 0;
            float r = 2.f * exp2f(-ediff);
            if (lm == 3) r *= 1.41421356f;
            if (r > thresh) r = thresh;
            r *= sqrt_1;
            float *x = X[c] + (CELT_EBANDS[i] << lm);
            int renorm = 0;
            for (int k = 0; k < (1 << lm); k++)
                if (!(collapse[i * C + c] & (1 << k))) {
                    for (int j = 0; j < n0; j++) { seed = lcg(seed); x[(j << lm) + k] = (seed & 0x8000) ? r : -r; }
                    renorm = 1;
                }
            if (renorm) renormalize_vector(x, n0 << lm, 1.f);
        }
    }
}

static void denormalise(const float *X, float *freq, const float *logE, int start, int end, int M, int silence)
{
    int n = M * 120;
    memset(freq, 0, n * sizeof(float));
    if (silence) return;
    for (int i = start; i < end; i++) {
        float lg = logE[i] + EMEANS[i];
        float g = exp2f(lg < 32.f ? lg : 32.f);
        for (int j = M * CELT_EBANDS[i]; j < M * CELT_EBANDS[i + 1]; j++) freq[j] = X[j] * g;
    }
}

OrcCelt *orc_celt_create(int channels)
{
    OrcCelt *st = (OrcCelt *)calloc(1, sizeof(OrcCelt));
    st->channels = channels;
    orc_celt_reset(st);
    return st;
}
void orc_celt_destroy(OrcCelt *st) { free(st); }
void orc_celt_reset(OrcCelt *st)
{
    int ch = st->channels; OrcCeltTrace *tr = st->trace;
    memset(st, 0, sizeof *st);
    st->channels = ch; st->trace = tr;
    for (int c = 0; c < 2; c++) for (int i = 0; i < 21; i++) st->logE1[c][i] = st->logE2[c][i] = -28.f;
    init_tables();
}
void orc_celt_set_trace(OrcCelt *st, OrcCeltTrace *tr) { st->trace = tr; }

/* Decode one CELT frame of `N` samples/channel from rd (celt/decoder.rs:2222-2323 + RFC 6716 4.3 tail).
 * C = channels coded in the packet; output: st->channels planar arrays of N floats, full-scale +-1. */
static int celt_frame(OrcCelt *st, OrcRd *rd, int N, int start, int end, int C, float *pcm0, float *pcm1, float decay);
int orc_celt_decode_frame(OrcCelt *st, OrcRd *rd, int N, int start, int end, int C, float *pcm0, float *pcm1)
{
    return celt_frame(st, rd, N, start, end, C, pcm0, pcm1, -1.f);
}
/* Concealment of a lost frame (no counterpart in the reference; the noise branch of libopus' celt_decode_lost): the frame is
 * decoded by the regular path from an empty payload, except that it is not taken for silence and that the band energies are the
 * last frame's lowered by `decay` (log2 units) instead of being read.  With no bits every band gets no pulses, so the spectrum
 * is the decoder's own noise filling -- LCG noise in the lowest bands, folded copies above -- shaped by the decaying energies.
 * The noise seed moves on by one LCG step per frame, so consecutive frames are uncorrelated. */
int orc_celt_conceal_frame(OrcCelt *st, int N, int start, int end, int C, float decay, float *pcm0, float *pcm1)
{
    OrcRd rd;
    orc_rd_init(&rd, (const uint8_t *)st, 0);                 /* an empty payload: no byte of it is read */
    return celt_frame(st, &rd, N, start, end, C, pcm0, pcm1, decay);
}
static int celt_frame(OrcCelt *st, OrcRd *rd, int N, int start, int end, int C, float *pcm0, float *pcm1, float decay)
{
    int lm = orc_ilog(N / 120) - 1, M = 1 << lm, CC = st->channels, conceal = decay >= 0.f;
    int total_bits = rd->size_bits, tell = orc_rd_tell(rd);
    uint32_t seed_in = st->rng;
    OrcCeltTrace *tr = st->trace;
    float Xbuf[2][960];
    uint8_t collapse[42];
    int tf_res[21], cap[21], offsets[21], pulses[21], fine_quant[21], fine_priority[21];
    memset(Xbuf, 0, sizeof Xbuf);
    memset(collapse, 0, sizeof collapse);
    if (tr) {
        memset(tr, 0, sizeof *tr);
        tr->seed_in = st->rng; tr->lm = lm; tr->start = start; tr->end = end; tr->C = C;
    }
    if (C == 1) for (int i = 0; i < 21; i++) if (st->oldE[1][i] > st->oldE[0][i]) st->oldE[0][i] = st->oldE[1][i]; /* :2266 */

    int silence = 0;
    if (conceal) silence = 0;
    else if (tell >= total_bits) silence = 1;
    else if (tell == 1) silence = orc_rd_logp(rd, 15);       /* celt/decoder.rs:2237 (RFC: only when tell==1) */
    if (silence) { orc_rd_to_end(rd); tell = total_bits; }

    int pf_pitch = 0, pf_tapset = 0, pf_on = 0, pf_qg = 0; float pf_gain = 0;
    if (start == 0 && tell + 16 <= total_bits) {              /* celt/decoder.rs:876-901 */
        int f_oct = enc_pf_period > 0 ? orc_ilog((uint32_t)(enc_pf_period + 1)) - 5 : 0;      /* period + 1 = (16 << octave) + bits */
        if (orc_force_on) orc_force(enc_pf_period > 0);
        if (orc_rd_logp(rd, 1)) {
            if (orc_force_on) orc_force(f_oct);
            int octave = orc_rd_uniform(rd, 6);
            if (orc_force_on) orc_force(enc_pf_period + 1 - (16 << f_oct));
            pf_pitch = (16 << octave) + orc_rd_rawbits(rd, 4 + octave) - 1;
            if (orc_force_on) orc_force(enc_pf_gain);
            pf_qg = orc_rd_rawbits(rd, 3);
            if (orc_force_on && orc_rd_tell(rd) + 2 <= total_bits) orc_force(enc_pf_tapset);
            if (orc_rd_tell(rd) + 2 <= total_bits) pf_tapset = orc_rd_icdf(rd, ICDF_CELT_TAPSET);
            pf_gain = 0.09375f * (pf_qg + 1);
            pf_on = 1;
        }
        tell = orc_rd_tell(rd);
    }
    int transient = 0;
    if (orc_force_on && lm > 0 && tell + 3 <= total_bits) orc_force(enc_transient);
    if (lm > 0 && tell + 3 <= total_bits) { transient = orc_rd_logp(rd, 3); tell = orc_rd_tell(rd); }
    int short_blocks = transient ? M : 0;
    int intra = tell + 3 <= total_bits ? orc_rd_logp(rd, 3) : 0;

    /* coarse energy: celt/decoder.rs:903-992 */
    {
        const uint8_t *model = intra ? CELT_E_PROB_INTRA[lm] : CELT_E_PROB_INTER[lm];
        float alpha = intra ? 0.f : ALPHA_COEF[lm], beta = intra ? 1.f - 4915.f / 32768.f : BETA_COEF[lm];
        float prev[2] = {0, 0};
        for (int i = 0; i < 21; i++)
            for (int c = 0; c < C; c++) {
                float *en = &st->oldE[c][i];
                if (i < start || i >= end) { *en = 0.f; continue; }
                if (conceal) { *en = *en - decay > -28.f ? *en - decay : -28.f; continue; }
                int avail = total_bits - orc_rd_tell(rd), qi;
                if (orc_force_on && avail >= 1) {
                    int cls = avail >= 15 ? 0 : avail >= 2 ? 1 : 2;
                    int64_t want = enc_hook(1, i, c, cls, 0, 0, (double)((*en > -9.f ? *en : -9.f) * alpha + prev[c]), NULL);
                    orc_force(cls == 0 ? want : cls == 1 ? (want == 0 ? 0 : want < 0 ? 1 : 2) : want < 0);
                }
                if (avail >= 15) { int k = (i < 20 ? i : 20) << 1; qi = orc_rd_laplace(rd, model[k] << 7, model[k + 1] << 6); }
                else if (avail >= 2) { qi = orc_rd_icdf(rd, ICDF_CELT_ENERGY_SMALL); qi = (qi >> 1) ^ -(qi & 1); }
                else if (avail >= 1) qi = -orc_rd_logp(rd, 1);
                else qi = -1;
                float q = (float)qi;
                *en = (*en > -9.f ? *en : -9.f) * alpha + prev[c] + q;
                prev[c] += beta * q;
            }
    }
    /* tf: celt/decoder.rs:994-1033 (RFC budget test) */
    {
        int budget = total_bits, logp = transient ? 2 : 4, curr = 0, changed = 0;
        tell = orc_rd_tell(rd);
        int rsv = lm > 0 && tell + logp + 1 <= budget;
        budget -= rsv;
        for (int i = start; i < end; i++) {
            if (orc_force_on && (enc_transient || enc_tf) && tell + logp <= budget) orc_force(i == start);
            if (tell + logp <= budget) { curr ^= orc_rd_logp(rd, logp); tell = orc_rd_tell(rd); changed |= curr; }
            tf_res[i] = curr;
            logp = transient ? 4 : 5;
        }
        int sel = 0;
        if (orc_force_on && rsv && CELT_TF_SELECT[lm][transient][0][changed] != CELT_TF_SELECT[lm][transient][1][changed]) orc_force(enc_tf == 2);
        if (rsv && CELT_TF_SELECT[lm][transient][0][changed] != CELT_TF_SELECT[lm][transient][1][changed])
            sel = orc_rd_logp(rd, 1);
        for (int i = start; i < end; i++) tf_res[i] = CELT_TF_SELECT[lm][transient][sel][tf_res[i]];
    }
    /* allocation: celt/decoder.rs:1035-1447, arithmetic per RFC 6716 rate.c where the two differ */
    tell = orc_rd_tell(rd);
    int spread = 2;
    if (orc_force_on && tell + 4 <= total_bits) orc_force(enc_spread);
    if (tell + 4 <= total_bits) spread = orc_rd_icdf(rd, ICDF_CELT_SPREAD);   /* :1043 (kept, not discarded) */
    for (int i = 0; i < 21; i++) cap[i] = (CELT_CAPS[lm][C - 1][i] + 64) * C * (CELT_BAND_WIDTH[i] << lm) >> 2;
    int dynalloc_logp = 6, tbits = total_bits << 3;
    int tellf = orc_rd_tell_frac(rd);
    for (int i = 0; i < 21; i++) offsets[i] = 0;
    for (int i = start; i < end; i++) {
        int width = C * CELT_BAND_WIDTH[i] << lm;
        int quanta = width << 3 < (48 > width ? 48 : width) ? width << 3 : (48 > width ? 48 : width);
        int loop_logp = dynalloc_logp, boost = 0;
        while (tellf + (loop_logp << 3) < tbits && boost < cap[i]) {
            int flag = orc_rd_logp(rd, loop_logp);
            tellf = orc_rd_tell_frac(rd);
            if (!flag) break;
            boost += quanta; tbits -= quanta; loop_logp = 1;
        }
        offsets[i] = boost;
        if (boost > 0 && dynalloc_logp > 2) dynalloc_logp--;
    }
    if (orc_force_on && tellf + 48 <= tbits) orc_force(5);
    int alloc_trim = tellf + 48 <= tbits ? orc_rd_icdf(rd, ICDF_CELT_ALLOC_TRIM) : 5;
    int bits = (total_bits << 3) - orc_rd_tell_frac(rd) - 1;
    int anti_rsv = transient && lm >= 2 && bits >= (lm + 2) << 3 ? 8 : 0;
    bits -= anti_rsv;
    int intensity = 0, dual_stereo = 0, balance = 0, coded_bands;
    {
        int total = bits > 0 ? bits : 0, skip_start = start;
        int skip_rsv = total >= 8 ? 8 : 0;
        total -= skip_rsv;
        int intensity_rsv = 0, dual_rsv = 0;
        if (C == 2) {
            intensity_rsv = CELT_LOG2_FRAC[end - start];
            if (intensity_rsv > total) intensity_rsv = 0;
            else { total -= intensity_rsv; dual_rsv = total >= 8 ? 8 : 0; total -= dual_rsv; }
        }
        int thresh[21], trim_offset[21], bits1[21], bits2[21];
        for (int j = start; j < end; j++) {
            int w = CELT_BAND_WIDTH[j];
            thresh[j] = (C << 3) > ((3 * w << lm << 3) >> 4) ? (C << 3) : ((3 * w << lm << 3) >> 4);
            trim_offset[j] = C * w * (alloc_trim - 5 - lm) * (end - j - 1) * (1 << (lm + 3)) >> 6;
            if (w << lm == 1) trim_offset[j] -= C << 3;
        }
        int lo = 1, hi = 10;
        do {
            int done = 0, psum = 0, mid = (lo + hi) >> 1;
            for (int j = end; j-- > start;) {
                int bj = C * CELT_BAND_WIDTH[j] * CELT_ALLOC_VECTORS[mid][j] << lm >> 2;
                if (bj > 0) { bj += trim_offset[j]; if (bj < 0) bj = 0; }
                bj += offsets[j];
                if (bj >= thresh[j] || done) { done = 1; psum += bj < cap[j] ? bj : cap[j]; }
                else if (bj >= C << 3) psum += C << 3;
            }
            if (psum > total) hi = mid - 1; else lo = mid + 1;
        } while (lo <= hi);
        hi = lo--;
        for (int j = start; j < end; j++) {
            int w = CELT_BAND_WIDTH[j];
            int b1 = C * w * CELT_ALLOC_VECTORS[lo][j] << lm >> 2;
            int b2 = hi >= 11 ? cap[j] : C * w * CELT_ALLOC_VECTORS[hi][j] << lm >> 2;
            if (b1 > 0) { b1 += trim_offset[j]; if (b1 < 0) b1 = 0; }
            if (b2 > 0) { b2 += trim_offset[j]; if (b2 < 0) b2 = 0; }
            if (lo > 0) b1 += offsets[j];
            b2 += offsets[j];
            if (offsets[j] > 0) skip_start = j;
            b2 = b2 - b1 > 0 ? b2 - b1 : 0;
            bits1[j] = b1; bits2[j] = b2;
        }
        /* interp_bits2pulses */
        int alloc_floor = C << 3, stereo = C > 1, logM = lm << 3, psum;
        lo = 0; hi = 64;
        for (int it = 0; it < 6; it++) {
            int mid = (lo + hi) >> 1, done = 0;
            psum = 0;
            for (int j = end; j-- > start;) {
                int t = bits1[j] + (mid * bits2[j] >> 6);
                if (t >= thresh[j] || done) { done = 1; psum += t < cap[j] ? t : cap[j]; }
                else if (t >= alloc_floor) psum += alloc_floor;
            }
            if (psum > total) hi = mid; else lo = mid;
        }
        psum = 0;
        {
            int done = 0;
            for (int j = end; j-- > start;) {
                int t = bits1[j] + (lo * bits2[j] >> 6);
                if (t < thresh[j] && !done) t = t >= alloc_floor ? alloc_floor : 0;
                else done = 1;
                if (t > cap[j]) t = cap[j];
                pulses[j] = t;
                psum += t;
            }
        }
        for (coded_bands = end;; coded_bands--) {
            int j = coded_bands - 1;
            if (j <= skip_start) { total += skip_rsv; break; }
            int left = total - psum;
            int span = CELT_EBANDS[coded_bands] - CELT_EBANDS[start];
            int percoeff = left / span;
            left -= span * percoeff;
            int rem = left - (CELT_EBANDS[j] - CELT_EBANDS[start]);
            if (rem < 0) rem = 0;
            int band_width = CELT_EBANDS[coded_bands] - CELT_EBANDS[j];
            int band_bits = pulses[j] + percoeff * band_width + rem;
            if (band_bits >= (thresh[j] > alloc_floor + 8 ? thresh[j] : alloc_floor + 8)) {
                if (orc_force_on) orc_force(1);
                if (orc_rd_logp(rd, 1)) break;
                psum += 8;
                band_bits -= 8;
            }
            psum -= pulses[j] + intensity_rsv;
            if (intensity_rsv > 0) intensity_rsv = CELT_LOG2_FRAC[j - start];
            psum += intensity_rsv;
            if (band_bits >= alloc_floor) { psum += alloc_floor; pulses[j] = alloc_floor; }
            else pulses[j] = 0;
        }
        if (orc_force_on && intensity_rsv > 0) orc_force(coded_bands - start);      /* (encoder: no intensity stereo ...) */
        if (intensity_rsv > 0) intensity = start + orc_rd_uniform(rd, coded_bands + 1 - start);
        if (intensity <= start) { total += dual_rsv; dual_rsv = 0; }
        if (orc_force_on && dual_rsv > 0) orc_force(enc_dual);                      /* (... the channels coded one by one, or jointly) */
        if (dual_rsv > 0) dual_stereo = orc_rd_logp(rd, 1);
        int left = total - psum;
        int span = CELT_EBANDS[coded_bands] - CELT_EBANDS[start];
        int percoeff = left / span;
        left -= span * percoeff;
        for (int j = start; j < coded_bands; j++) pulses[j] += percoeff * CELT_BAND_WIDTH[j];
        for (int j = start; j < coded_bands; j++) {
            int t = left < CELT_BAND_WIDTH[j] ? left : CELT_BAND_WIDTH[j];
            pulses[j] += t;
            left -= t;
        }
        int j;
        for (j = start; j < coded_bands; j++) {
            int n0 = CELT_BAND_WIDTH[j], n = n0 << lm, bit = pulses[j] + balance, excess;
            if (n > 1) {
                excess = bit - cap[j] > 0 ? bit - cap[j] : 0;
                pulses[j] = bit - excess;
                int den = C * n + ((C == 2 && n > 2 && !dual_stereo && j < intensity) ? 1 : 0);
                int nclogn = den * (CELT_LOG_N[j] + logM);
                int offset = (nclogn >> 1) - den * 21;
                if (n == 2) offset += den << 3 >> 2;
                if (pulses[j] + offset < den * 2 << 3) offset += nclogn >> 2;
                else if (pulses[j] + offset < den * 3 << 3) offset += nclogn >> 3;
                int eb = pulses[j] + offset + (den << 2);
                if (eb < 0) eb = 0;
                eb = (eb / den) >> 3;
                if (C * eb > (pulses[j] >> 3)) eb = pulses[j] >> stereo >> 3;
                if (eb > 8) eb = 8;
                fine_quant[j] = eb;
                fine_priority[j] = eb * (den << 3) >= pulses[j] + offset;
                pulses[j] -= C * eb << 3;
            } else {
                excess = bit - (C << 3) > 0 ? bit - (C << 3) : 0;
                pulses[j] = bit - excess;
                fine_quant[j] = 0;
                fine_priority[j] = 1;
            }
            if (excess > 0) {
                int extra_fine = excess >> (stereo + 3);
                if (extra_fine > 8 - fine_quant[j]) extra_fine = 8 - fine_quant[j];
                fine_quant[j] += extra_fine;
                int extra_bits = extra_fine * C << 3;
                fine_priority[j] = extra_bits >= excess - balance;
                excess -= extra_bits;
            }
            balance = excess;
        }
        for (; j < end; j++) {
            fine_quant[j] = pulses[j] >> stereo >> 3;
            pulses[j] = 0;
            fine_priority[j] = fine_quant[j] < 1;
        }
    }
    /* fine energy: celt/decoder.rs:1449-1464 */
    for (int i = start; i < end; i++) {
        if (fine_quant[i] <= 0) continue;
        for (int c = 0; c < C; c++) {
            if (orc_force_on) orc_force(enc_hook(2, i, c, fine_quant[i], 0, 0, (double)st->oldE[c][i], NULL));
            int q2 = orc_rd_rawbits(rd, fine_quant[i]);
            st->oldE[c][i] += ((float)q2 + 0.5f) * (float)(1 << (14 - fine_quant[i])) * (1.f / 16384) - 0.5f;
        }
    }
    if (tr) {
        tr->silence = silence; tr->transient = transient; tr->intra = intra; tr->spread = spread;
        tr->pf_on = pf_on; tr->pf_period = pf_pitch; tr->pf_gain_q = pf_qg; tr->pf_tapset = pf_tapset;
        tr->alloc_trim = alloc_trim; tr->intensity = intensity; tr->dual_stereo = dual_stereo;
        tr->codedbands = coded_bands; tr->anticollapse_rsv = anti_rsv;
        for (int i = start; i < end; i++) {
            tr->tf_change[i] = tf_res[i]; tr->pulses_alloc[i] = pulses[i]; tr->fine_bits[i] = fine_quant[i];
            tr->fine_prio[i] = fine_priority[i]; tr->boost[i] = offsets[i];
        }
    }
    float *X[2] = {Xbuf[0], Xbuf[1]};
    quant_all_bands(st, rd, start, end, X[0], C == 2 ? X[1] : NULL, collapse, pulses, short_blocks, spread, dual_stereo,
                    intensity, tf_res, (total_bits << 3) - anti_rsv, balance, lm, coded_bands, tr);
    int anti_on = 0;
    if (anti_rsv > 0) anti_on = orc_rd_rawbits(rd, 1);
    {   /* energy finalise (RFC 6716 quant_bands.c unquant_energy_finalise) */
        int bits_left = total_bits - orc_rd_tell(rd);
        for (int prio = 0; prio < 2; prio++)
            for (int i = start; i < end && bits_left >= C; i++) {
                if (fine_quant[i] >= 8 || fine_priority[i] != prio) continue;
                for (int c = 0; c < C; c++) {
                    if (orc_force_on) orc_force(enc_hook(3, i, c, fine_quant[i], 0, 0, (double)st->oldE[c][i], NULL));
                    int q2 = orc_rd_rawbits(rd, 1);
                    st->oldE[c][i] += ((float)q2 - 0.5f) * (float)(1 << (14 - fine_quant[i] - 1)) * (1.f / 16384);
                    bits_left--;
                }
            }
    }
    if (tr) {
        tr->anticollapse_on = anti_on;
        for (int c = 0; c < C; c++) {
            memcpy(tr->X[c], Xbuf[c], sizeof Xbuf[c]);
            for (int i = 0; i < 21; i++) { tr->energy[c][i] = st->oldE[c][i]; tr->collapse[c][i] = collapse[i * C + c]; }
        }
        tr->seed_bands = st->rng;
    }
    if (anti_on) anti_collapse(X, collapse, lm, C, start, end, st->oldE, st->logE1, st->logE2, pulses, st->rng);
    if (silence) for (int c = 0; c < C; c++) for (int i = 0; i < 21; i++) st->oldE[c][i] = -28.f;

    /* synthesis */
    float freq[2][960];
    int B = transient ? M : 1, NB = transient ? 120 : N;
    if (CC == 2 && C == 1) {
        denormalise(X[0], freq[0], st->oldE[0], start, end, M, silence);
        memcpy(freq[1], freq[0], sizeof freq[0]);
    } else if (CC == 1 && C == 2) {
        denormalise(X[0], freq[0], st->oldE[0], start, end, M, silence);
        denormalise(X[1], freq[1], st->oldE[1], start, end, M, silence);
        for (int i = 0; i < N; i++) freq[0][i] = 0.5f * freq[0][i] + 0.5f * freq[1][i];
    } else {
        for (int c = 0; c < CC; c++) denormalise(X[c], freq[c], st->oldE[c], start, end, M, silence);
    }
    if (tr) for (int c = 0; c < CC; c++) memcpy(tr->freq[c], freq[c], sizeof freq[c]);
    float *pcm[2] = {pcm0, pcm1};
    for (int c = 0; c < CC; c++) {
        float *out = st->mem[c] + ORC_CELT_HIST;
        for (int b = 0; b < B; b++) imdct_block(&freq[c][b], B, NB, out + NB * b);
        comb_filter(out, st->pf_period_old, st->pf_period, 120, st->pf_gain_old, st->pf_gain, st->pf_tapset_old, st->pf_tapset);
        if (lm != 0)
            comb_filter(out + 120, st->pf_period, pf_pitch, N - 120, st->pf_gain, pf_gain, st->pf_tapset, pf_tapset);
        float m = st->deemph[c];
        for (int j = 0; j < N; j++) {
            float t = out[j] + m;
            m = 0.85000610f * t;
            pcm[c][j] = t * (1.f / 32768.f);
        }
        st->deemph[c] = m;
        memmove(st->mem[c], st->mem[c] + N, (ORC_CELT_HIST + 60) * sizeof(float));
    }
    st->pf_period_old = st->pf_period; st->pf_gain_old = st->pf_gain; st->pf_tapset_old = st->pf_tapset;
    st->pf_period = pf_pitch; st->pf_gain = pf_gain; st->pf_tapset = pf_tapset;
    if (lm != 0) { st->pf_period_old = st->pf_period; st->pf_gain_old = st->pf_gain; st->pf_tapset_old = st->pf_tapset; }

    if (C == 1) memcpy(st->oldE[1], st->oldE[0], sizeof st->oldE[0]);
    if (!transient) {
        memcpy(st->logE2, st->logE1, sizeof st->logE1);
        memcpy(st->logE1, st->oldE, sizeof st->oldE);
    } else {
        for (int c = 0; c < 2; c++) for (int i = 0; i < 21; i++) if (st->oldE[c][i] < st->logE1[c][i]) st->logE1[c][i] = st->oldE[c][i];
    }
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 21; i++)
            if (i < start || i >= end) { st->oldE[c][i] = 0; st->logE1[c][i] = st->logE2[c][i] = -28.f; }
    if (conceal) {
        st->rng = seed_in * 1664525u + 1013904223u;              /* (of the frame's first seed: the draws above are not counted) */
        if (tr) tr->seed_out = st->rng;
        return 0;
    }
    st->rng = rd->range;                                      /* celt/decoder.rs:2322 */
    if (tr) { tr->seed_out = st->rng; tr->tell_end = orc_rd_tell(rd); }
    return orc_rd_tell(rd) > total_bits ? -1 : 0;
}
