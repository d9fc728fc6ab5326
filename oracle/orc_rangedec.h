/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under opus_amd/ may include, link or call this.
 *
 * CPU restatement of the reference's range decoder (lu-zero/opus src/entropy.rs:260-543) in plain C.
 * Pinned by the reference's own KATs (src/entropy.rs:82-133 reverse_bitread, :135-257
 * decode_laplace) -> tests/test_oracle_kats.py.
 *
 * Divergences from the reference, all on paths its tests do not cover ("parity unpinned"), taken
 * from RFC 6716 section 4.1 instead:
 *   - decode_step  (entropy.rs:487-512): the reference maps the tail symbol with (k0+1)/2 and uses
 *     3*(k+1) as the tail's low bound; RFC: x = x0+1+(fs-3*(x0+1)), fl = (x-1-x0)+3*(x0+1).
 *   - decode_uniform (entropy.rs:427-445): result clamped to len-1 (RFC 4.1.5, ec_dec_uint).
 */
#ifndef ORC_RANGEDEC_H
#define ORC_RANGEDEC_H
#include <stdint.h>
#include <string.h>
#include "rfc6716_tables.h"

typedef struct {
    const uint8_t *buf;
    int len;
    uint32_t range, value;   /* entropy.rs:267-268; value is the 31-bit inverted code */
    int total;               /* entropy.rs:269: bits consumed incl. the 9 start-up bits */
    int fpos;                /* forward reader position, in bits (MSB first, zero padded) */
    uint64_t ewin;           /* reverse raw-bit reader: window, bits held, bytes taken from the end */
    int ebits, epos;
    int size_bits;
} OrcRd;

/* Optional symbol trace (tests/test_lbrr.py re-encodes decoded symbols into new packets): when orc_trace_buf is set, every
 * decoded symbol is logged as its (low, high, total) triple; (0, 0, tag) entries are markers. */
extern uint32_t *orc_trace_buf;
extern int orc_trace_cap, orc_trace_n;
static inline void orc_trace3(uint32_t a, uint32_t b, uint32_t c)
{
    if (orc_trace_buf && orc_trace_n < orc_trace_cap) {
        uint32_t *e = orc_trace_buf + 3 * orc_trace_n++;
        e[0] = a; e[1] = b; e[2] = c;
    }
}

/* Forced mode (tests/test_loopback.py: the decoder run as an ENCODER).  With orc_force_on set the primitives below do not look at
 * the payload: each takes the symbol queued with orc_force() (0 when none is queued), updates the range and the bit count exactly
 * as if it had decoded that symbol, and logs it to the trace -- raw bits as (0xFFFFFFFF, value, count) -- from which the test
 * range-ENcodes a packet.  The decoder thereby makes every decision of the bit allocation itself; the test only chooses values. */
extern int orc_force_on, orc_force_have;
extern int64_t orc_force_val;
static inline void orc_force(int64_t v) { orc_force_val = v; orc_force_have = 1; }
static inline int orc_forced(int64_t *v)
{
    if (!orc_force_on) return 0;
    *v = orc_force_have ? orc_force_val : 0;
    orc_force_have = 0;
    return 1;
}

static inline int orc_ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }  /* maths.rs:8-18 */

static inline uint32_t orc_fwd_byte(OrcRd *rd)   /* entropy.rs:38-56 (8-bit get at any bit offset) */
{
    int i = rd->fpos >> 3, s = rd->fpos & 7;
    uint32_t b0 = i < rd->len ? rd->buf[i] : 0, b1 = i + 1 < rd->len ? rd->buf[i + 1] : 0;
    rd->fpos += 8;
    return (((b0 << 8) | b1) >> (8 - s)) & 0xFF;
}

static inline void orc_rd_normalize(OrcRd *rd)   /* entropy.rs:290-299 */
{
    while (rd->range <= (1u << 23)) {
        uint32_t v = orc_fwd_byte(rd) ^ 0xFF;
        rd->value = ((rd->value << 8) | v) & 0x7FFFFFFFu;
        rd->range <<= 8;
        rd->total += 8;
    }
}

static inline void orc_rd_init(OrcRd *rd, const uint8_t *buf, int len)   /* entropy.rs:301-316 */
{
    memset(rd, 0, sizeof *rd);
    rd->buf = buf; rd->len = len; rd->size_bits = len * 8;
    uint32_t b0 = len > 0 ? buf[0] : 0;
    rd->value = 127 - (b0 >> 1);
    rd->fpos = 7;
    rd->range = 128;
    rd->total = 9;
    orc_rd_normalize(rd);
}

static inline int orc_rd_tell(const OrcRd *rd) { return rd->total - orc_ilog(rd->range); }  /* :375 */

static inline int orc_rd_tell_frac(const OrcRd *rd)   /* entropy.rs:380-392 */
{
    int lg = orc_ilog(rd->range);
    uint32_t rq15 = rd->range >> (lg - 16);
    for (int i = 0; i < 3; i++) {
        rq15 = (rq15 * rq15) >> 15;
        int lastbit = rq15 >> 16;
        lg = lg * 2 | lastbit;
        rq15 >>= lastbit;
    }
    return rd->total * 8 - lg;
}

static inline int orc_rd_avail(const OrcRd *rd) { return rd->size_bits - orc_rd_tell(rd); }

static inline void orc_rd_update(OrcRd *rd, uint32_t scale, uint32_t low, uint32_t high, uint32_t total)
{   /* entropy.rs:318-330 */
    orc_trace3(low, high, total);
    uint32_t s = scale * (total - high);
    rd->value -= s;
    rd->range = low ? scale * (high - low) : rd->range - s;
    orc_rd_normalize(rd);
}

static inline uint32_t orc_rd_symbol(OrcRd *rd, uint32_t total, uint32_t *scale)   /* :332-337 */
{
    *scale = rd->range / total;
    uint32_t k = rd->value / *scale + 1;
    return total - (k < total ? k : total);
}

static inline int orc_rd_logp(OrcRd *rd, int logp)   /* entropy.rs:339-355 */
{
    uint32_t scale = rd->range >> logp;
    int k;
    int64_t fv;
    if (orc_forced(&fv)) { k = fv != 0; if (k) rd->range = scale; else rd->range -= scale; }
    else if (scale > rd->value) { rd->range = scale; k = 1; }
    else { rd->range -= scale; rd->value -= scale; k = 0; }
    orc_trace3(k ? (1u << logp) - 1 : 0, k ? 1u << logp : (1u << logp) - 1, 1u << logp);
    orc_rd_normalize(rd);
    return k;
}

/* ICDF by pool offset (see rfc6716_tables.h). entropy.rs:357-372 */
static inline int orc_rd_icdf_tab(OrcRd *rd, const uint16_t *t);
static inline int orc_rd_icdf(OrcRd *rd, int off) { return orc_rd_icdf_tab(rd, RFC_ICDF_POOL + off); }
/* the same on a table outside the pool: {total, count, cumulative upper bounds ...} */
static inline int orc_rd_icdf_tab(OrcRd *rd, const uint16_t *t)
{
    uint32_t total = t[0], scale;
    uint32_t sym = orc_rd_symbol(rd, total, &scale);
    int k = 0;
    int64_t fv;
    if (orc_forced(&fv)) k = (int)fv < (int)t[1] ? (int)fv : (int)t[1] - 1;
    else while (t[2 + k] <= sym) k++;
    orc_rd_update(rd, scale, k ? t[1 + k] : 0, t[2 + k], total);
    return k;
}

static inline uint32_t orc_rd_rawbits(OrcRd *rd, int n)   /* entropy.rs:5-19, 422-425 */
{
    if (n == 0) return 0;
    int64_t fv;
    if (orc_forced(&fv)) {
        uint32_t fb = (uint32_t)fv & (uint32_t)((1ull << n) - 1);
        rd->total += n;
        orc_trace3(0xFFFFFFFFu, fb, (uint32_t)n);
        return fb;
    }
    while (rd->ebits < n) {
        uint64_t b = rd->epos < rd->len ? rd->buf[rd->len - 1 - rd->epos] : 0;
        rd->ewin |= b << rd->ebits;
        rd->ebits += 8; rd->epos++;
    }
    uint32_t v = (uint32_t)(rd->ewin & ((1ull << n) - 1));
    rd->ewin >>= n; rd->ebits -= n;
    rd->total += n;
    return v;
}

static inline uint32_t orc_rd_uniform(OrcRd *rd, uint32_t len)   /* entropy.rs:427-445 */
{
    int bits = orc_ilog(len - 1);
    uint32_t total = bits > 8 ? ((len - 1) >> (bits - 8)) + 1 : len, scale;
    uint32_t k = orc_rd_symbol(rd, total, &scale);
    int64_t fv;
    if (orc_forced(&fv)) {
        uint32_t want = (uint64_t)fv < len ? (uint32_t)fv : len - 1;
        k = bits > 8 ? want >> (bits - 8) : want;
        if (bits > 8) orc_force(want & ((1u << (bits - 8)) - 1));       /* the low bits follow as raw bits */
    }
    orc_rd_update(rd, scale, k, k + 1, total);
    if (bits > 8) {
        k = (k << (bits - 8)) | orc_rd_rawbits(rd, bits - 8);
        if (k > len - 1) k = len - 1;
    }
    return k;
}

static inline int orc_rd_laplace(OrcRd *rd, uint32_t symbol, int decay)   /* entropy.rs:448-485 */
{
    uint32_t scale = rd->range >> 15;
    uint32_t center = rd->value / scale + 1;
    center = (1u << 15) - (center < (1u << 15) ? center : (1u << 15));
    int value = 0;
    uint32_t low = 0;
    int64_t fv;
    if (orc_forced(&fv)) {                  /* the intervals the loop below walks, up to the wanted value (RFC 6716 laplace.c) */
        int want = (int)fv, mag = want < 0 ? -want : want;
        if (mag) {
            value = 1;
            low = symbol;
            symbol = 1 + ((32768 - 32 - symbol) * (16384 - (uint32_t)decay) >> 15);
            while (symbol > 1 && value < mag) {
                value++;
                symbol *= 2;
                low += symbol;
                symbol = (((symbol - 2) * (uint32_t)decay) >> 15) + 1;
            }
            if (symbol <= 1 && value < mag) {
                uint32_t dist = (uint32_t)(mag - value), room = (32768 - 2 - low) >> 1;       /* both signs must still fit below 32768 */
                if (dist > room) dist = room;
                value += (int)dist;
                low += 2 * dist;
            }
            if (want < 0) value = -value;
            else low += symbol;
        }
        uint32_t fhigh = low + symbol;
        orc_rd_update(rd, scale, low, fhigh < 32768 ? fhigh : 32768, 32768);
        return value;
    }
    if (center >= symbol) {
        value = 1;
        low = symbol;
        symbol = 1 + ((32768 - 32 - symbol) * (16384 - (uint32_t)decay) >> 15);
        while (symbol > 1 && center >= low + 2 * symbol) {
            value++;
            symbol *= 2;
            low += symbol;
            symbol = (((symbol - 2) * (uint32_t)decay) >> 15) + 1;
        }
        if (symbol <= 1) {
            uint32_t dist = (center - low) >> 1;
            value += (int)dist;
            low += 2 * dist;
        }
        if (center < low + symbol) value = -value;
        else low += symbol;
    }
    uint32_t high = low + symbol;
    orc_rd_update(rd, scale, low, high < 32768 ? high : 32768, 32768);
    return value;
}

static inline uint32_t orc_rd_step(OrcRd *rd, uint32_t k0)   /* entropy.rs:487-512, RFC-corrected */
{
    uint32_t k1 = (k0 + 1) * 3, total = k1 + k0, scale;
    uint32_t sym = orc_rd_symbol(rd, total, &scale);
    uint32_t k = sym < k1 ? sym / 3 : sym - (k0 + 1) * 2;
    int64_t fv;
    if (orc_forced(&fv)) k = (uint64_t)fv <= 2 * k0 ? (uint32_t)fv : 2 * k0;
    if (k <= k0) orc_rd_update(rd, scale, 3 * k, 3 * (k + 1), total);
    else orc_rd_update(rd, scale, (k - 1 - k0) + k1, (k - k0) + k1, total);
    return k;
}

static inline uint32_t orc_isqrt(uint32_t v)   /* exact floor sqrt (integer-sqrt 0.1.2 semantics) */
{
    uint32_t r = 0, b = 1u << 15;
    while (b) { uint32_t t = r | b; if ((uint64_t)t * t <= v) r = t; b >>= 1; }
    return r;
}

static inline uint32_t orc_rd_triangular(OrcRd *rd, uint32_t qn)   /* entropy.rs:514-538 */
{
    uint32_t qn2 = qn >> 1, total = (qn2 + 1) * (qn2 + 1), scale;
    uint32_t center = orc_rd_symbol(rd, total, &scale);
    uint32_t k, low, sym;
    int64_t fv;
    if (orc_forced(&fv)) {
        k = (uint64_t)fv <= qn ? (uint32_t)fv : qn;
        if (k <= qn2) { low = k * (k + 1) >> 1; sym = k + 1; }
        else { low = total - ((qn + 1 - k) * (qn + 2 - k) >> 1); sym = qn + 1 - k; }
        orc_rd_update(rd, scale, low, low + sym, total);
        return k;
    }
    if (center < total >> 1) {
        k = (orc_isqrt(8 * center + 1) - 1) >> 1;
        low = k * (k + 1) >> 1;
        sym = k + 1;
    } else {
        k = (2 * (qn + 1) - orc_isqrt(8 * (total - center - 1) + 1)) >> 1;
        low = total - ((qn + 1 - k) * (qn + 2 - k) >> 1);
        sym = qn + 1 - k;
    }
    orc_rd_update(rd, scale, low, low + sym, total);
    return k;
}

static inline void orc_rd_to_end(OrcRd *rd) { rd->total += rd->size_bits - orc_rd_tell(rd); }  /* :540 */

#endif
