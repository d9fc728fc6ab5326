/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under opus_amd/ may include, link or call this.
 *
 * CPU restatement of the reference's packet framing (lu-zero/opus src/packet.rs) and per-packet driver
 * (src/decoder.rs:48-127), plus the 8/12/16 -> 48 kHz resampler the reference lacks (the
 * build's own design, "parity unpinned").  Framing follows RFC 6716 section 3 where packet.rs is
 * defective (SURVEY section 8 notes: code-1 split at packet.rs:133, code-3 VBR last frame at :177-188,
 * empty packet at :215).
 */
#include <math.h>
#include <stdlib.h>
#include "orc_rangedec.h"
#include "orc_api.h"

/* ------------------------------------------------------------------ entropy KAT shims */
void orc_kat_reverse_bits(const uint8_t *buf, int len, const int *widths, int n, uint32_t *out)
{   /* entropy.rs:82-133 */
    OrcRd rd;
    orc_rd_init(&rd, buf, len);
    for (int i = 0; i < n; i++) out[i] = orc_rd_rawbits(&rd, widths[i]);
}
void orc_kat_laplace(const uint8_t *buf, int len, const int *symbol, const int *decay, int n, int *out)
{   /* entropy.rs:135-257 */
    OrcRd rd;
    orc_rd_init(&rd, buf, len);
    for (int i = 0; i < n; i++) out[i] = orc_rd_laplace(&rd, symbol[i], decay[i]);
}

/* ------------------------------------------------------------------ packet.rs */
static int frame_len_1or2(const uint8_t *b, int avail, int *used)   /* packet.rs:77-89 */
{
    if (avail < 1) return -1;
    int v = b[0];
    if (v >= 252) {
        if (avail < 2) return -1;
        *used = 2;
        return v + 4 * b[1];
    }
    *used = 1;
    return v;
}

int orc_packet_parse(const uint8_t *buf, int len, OrcPacket *p)   /* packet.rs:202-317 */
{
    memset(p, 0, sizeof *p);
    if (len < 1) return -1;
    int toc = buf[0], config = toc >> 3;
    p->code = toc & 3;
    p->stereo = (toc >> 2) & 1;
    if (config < 12) {
        p->mode = 0; p->bandwidth = config >> 2;
        static const int d[4] = {480, 960, 1920, 2880};
        p->frame_size = d[config & 3];
    } else if (config < 16) {
        p->mode = 1; p->bandwidth = config < 14 ? 3 : 4;
        p->frame_size = (config & 1) ? 960 : 480;
    } else {
        p->mode = 2;
        static const int bw[4] = {0, 2, 3, 4};
        p->bandwidth = bw[(config - 16) >> 2];
        p->frame_size = 120 << (config & 3);
    }
    int pos = 1, rem = len - 1;
    switch (p->code) {
    case 0:
        p->nframes = 1; p->off[0] = pos; p->len[0] = rem;
        break;
    case 1:
        if (rem & 1) return -1;
        p->nframes = 2;
        p->off[0] = pos; p->len[0] = rem / 2;
        p->off[1] = pos + rem / 2; p->len[1] = rem / 2;
        break;
    case 2: {
        int used, l = frame_len_1or2(buf + pos, rem, &used);
        if (l < 0 || l + used > rem) return -1;
        p->nframes = 2;
        p->off[0] = pos + used; p->len[0] = l;
        p->off[1] = pos + used + l; p->len[1] = rem - used - l;
        break;
    }
    default: {
        if (rem < 1) return -1;
        int fc = buf[pos], vbr = fc >> 7, pad = (fc >> 6) & 1, count = fc & 0x3f;
        pos++; rem--;
        if (count == 0 || count * p->frame_size > 5760) return -1;
        if (pad) {                                   /* packet.rs:91-111 */
            int padding = 0, b;
            do {
                if (rem < 1) return -1;
                b = buf[pos++]; rem--;
                padding += b == 255 ? 254 : b;
            } while (b == 255);
            if (padding > rem) return -1;
            p->padding = padding;
            rem -= padding;
        }
        p->nframes = count;
        if (vbr) {
            int total = 0;
            for (int i = 0; i < count - 1; i++) {
                int used, l = frame_len_1or2(buf + pos, rem, &used);
                if (l < 0) return -1;
                pos += used; rem -= used;
                p->len[i] = l; total += l;
            }
            if (total > rem) return -1;
            for (int i = 0; i < count - 1; i++) { p->off[i] = pos; pos += p->len[i]; }
            p->off[count - 1] = pos; p->len[count - 1] = rem - total;
        } else {
            if (rem % count) return -1;
            int l = rem / count;
            for (int i = 0; i < count; i++) { p->off[i] = pos + i * l; p->len[i] = l; }
        }
    }
    }
    for (int i = 0; i < p->nframes; i++) if (p->len[i] > 1275) return -1;
    return 0;
}

/* ------------------------------------------------------------------ resampler (no reference code) */
/* Zero-stuff by L = 48/rate_khz, 73-tap Kaiser(beta 8) windowed-sinc low-pass at 0.46*fs_in, gain L.
 * Group delay 36 samples @48 kHz for every input rate.  y[m] = sum_k h[m - kL] x[k], k ascending. */
#define RS_HALF 36
#define RS_TAPS (2 * RS_HALF + 1)
static float g_rs[3][RS_TAPS];
static int g_rs_init = 0;
static double bessel_i0(double x) { double s = 1, t = 1; for (int k = 1; k < 40; k++) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; } return s; }
static void rs_init(void)
{
    static const int rates[3] = {8, 12, 16};
    for (int r = 0; r < 3; r++) {
        int L = 48 / rates[r];
        double fc = 0.46 * rates[r] / 48.0;          /* cycles/sample at 48 kHz */
        for (int n = 0; n < RS_TAPS; n++) {
            double t = n - RS_HALF;
            double sinc = t == 0 ? 2 * fc : sin(2 * M_PI * fc * t) / (M_PI * t);
            double w = bessel_i0(8.0 * sqrt(1 - (t / RS_HALF) * (t / RS_HALF))) / bessel_i0(8.0);
            g_rs[r][n] = (float)(L * sinc * w);
        }
    }
    g_rs_init = 1;
}
const float *orc_resample_taps(int rate_khz) { if (!g_rs_init) rs_init(); return g_rs[rate_khz == 8 ? 0 : rate_khz == 12 ? 1 : 2]; }

void orc_resample_to_48k(const float *in, int n_in, int rate_khz, float *hist, float *out)
{   /* hist: last 32 input samples of the previous call (zeros at reset), updated on return */
    const float *h = orc_resample_taps(rate_khz);
    int L = 48 / rate_khz;
    for (int m = 0; m < n_in * L; m++) {
        int k_hi = m / L;                                   /* floor(m/L) */
        int k_lo = (m - 2 * RS_HALF + L - 1 + 64 * L) / L - 64;   /* ceil((m-72)/L) */
        float acc = 0;
        for (int k = k_lo; k <= k_hi; k++) {
            float x = k >= 0 ? in[k] : hist[32 + k];
            acc += h[m - k * L] * x;
        }
        out[m] = acc;
    }
    float nh[32];
    for (int i = 0; i < 32; i++) { int k = n_in - 32 + i; nh[i] = k >= 0 ? in[k] : hist[32 + k]; }
    memcpy(hist, nh, sizeof nh);
}

/* ------------------------------------------------------------------ decoder.rs driver */
int orc_drop_redundancy = 0;                 /* test switch: parse the redundant frames and drop them, as the reference does */
void orc_set_drop_redundancy(int v) { orc_drop_redundancy = v; }
long orc_redundant_frames[2] = {0, 0};       /* test statistics: redundant frames decoded, [0] SILK -> CELT, [1] CELT -> SILK */
struct OrcDecoder {
    int channels, prev_mode;
    int last_frame_size;          /* samples per channel of the last packet (what a lost packet is taken to span) */
    int last_opus_frame;          /* samples per channel of one Opus frame of the last packet */
    int last_bandwidth, last_stereo;
    int lost_silk, lost_celt;     /* SILK / CELT frames concealed since the last decoded packet */
    int prev_redundancy;          /* the previous frame ended in a SILK -> CELT redundant frame: CELT continues from it */
    uint32_t final_range;         /* of the last frame: the main decoder's range XOR the redundant frame's (RFC 6716 4.5.1.4) */
    OrcSilk *silk;
    OrcCelt *celt;
    float rs_hist[2][32];
};

OrcDecoder *orc_decoder_create(int channels)   /* decoder.rs:133-182 (configure) */
{
    OrcDecoder *d = (OrcDecoder *)calloc(1, sizeof *d);
    d->channels = channels; d->prev_mode = -1; d->last_frame_size = d->last_opus_frame = 960; d->last_bandwidth = 4;
    d->silk = orc_silk_create(channels > 1);
    d->celt = orc_celt_create(channels);
    return d;
}
void orc_decoder_destroy(OrcDecoder *d) { orc_silk_destroy(d->silk); orc_celt_destroy(d->celt); free(d); }
uint32_t orc_decoder_final_range(const OrcDecoder *d) { return d->final_range; }
void orc_decoder_set_celt_trace(OrcDecoder *d, OrcCeltTrace *t) { orc_celt_set_trace(d->celt, t); }

int orc_decoder_decode(OrcDecoder *d, const uint8_t *pkt, int len, float *pcm, int cap)
{   /* decoder.rs:48-127 */
    OrcPacket p;
    if (len > 0) {                                            /* a packet of empty frames (TOC only: DTX) is concealed as well */
        if (orc_packet_parse(pkt, len, &p)) return -1;
        int empty = 1;
        for (int f = 0; f < p.nframes; f++) empty = empty && p.len[f] == 0;
        if (empty) { d->last_frame_size = p.nframes * p.frame_size; d->last_opus_frame = p.frame_size; len = 0; }
    }
    if (len == 0) {
        /* A lost packet (or a DTX gap) is concealed over the span of the last packet; no counterpart in the reference (it panics
           on an empty packet, packet.rs:215-225).  In the mode of the last packet, as libopus does it:
           SILK    the SILK frames continue from the last one's parameters (orc_silk_conceal: pitch repetition / noise);
           CELT    noise shaped by the last band energies, which decay by 1.5 (first frame) and 0.5 per frame in log2 units;
           hybrid  both, the CELT part from band 17.
           Decoder state runs on, so the next packet continues it.  Before any packet, and when SILK has nothing to continue
           from: silent CELT frames (the MDCT overlap fades out over 2.5 ms, then zeros) and SILK restarts. */
        static const uint8_t silence[2] = {0xFF, 0xFF};
        static const int celt_end[5] = {13, 17, 17, 19, 21};
        static const int silk_rate[3] = {8, 12, 16};
        int CC = d->channels, total = d->last_frame_size, N = d->last_opus_frame, mode = d->prev_mode;
        if (total > cap) return -3;
        int silk_ok = mode >= 0 && mode != 2 && N >= 480 && orc_silk_can_conceal(d->silk);
        int celt_noise = mode == 2 || (mode == 1 && silk_ok);
        for (int o = 0; o < total; o += N) {
            float ch[2][2880];
            float *dst = pcm + (size_t)o * CC;
            memset(ch, 0, sizeof ch);
            if (silk_ok) {
                float sl[960], sr[960];
                int bw = d->last_bandwidth < 2 ? d->last_bandwidth : 2;
                int n = orc_silk_conceal(d->silk, bw, N / 48, d->last_stereo, d->lost_silk, sl, sr);
                d->lost_silk += N <= 960 ? 1 : N / 960;
                if (CC == 2) {
                    orc_resample_to_48k(sl, n, silk_rate[bw], d->rs_hist[0], ch[0]);
                    orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[1], ch[1]);
                } else orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[0], ch[0]);
            }
            if (!(silk_ok && mode == 0))
                for (int q = 0, M; q < N; q += M) {             /* in CELT frame sizes, largest first */
                    int left = N - q;
                    M = left >= 960 ? 960 : left >= 480 ? 480 : left >= 240 ? 240 : 120;
                    float c0[960], c1[960];
                    if (celt_noise)
                        orc_celt_conceal_frame(d->celt, M, mode == 1 ? 17 : 0, celt_end[d->last_bandwidth], d->last_stereo + 1,
                                               d->lost_celt++ ? 0.5f : 1.5f, c0, c1);
                    else {
                        OrcRd r;
                        orc_rd_init(&r, silence, 2);
                        orc_celt_decode_frame(d->celt, &r, M, 0, 21, CC, c0, c1);
                    }
                    for (int i = 0; i < M; i++) { ch[0][q + i] += c0[i]; if (CC == 2) ch[1][q + i] += c1[i]; }
                }
            for (int i = 0; i < N; i++) for (int c = 0; c < CC; c++) dst[i * CC + c] = ch[c][i];
        }
        if (!silk_ok) { orc_silk_flush(d->silk); memset(d->rs_hist, 0, sizeof d->rs_hist); }
        d->prev_redundancy = 0;
        d->final_range = 0;
        return total;
    }
    if (orc_packet_parse(pkt, len, &p)) return -1;
    int CC = d->channels, total = p.nframes * p.frame_size;
    if (total > cap) return -3;
    static const int celt_end[5] = {13, 17, 17, 19, 21};     /* packet.rs:33-42 */
    static const int silk_rate[3] = {8, 12, 16};
    /* Mode transitions (decoder.rs:61-67, :79 and, where the reference parses and drops, RFC 6716 4.5): SILK restarts after
       CELT-only packets; CELT restarts when the mode changes -- unless the previous frame ended in a SILK -> CELT redundant
       frame, from which CELT continues -- and only after a CELT -> SILK redundant frame has been decoded from the old state. */
    int celt_restart = p.mode != d->prev_mode && d->prev_mode >= 0 && !d->prev_redundancy && p.mode != 0;
    if (p.mode != d->prev_mode && d->prev_mode == 2) { orc_silk_flush(d->silk); memset(d->rs_hist, 0, sizeof d->rs_hist); }
    const float *win = orc_celt_window();
    for (int f = 0; f < p.nframes; f++) {
        OrcRd rd;
        int flen = p.len[f], N = p.frame_size;
        float *dst = pcm + (size_t)f * N * CC;
        float ch[2][2880];
        memset(ch, 0, sizeof ch);
        orc_rd_init(&rd, pkt + p.off[f], flen);
        if (p.mode != 2) {
            float sl[960], sr[960];
            int bw = p.bandwidth < 2 ? p.bandwidth : 2;
            int n = orc_silk_decode(d->silk, &rd, bw, N / 48, p.stereo, sl, sr);
            if (n < 0) return n;
            /* reference output buffers: right is always written, left only for stereo_out (:2104-2113) */
            if (CC == 2) {
                orc_resample_to_48k(sl, n, silk_rate[bw], d->rs_hist[0], ch[0]);
                orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[1], ch[1]);
            } else orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[0], ch[0]);
        } else orc_silk_flush(d->silk);                       /* decoder.rs:79 */
        int consumed = orc_rd_tell(&rd), redundancy = 0, celt_to_silk = 0, rbytes = 0;     /* decoder.rs:82-115 */
        const uint8_t *rdata = NULL;
        if (p.mode == 1 && consumed + 37 <= flen * 8) redundancy = orc_rd_logp(&rd, 12);
        else if (p.mode == 0 && consumed + 17 <= flen * 8) redundancy = 1;
        if (redundancy) {
            celt_to_silk = orc_rd_logp(&rd, 1);               /* position flag: 1 = the redundant frame belongs in front */
            rbytes = p.mode == 1 ? (int)orc_rd_uniform(&rd, 256) + 2 : flen - ((orc_rd_tell(&rd) + 7) >> 3);
            if (rbytes >= flen) return -1;
            int nl = flen - rbytes;
            if (nl * 8 < orc_rd_tell(&rd)) { nl = 0; redundancy = 0; rbytes = 0; }
            else rdata = pkt + p.off[f] + nl;
            rd.len = nl; rd.size_bits = nl * 8;
            if (orc_drop_redundancy) redundancy = 0;
        }
        /* RFC 6716 4.5.1: the redundant frame is a 5 ms CELT frame from band 0 (the reference drops it, decoder.rs:84-115) */
        float red[2][240];
        uint32_t red_range = 0;
        memset(red, 0, sizeof red);
        if (redundancy) orc_redundant_frames[celt_to_silk]++;
        if (redundancy && celt_to_silk) {                     /* CELT -> SILK: decoded first, from the running CELT state */
            OrcRd r2;
            orc_rd_init(&r2, rdata, rbytes);
            orc_celt_decode_frame(d->celt, &r2, 240, 0, celt_end[p.bandwidth], p.stereo + 1, red[0], red[1]);
            red_range = r2.range;
        }
        if (f == 0 && celt_restart) orc_celt_reset(d->celt);
        if (p.mode != 0 && N <= 960) {
            float c0[960], c1[960];
            int start = p.mode == 1 ? 17 : 0;
            orc_celt_decode_frame(d->celt, &rd, N, start, celt_end[p.bandwidth], p.stereo + 1, c0, c1);
            for (int i = 0; i < N; i++) { ch[0][i] += c0[i]; if (CC == 2) ch[1][i] += c1[i]; }
        } else if (p.mode == 0 && d->prev_mode == 1 && !(redundancy && celt_to_silk && d->prev_redundancy)) {
            /* hybrid -> SILK: a 2.5 ms silent CELT frame lets the MDCT overlap of the last hybrid frame fade out (RFC 6716 4.5.2) */
            static const uint8_t silence[2] = {0xFF, 0xFF};
            float c0[960], c1[960];
            OrcRd r3;
            orc_rd_init(&r3, silence, 2);
            orc_celt_decode_frame(d->celt, &r3, 120, 0, celt_end[p.bandwidth], p.stereo + 1, c0, c1);
            for (int i = 0; i < 120; i++) { ch[0][i] += c0[i]; if (CC == 2) ch[1][i] += c1[i]; }
        }
        if (redundancy && !celt_to_silk) {                    /* SILK -> CELT: decoded last, from a fresh CELT state, which the next
                                                                 (CELT) packet continues; cross-faded over the frame's last 2.5 ms */
            OrcRd r2;
            orc_rd_init(&r2, rdata, rbytes);
            orc_celt_reset(d->celt);
            orc_celt_decode_frame(d->celt, &r2, 240, 0, celt_end[p.bandwidth], p.stereo + 1, red[0], red[1]);
            red_range = r2.range;
            for (int c = 0; c < CC; c++)
                for (int i = 0; i < 120; i++) {
                    float w = win[i] * win[i];
                    ch[c][N - 120 + i] = w * red[c][120 + i] + (1.f - w) * ch[c][N - 120 + i];
                }
        }
        if (redundancy && celt_to_silk) {                     /* first 2.5 ms from the redundant frame, the next 2.5 ms cross-faded */
            for (int c = 0; c < CC; c++) {
                for (int i = 0; i < 120; i++) ch[c][i] = red[c][i];
                for (int i = 0; i < 120; i++) {
                    float w = win[i] * win[i];
                    ch[c][120 + i] = w * ch[c][120 + i] + (1.f - w) * red[c][120 + i];
                }
            }
        }
        for (int i = 0; i < N; i++) for (int c = 0; c < CC; c++) dst[i * CC + c] = ch[c][i];
        d->final_range = rd.range ^ red_range;
        d->last_frame_size = total; d->last_opus_frame = N;
        d->last_bandwidth = p.bandwidth; d->last_stereo = p.stereo;
        d->lost_silk = d->lost_celt = 0;
        d->prev_mode = p.mode;
        d->prev_redundancy = redundancy && !celt_to_silk;
        celt_restart = 0;
    }
    d->prev_mode = p.mode;
    return total;
}

/* In-band FEC recovery (RFC 6716 4.2.4-4.2.7; libopus' decode_fec; no counterpart in the reference, which returns Unsupported at
 * the LBRR flag, silk/decoder.rs:2070-2072): pkt is the packet FOLLOWING a lost one.  Returns the lost packet's audio over pkt's
 * duration: pkt's LBRR frames decoded into the running state; the CELT part of a hybrid packet, and everything of a packet
 * without LBRR frames, concealed as after a loss.  The caller decodes pkt itself afterwards. */
int orc_decoder_decode_fec(OrcDecoder *d, const uint8_t *pkt, int len, float *pcm, int cap)
{
    OrcPacket p;
    if (len <= 0) return orc_decoder_decode(d, NULL, 0, pcm, cap);
    if (orc_packet_parse(pkt, len, &p)) return -1;
    int any = 0;
    for (int f = 0; f < p.nframes && p.mode != 2; f++) {
        if (p.len[f] == 0) continue;
        OrcRd rd;
        orc_rd_init(&rd, pkt + p.off[f], p.len[f]);
        int nsf = p.frame_size <= 960 ? 1 : p.frame_size / 960, a;
        for (int i = 0; i < nsf; i++) (void)orc_rd_logp(&rd, 1);
        a = orc_rd_logp(&rd, 1);
        if (p.stereo) {
            for (int i = 0; i < nsf; i++) (void)orc_rd_logp(&rd, 1);
            a |= orc_rd_logp(&rd, 1);
        }
        any |= a;
    }
    if (!any) {
        d->last_frame_size = p.nframes * p.frame_size; d->last_opus_frame = p.frame_size;
        return orc_decoder_decode(d, NULL, 0, pcm, cap);
    }
    int CC = d->channels, total = p.nframes * p.frame_size;
    if (total > cap) return -3;
    static const int silk_rate[3] = {8, 12, 16};
    static const uint8_t silence[2] = {0xFF, 0xFF};
    if (p.mode != d->prev_mode && d->prev_mode == 2) { orc_silk_flush(d->silk); memset(d->rs_hist, 0, sizeof d->rs_hist); }
    if (p.mode != d->prev_mode && d->prev_mode >= 0 && !d->prev_redundancy && p.mode != 0) orc_celt_reset(d->celt);
    for (int f = 0; f < p.nframes; f++) {
        OrcRd rd;
        int N = p.frame_size;
        float *dst = pcm + (size_t)f * N * CC;
        float ch[2][2880], sl[960], sr[960];
        memset(ch, 0, sizeof ch);
        orc_rd_init(&rd, pkt + p.off[f], p.len[f]);
        int bw = p.bandwidth < 2 ? p.bandwidth : 2;
        int n = orc_silk_decode_fec(d->silk, &rd, bw, N / 48, p.stereo, sl, sr);
        if (n < 0) return n;
        if (CC == 2) {
            orc_resample_to_48k(sl, n, silk_rate[bw], d->rs_hist[0], ch[0]);
            orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[1], ch[1]);
        } else orc_resample_to_48k(sr, n, silk_rate[bw], d->rs_hist[0], ch[0]);
        if (p.mode == 1 || d->prev_mode != 0) {          /* the CELT part is not in the copy; a running MDCT overlap fades out */
            for (int o = 0, M; o < N; o += M) {
                M = N - o >= 960 ? 960 : N - o >= 480 ? 480 : N - o >= 240 ? 240 : 120;
                float c0[960], c1[960];
                OrcRd r;
                orc_rd_init(&r, silence, 2);
                orc_celt_decode_frame(d->celt, &r, M, 0, 21, CC, c0, c1);
                for (int i = 0; i < M; i++) { ch[0][o + i] += c0[i]; if (CC == 2) ch[1][o + i] += c1[i]; }
            }
        }
        for (int i = 0; i < N; i++) for (int c = 0; c < CC; c++) dst[i * CC + c] = ch[c][i];
        d->prev_mode = p.mode;
    }
    d->prev_redundancy = 0;
    d->last_frame_size = total; d->last_opus_frame = p.frame_size;
    d->last_bandwidth = p.bandwidth; d->last_stereo = p.stereo;
    d->lost_silk = d->lost_celt = 0;
    d->final_range = 0;
    return total;
}

void orc_kat_rd_init(OrcRd *rd, const uint8_t *buf, int len) { orc_rd_init(rd, buf, len); }
