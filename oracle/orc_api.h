/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_celt.c).  Public surface of liborc.so, bound with
 * ctypes from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- and from nowhere else. */
#ifndef ORC_API_H
#define ORC_API_H
#include <stdint.h>
#include "orc_rangedec.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEAVES 1024
#define ORC_CELT_HIST 1024

enum { ORC_LEAF_PVQ = 0, ORC_LEAF_ZERO = 1, ORC_LEAF_NOISE = 2, ORC_LEAF_FOLD = 3, ORC_LEAF_N1 = 4 };

typedef struct {          /* one decode_band_no_split / decode_band_1 call (celt/decoder.rs:1601, :1466) */
    int32_t band, ch, off, n, k, blocks, kind;
    uint32_t idx;
    float gain;
} OrcLeaf;

typedef struct {          /* everything the reference parses for one CELT frame, plus stage outputs */
    int32_t lm, start, end, C, silence, transient, intra, spread;
    int32_t pf_on, pf_period, pf_gain_q, pf_tapset;
    int32_t alloc_trim, intensity, dual_stereo, codedbands, anticollapse_rsv, anticollapse_on, tell_end;
    uint32_t seed_in, seed_bands, seed_out;
    int32_t tf_change[21], pulses_alloc[21], fine_bits[21], fine_prio[21], boost[21];
    int32_t band_b[21], band_lowband[21], itheta[21], inv[21];
    float energy[2][21];
    uint8_t collapse[2][21];
    uint8_t pad_[2];
    int32_t nleaves;
    OrcLeaf leaf[ORC_MAX_LEAVES];
    int16_t pulses[2][960];       /* PVQ pulse vectors at their coefficient positions */
    float X[2][960];              /* normalised coefficients after band decoding */
    float freq[2][960];           /* denormalised MDCT coefficients */
} OrcCeltTrace;

typedef struct OrcCelt {
    int channels;
    float oldE[2][21], logE1[2][21], logE2[2][21];
    float mem[2][ORC_CELT_HIST + 960 + 60];
    float deemph[2];
    int pf_period, pf_period_old, pf_tapset, pf_tapset_old;
    float pf_gain, pf_gain_old;
    uint32_t rng;
    OrcCeltTrace *trace;
} OrcCelt;

/* CELT */
OrcCelt *orc_celt_create(int channels);
void orc_celt_destroy(OrcCelt *);
void orc_celt_reset(OrcCelt *);
void orc_celt_set_trace(OrcCelt *, OrcCeltTrace *);
int orc_celt_decode_frame(OrcCelt *, OrcRd *, int N, int start, int end, int C, float *pcm0, float *pcm1);
int orc_celt_conceal_frame(OrcCelt *, int N, int start, int end, int C, float decay, float *pcm0, float *pcm1);
uint32_t orc_cwrsi(int n, int k, uint32_t i, int *y);
uint32_t orc_pvq_v(int n, int k);
uint32_t orc_icwrs(int n, const int *y);
/* the decoder run as an encoder (tests/test_loopback.py; see orc_celt.c): NULL switches it off */
typedef int64_t (*OrcEncHook)(int kind, int band, int ch, int a, int b, int c, double f, int *y);
void orc_set_encoder(OrcEncHook hook);
void orc_set_encoder_opts(int transient, int dual_stereo, int spread);
void orc_set_encoder_postfilter(int period, int gain_index, int tapset);
void orc_set_encoder_tf(int mode);
int orc_extract_collapse_mask(const int *y, int n, int b);
void orc_haar1(float *x, int n0, int stride);
void orc_exp_rotation(float *x, int len, int stride, int k, int spread);
void orc_stereo_merge(float *x, float *y, float mid, int n);
void orc_fft(const float *in_re_im, int stride, int n, float *out_re_im);
void orc_imdct_raw(const float *in, int stride, int n2, float *D);
void orc_imdct_raw_direct(const float *in, int stride, int n2, double *D);
const float *orc_celt_window(void);

/* SILK (internal-rate output, silk/decoder.rs) */
typedef struct OrcSilk OrcSilk;
OrcSilk *orc_silk_create(int stereo_out);
void orc_silk_destroy(OrcSilk *);
void orc_silk_flush(OrcSilk *);
/* bandwidth: 0 NB, 1 MB, 2 WB; duration_ms 10/20/40/60; returns samples/channel or <0 */
int orc_silk_decode(OrcSilk *, OrcRd *, int bandwidth, int duration_ms, int stereo_pkt, float *left, float *right);
int orc_silk_decode_fec(OrcSilk *, OrcRd *, int bandwidth, int duration_ms, int stereo_pkt, float *left, float *right);
int orc_silk_can_conceal(const OrcSilk *);
int orc_silk_conceal(OrcSilk *, int bandwidth, int duration_ms, int stereo_pkt, int lost, float *left, float *right);
void orc_lsf_to_lpc(const int16_t *nlsf, int order, float *lpc);

/* entropy KAT shims */
void orc_kat_reverse_bits(const uint8_t *buf, int len, const int *widths, int n, uint32_t *out);
void orc_kat_laplace(const uint8_t *buf, int len, const int *symbol, const int *decay, int n, int *out);

/* packets + whole decoder (decoder.rs:16-188, packet.rs) */
typedef struct {
    int mode;          /* 0 SILK, 1 HYBRID, 2 CELT */
    int bandwidth;     /* 0 NB 1 MB 2 WB 3 SWB 4 FB */
    int frame_size;    /* samples @48k per frame */
    int stereo, code, nframes, padding;
    int off[48], len[48];
} OrcPacket;
int orc_packet_parse(const uint8_t *buf, int len, OrcPacket *p);

typedef struct OrcDecoder OrcDecoder;
OrcDecoder *orc_decoder_create(int channels);
void orc_decoder_destroy(OrcDecoder *);
uint32_t orc_decoder_final_range(const OrcDecoder *);   /* last frame: main range XOR the redundant frame's */
void orc_decoder_set_celt_trace(OrcDecoder *, OrcCeltTrace *);
/* decode one packet -> interleaved f32 @48 kHz; returns samples per channel, <0 on error */
int orc_decoder_decode(OrcDecoder *, const uint8_t *pkt, int len, float *pcm, int cap_samples);
int orc_decoder_decode_fec(OrcDecoder *, const uint8_t *pkt, int len, float *pcm, int cap_samples);
/* 8/12/16 kHz -> 48 kHz polyphase FIR used by the decoder (build's own design) */
void orc_resample_to_48k(const float *in, int n_in, int rate_khz, float *hist /*[32]*/, float *out);

#ifdef __cplusplus
}
#endif
#endif
