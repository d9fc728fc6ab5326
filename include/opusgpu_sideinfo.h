/* Side-info records: the host/device cut of the Opus decode path.
 *
 * The host keeps everything that touches the range decoder (reference: src/entropy.rs, src/packet.rs,
 * the integer control of src/celt/decoder.rs and src/silk/decoder.rs); it emits these records, which
 * the HIP kernels consume.  Every field names the reference line that produces it.  Plain C layout,
 * shared verbatim by host C++ and device code.
 */
#ifndef OPUSGPU_SIDEINFO_H
#define OPUSGPU_SIDEINFO_H
#include <stdint.h>

#define OG_MAX_BANDS 21
#define OG_MAX_FRAME 960

/* ---- CELT ------------------------------------------------------------------------------------ */
enum {                       /* OgCeltHdr.flags */
    OG_F_TRANSIENT = 1,      /* celt/decoder.rs:2255-2263 */
    OG_F_SILENCE = 2,        /* celt/decoder.rs:2237-2248 */
    OG_F_STEREO = 4,         /* packet stereo flag: two coded channels (celt/decoder.rs:868) */
    OG_F_ANTICOLLAPSE = 8,   /* the bit reserved at celt/decoder.rs:1092, read per RFC 6716 4.3.5 */
    OG_F_RESET = 16,         /* decoder state reset before this frame (mode switch / first frame) */
    OG_F_OUT_STEREO = 32,    /* stream has two output channels (decoder.rs:145) */
    OG_F_ACCUM = 64          /* hybrid: add to the PCM the SILK kernels already wrote (decoder.rs:117-123) */
};

enum {                       /* OgCeltBand.flags */
    OG_B_JOINT = 1,          /* band coded as mid/side with a theta (celt/decoder.rs:1795-1976) */
    OG_B_INV = 2,            /* side inversion flag (celt/decoder.rs:1554-1559) */
    OG_B_N2SWAP = 4,         /* N==2 stereo special case, itheta > 8192 (celt/decoder.rs:1835-1839) */
    OG_B_N2NEG = 8,          /* N==2 stereo special case, sign bit set (celt/decoder.rs:1841) */
    OG_B_DUAL = 16,          /* dual stereo: X and Y coded independently (celt/decoder.rs:2155-2193) */
    OG_B_MIDFILL0 = 32,      /* compute_theta cleared the mid's fill (itheta == 16384, celt/decoder.rs:1573) */
    OG_B_MERGE_NORM = 64,    /* dual -> intensity switch at this band: norm = (norm+norm2)/2 (:2133-2138) */
    OG_B_LAST = 128,         /* last coded band: no lowband_out (celt/decoder.rs:2143-2147) */
    OG_B_FOLD_X = 256,       /* an X-group fill leaf has a folding source: the lowband copy must be prepared */
    OG_B_FOLD_Y = 512        /* same for the Y group (dual stereo only) */
};

#if defined(__cplusplus)
#define OG_ALIGNED(n) alignas(n)
#else
#define OG_ALIGNED(n) _Alignas(n)
#endif
typedef struct {             /* one band of one frame, 16 bytes, dword-aligned (it sits at a multiple of 4 in OgCeltHdr: the
                                kernels fetch it as four dwords instead of a dozen bytes and shorts) */
    OG_ALIGNED(4) uint16_t imid;           /* Q15 cos/sin of theta (celt/decoder.rs:1564-1582) */
    uint16_t iside;
    uint16_t flags;
    uint16_t leaf0;          /* the band's first leaf, relative to OgCeltHdr.leaf_begin.  The band's leaves are grouped:
                                the nx leaves that write the X buffer, then the ny that write Y; inside each group the
                                fill leaves (K == 0) come first.  Decode order is kept inside each run, which is all the
                                noise generator's sequence needs (celt/decoder.rs:1595-1599; side leaves never draw). */
    uint8_t nx, ny;          /* leaves writing X / Y */
    uint8_t fx, fy;          /* of those, fill-path leaves (K == 0, N > 1) at the front of the group */
    uint8_t lowband;         /* effective_lowband / M, 0xFF = none (celt/decoder.rs:2088-2129) */
    int8_t tf_change;        /* celt/decoder.rs:994-1033 */
    uint8_t fold_lo, fold_hi;/* bands [fold_lo, fold_hi) whose collapse masks form the fill (:2097-2120) */
} OgCeltBand;

enum {                       /* OgCeltLeaf.meta */
    OG_L_LOGB_MASK = 7,      /* log2 of the leaf's block count B (1..16: tf time-division can double 8 short blocks) */
    OG_L_FILL_DEAD = 8,      /* compute_theta cleared this subtree's fill bits (itheta 0 / 16384, :1567-1577) */
    OG_L_CH_Y = 32,          /* leaf writes the Y (second channel) buffer */
    OG_L_N1 = 64,            /* decode_band_1 leaf: index is the sign bit (celt/decoder.rs:1466-1495) */
    OG_L_HAS_LOWBAND = 128   /* a folding source exists for this leaf's quant_band */
};

typedef struct {             /* one decode_band_no_split / decode_band_1 call, 16 bytes */
    uint32_t index;          /* decode_uniform(V(N,K)) result (celt/decoder.rs:670) or N==1 sign */
    float gain;              /* product of mid/side gains down the recursion (:1912,1932,1949,1970) */
    uint8_t off;             /* offset inside the band */
    uint8_t n;               /* dimension N */
    uint8_t k;               /* pulses K, 0 = fill path (celt/decoder.rs:1626-1661) */
    uint8_t meta;
    uint8_t cmshift;         /* position of the leaf's blocks inside its quant_band: its collapse mask is
                                shifted left by this (:1934) and its fill is the quant_band fill shifted right by it */
    uint8_t band;            /* band index */
    uint16_t pos;            /* first coefficient of the leaf inside its channel's 960-bin frame: (ebands[band] << LM) + off */
} OgCeltLeaf;

/* RFC 6716 4.5.1 redundant frames (the reference parses and drops them, decoder.rs:84-115): a 5 ms CELT frame from band 0
 * that bridges a CELT <-> SILK / hybrid switch.  w = the CELT window (120 samples), i = sample inside the record:
 *   OG_X_RED_HEAD        the redundant frame of a CELT -> SILK switch, decoded first, from the running CELT state, at the start
 *                        of its Opus frame: pcm = v for i < 120, (1 - w^2) v + w^2 pcm for 120 <= i < 240 (pcm: the SILK part)
 *   OG_X_AFTER_RED_HEAD  the hybrid frame's own CELT part behind such a frame: nothing for i < 120, pcm += w^2 v for
 *                        120 <= i < 240, pcm += v after that
 *   OG_X_RED_TAIL        the redundant frame of a SILK -> CELT switch, decoded last, from a reset CELT state, placed so that
 *                        it ends with its Opus frame: nothing for i < 120, pcm = w^2 v + (1 - w^2) pcm for 120 <= i < 240 */
enum { OG_X_NONE = 0, OG_X_RED_HEAD = 1, OG_X_AFTER_RED_HEAD = 2, OG_X_RED_TAIL = 3 };
#define OG_SPREAD_MASK 3
#define OG_X_SHIFT 4
#define OG_ROUND_SHIFT 6
/* launch rounds of one Opus frame: its SILK frames (up to three 20 ms pieces) take rounds 0-2, each round's CELT records run
 * after its SILK frames: 0 = redundant frame in front / overlap flush, 1 = the overlap flush when a redundant frame took round 0
 * (never two records of one stream in one round: they would race on its state), 2 = the frame's own CELT part, 3 = redundant
 * frame behind */
enum { OG_ROUND_HEAD = 0, OG_ROUND_FLUSH = 1, OG_ROUND_MAIN = 2, OG_ROUND_TAIL = 3, OG_ROUNDS_PER_FRAME = 4 };

typedef struct {             /* one PVQ leaf (K > 0) to expand: the unit of work of the K1 kernel */
    uint32_t leaf;           /* index into the launch's leaf array */
    uint32_t frame;          /* bits 0..29: index into the launch's OgCeltHdr array; bits 30..31: that frame's `spread`
                                (copied here by opusgpu_sort_pvq_leaves so that K1 does not have to fetch the header) */
} OgLeafRef;
#define OG_REF_FRAME_MASK 0x3FFFFFFFu
#define OG_REF_SPREAD_SHIFT 30

typedef struct {             /* one CELT frame; fixed stride */
    uint32_t stream;         /* state slot / PCM row of the stream this frame belongs to */
    uint32_t pcm_off;        /* first output sample (per channel) inside the stream's PCM row (the host parser leaves the offset
                                inside the record's Opus frame here: 0 but for OG_X_RED_TAIL; the caller adds the frame's place) */
    uint32_t leaf_begin;     /* first leaf in the batch leaf array */
    uint32_t seed;           /* LCG seed at frame entry = previous frame's final range (:2322) */
    uint16_t nleaves;
    uint8_t lm;              /* celt/decoder.rs:2233 */
    uint8_t flags;
    uint8_t start, end;      /* band range (decoder.rs:119-120) */
    uint8_t spread;          /* bits 0-1: celt/decoder.rs:1043 (decoded, used as RFC 6716 requires).  Bits 4-5: OG_X_* cross-fade of
                                this record's PCM store (redundant frames, RFC 6716 4.5.1).  Bits 6-7: which of its Opus frame's
                                four launch rounds the record belongs to (OG_ROUND_*), for a host that batches streams. */
    uint8_t intensity;       /* celt/decoder.rs:1331 */
    /* post-filter parameter sets: [0] = old, [1] = current, [2] = new (parsed in this frame, :876-901).
       gain_q = 0 means off, else gain = 0.09375 * gain_q */
    uint16_t pf_period[3];
    uint8_t pf_gain_q[3];
    uint8_t pf_tapset[3];
    uint16_t pulses_depth[OG_MAX_BANDS];     /* anti-collapse depth (1+pulses[i])/width >> LM */
    float energy[2][OG_MAX_BANDS];           /* final log2 band energy, mean removed (:948, :1461 + finalise) */
    float prev_min[2][OG_MAX_BANDS];         /* min(oldLogE, oldLogE2) per band for anti-collapse */
    OgCeltBand band[OG_MAX_BANDS];
} OgCeltHdr;

/* ---- SILK ------------------------------------------------------------------------------------ */
enum {
    OG_S_VOICED = 1,         /* silk/decoder.rs:1671-1709 */
    OG_S_INTERP = 2,         /* interpolated LPC for subframes 0-1 (silk/decoder.rs:1434-1455) */
    OG_S_INTERP4 = 4,        /* interp_factor4 (silk/decoder.rs:1435) */
    OG_S_RESET = 8,          /* channel state flushed before this frame (silk/decoder.rs:1642-1662) */
    OG_S_CODED = 16          /* channel frame present (side missing when mid-only, :2093-2099) */
};

typedef struct {             /* one 10/20 ms SILK frame of one channel (mid or side); 1024-byte stride */
    uint8_t flags;
    uint8_t order;           /* 10 | 16 (silk/decoder.rs:1727-1733) */
    uint8_t subframes;       /* 2 | 4 (silk/decoder.rs:1943-1961) */
    uint8_t sf_size;         /* 40 | 60 | 80 (silk/decoder.rs:1964-1969) */
    float ltpscale;          /* silk/decoder.rs:1752-1756 */
    float gain[4];           /* silk/decoder.rs:1718 */
    int16_t pitch_lag[4];    /* silk/decoder.rs:1503 */
    float ltp_taps[4][5];    /* silk/decoder.rs:1516 */
    float lpc[16];           /* silk/decoder.rs:1459 */
    float lpc_interp[16];    /* silk/decoder.rs:1445-1447 */
    int32_t excitation[320]; /* pre-division integer ex (silk/decoder.rs:1626-1634); f32 = ex / 2^23 */
} OgSilkChan;

enum {                       /* OgSilkFrame.flags */
    OG_SF_STEREO = 1,        /* packet codes mid + side (silk/decoder.rs:1946) */
    OG_SF_STEREO_OUT = 2,    /* stream has two output channels (silk/decoder.rs:1916) */
    OG_SF_MID_ONLY = 4,      /* side not coded in this frame (silk/decoder.rs:2005-2009) */
    OG_SF_RESAMPLER_RESET = 8,/* clear the 48 kHz resampler history first (stream start / after a CELT-only packet) */
    OG_SF_CONCEAL = 16       /* a frame with nothing to decode (in-band FEC recovery of a packet whose successor carries no LBRR
                                copy of this frame, RFC 6716 4.2.4): silence at the internal rate, through the resampler */
};

typedef struct {             /* one SILK frame (mid + side) of one stream */
    uint8_t flags;
    uint8_t bandwidth;       /* 0 NB 1 MB 2 WB */
    uint16_t f_size;         /* samples per frame at the internal rate */
    float w0, w1;            /* stereo weights (silk/decoder.rs:2000-2001) */
    float prev_w0, prev_w1;  /* the previous frame's weights, interpolated from (silk/decoder.rs:2015-2026) */
    uint32_t stream;         /* state slot / PCM row */
    uint32_t pcm_off;        /* first 48 kHz output sample (per channel) inside the stream's PCM row */
    uint32_t pad_;           /* (32-byte head: the records, their channels and the excitation arrays are 16-byte aligned) */
    OgSilkChan ch[2];
} OgSilkFrame;

#endif
