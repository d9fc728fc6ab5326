/* opusgpu.h -- C ABI of the opus_amd batched Opus decode engine (MI355X / gfx950).
 *
 * This is the drop-in boundary for the decode path of lu-zero/opus.  The reference exposes one object,
 * `OPUS_DESCR: &dyn Descriptor` (src/decoder.rs:190-197), whose `create()` returns a rust-av
 * `Decoder` with five methods (src/decoder.rs:42-188).  A Rust `impl Decoder` would bind the functions
 * below one to one (INTEGRATION.md shows the stub); every function cites the method it replaces.
 *
 * Three layers, all plain C (pointers + sizes, no C++/torch types):
 *   1. opusgpu_decoder_*  one stream, same call order as the reference:
 *                         set_extradata -> configure -> (send_packet -> receive_frame)* -> flush
 *   2. opusgpu_batch_*    thousands of independent streams per call (what north_star adds)
 *   3. opusgpu_parse_* / opusgpu_launch_*   the thin inner FFI: host side-info emission and the raw
 *                         kernel launches on caller-owned device memory (no allocation inside)
 *
 * Errors mirror av_codec::error::Error as used by the reference (packet.rs:84,130,145,163,192;
 * decoder.rs:104,160,167; silk/decoder.rs:2071).  Nothing here aborts or panics.
 */
#ifndef OPUSGPU_H
#define OPUSGPU_H
#include <stddef.h>
#include <stdint.h>
#include "opusgpu_sideinfo.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    OPUSGPU_OK = 0,
    OPUSGPU_INVALID_DATA = -1,          /* Error::InvalidData */
    OPUSGPU_CONFIG_INVALID = -2,        /* Error::ConfigurationInvalid (decoder.rs:160) */
    OPUSGPU_CONFIG_INCOMPLETE = -3,     /* Error::ConfigurationIncomplete (decoder.rs:167) */
    OPUSGPU_UNSUPPORTED = -4,           /* Error::Unsupported (silk/decoder.rs:2071: LBRR; >2 channels decoder.rs:171) */
    OPUSGPU_MORE_DATA_NEEDED = -5,      /* Error::MoreDataNeeded (decoder.rs:129) */
    OPUSGPU_DEVICE = -6,                /* HIP runtime error / no GPU; opusgpu_last_error() has the text */
    OPUSGPU_BAD_ARG = -7
};

#define OPUSGPU_MAX_FRAME 5760          /* 120 ms @ 48 kHz: the longest packet (RFC 6716 3.2.5) */

const char *opusgpu_version(void);
const char *opusgpu_last_error(void);   /* thread-local text of the last OPUSGPU_DEVICE failure */
const char *opusgpu_strerror(int code);

/* ------------------------------------------------------------------------------------------------
 * Layer 1: one stream.  Replaces `struct Dec` + `impl Decoder for Dec` (src/decoder.rs:16-188).
 * Not thread-safe per handle, exactly as the reference's `&mut self` methods. */
typedef struct opusgpu_decoder opusgpu_decoder;

/* Descriptor::describe (decoder.rs:37-39) -> &Descr{codec, name, desc, mime} of OPUS_DESCR (decoder.rs:190-197).  The
 * strings are static; codec / name / mime are the reference's ("opus", "opus", "audio/OPUS"), desc names this engine. */
typedef struct {
    const char *codec, *name, *desc, *mime;
} OgDescr;
const OgDescr *opusgpu_describe(void);

/* Descriptor::create (decoder.rs:33-35).  device = HIP device ordinal. */
int opusgpu_decoder_create(opusgpu_decoder **out, int device);
void opusgpu_decoder_destroy(opusgpu_decoder *);
/* Decoder::set_extradata (decoder.rs:45-47): the OpusHead blob, copied. */
int opusgpu_decoder_set_extradata(opusgpu_decoder *, const uint8_t *extra, size_t len);
/* Decoder::configure (decoder.rs:133-182): channels from byte 9; no extradata -> 2 channels.  Mapping families 1 and 255
 * (where the reference hits unimplemented!(), decoder.rs:151-157, :170-171) configure a multistream decoder: send_packet
 * then takes RFC 7845 5.1.1 multistream packets and receive_frame returns `channels` interleaved channels. */
int opusgpu_decoder_configure(opusgpu_decoder *);
/* Decoder::send_packet (decoder.rs:48-127): parse, decode on the GPU, queue the PCM.  len == 0 (pkt may be NULL) reports a
 * lost packet: the span of the last packet is concealed in that packet's mode -- SILK frames continued from the last frame's
 * parameters (pitch repetition / noise, attenuated), CELT noise under band energies that decay from frame to frame, both for
 * hybrid -- and the decoder state runs on into the next packet; before any packet: silence.  The reference has no concealment
 * and panics on an empty packet (packet.rs:215-225). */
int opusgpu_decoder_send_packet(opusgpu_decoder *, const uint8_t *pkt, size_t len);
/* In-band FEC recovery (RFC 6716 4.2.4-4.2.7; libopus' decode_fec) -- the reference returns Error::Unsupported at the LBRR
 * flag (silk/decoder.rs:2070-2072): when a packet was lost and the NEXT one has arrived, send the next one here first.  Queues
 * the lost packet's audio over the next packet's duration -- its LBRR frames (low-rate copies of the lost packet's SILK frames)
 * decoded into the running decoder state; frames without a copy, the CELT part of a hybrid packet, and all of a CELT-only
 * packet or one without LBRR frames are concealed as by send_packet(len 0) -- then call send_packet with the same packet. */
int opusgpu_decoder_send_packet_fec(opusgpu_decoder *, const uint8_t *pkt, size_t len);
/* Decoder::receive_frame (decoder.rs:128-132, unimplemented!() in the reference): pops the oldest
 * decoded packet as interleaved f32 @ 48 kHz.  OPUSGPU_MORE_DATA_NEEDED when the queue is empty. */
int opusgpu_decoder_receive_frame(opusgpu_decoder *, float *pcm, size_t cap_floats, size_t *nsamples, int *channels);
/* Decoder::flush (decoder.rs:184-187, unimplemented!() in the reference): drop queued output and
 * reset all decoder state (Silk::flush silk/decoder.rs:1934-1940 + CELT reset). */
int opusgpu_decoder_flush(opusgpu_decoder *);

/* OpusHead (RFC 7845 5.1) as Dec::configure reads it (decoder.rs:133-182: channels byte 9, output gain bytes 16-17,
 * mapping family byte 18, stream / coupled counts and channel mapping behind them).  Family 0 fills the implicit
 * mono / stereo mapping (decoder.rs:158-165).  Returns OPUSGPU_CONFIG_INVALID for a malformed header,
 * OPUSGPU_CONFIG_INCOMPLETE when a family != 0 header is cut short (decoder.rs:167), OPUSGPU_UNSUPPORTED for a
 * major version other than 0. */
typedef struct {
    uint8_t version, channels;
    uint16_t pre_skip;                  /* 48 kHz samples to drop at the start of the stream */
    uint32_t input_sample_rate;         /* informational (the RFC vectors carry 44100) */
    int16_t output_gain_q8;             /* dB in Q7.8, to be applied to the decoded PCM */
    uint8_t mapping_family, stream_count, coupled_count;
    uint8_t mapping[255];
} OgOpusHead;
int opusgpu_opus_head_parse(const uint8_t *extra, size_t len, OgOpusHead *out);
/* Layer-1 decoders apply what the header asks for, which the reference reads and then ignores (decoder.rs:137-150):
 * receive_frame drops the first pre_skip samples of the stream and scales the PCM by 10^(output_gain_q8 / 5120).
 * Both are no-ops for the RFC 6716 vectors (pre-skip 0, gain 0). */

/* ------------------------------------------------------------------------------------------------
 * Container layer: the packets of one Opus track out of a Matroska (.mka/.mkv/.webm) or Ogg (.opus) file image.
 * Stands in for the `matroska` + `av-format` crates the reference's integration tests use to feed send_packet
 * (decoder.rs:202-232).  The handle copies what it needs; `data` may be released after open.  Matroska: first
 * A_OPUS track, CodecPrivate = OpusHead, SimpleBlocks and BlockGroups, all four lacing modes, unknown-size
 * Segment / Cluster.  Ogg: first logical stream that starts with OpusHead, page CRCs verified, packets spanning pages
 * reassembled, OpusTags skipped.  pts = nanoseconds (Matroska) or the granule position of the page the packet ends
 * on, -1 for the others (Ogg). */
typedef struct opusgpu_demux opusgpu_demux;
int opusgpu_demux_open(opusgpu_demux **out, const uint8_t *data, size_t len);
void opusgpu_demux_close(opusgpu_demux *);
int opusgpu_demux_info(const opusgpu_demux *, int *container /* 0 Matroska, 1 Ogg */, const uint8_t **extradata,
                       size_t *extralen, uint32_t *npackets);
/* Packet i (valid until close).  OPUSGPU_MORE_DATA_NEEDED past the end. */
int opusgpu_demux_packet(const opusgpu_demux *, uint32_t i, const uint8_t **pkt, size_t *len, int64_t *pts);

/* ------------------------------------------------------------------------------------------------
 * Layer 2: a batch of independent streams on one GPU (one handle per GPU; streams shard across GPUs
 * with no collective).  Each stream is a `Dec` of the reference. */
typedef struct opusgpu_batch opusgpu_batch;

int opusgpu_batch_create(opusgpu_batch **out, uint32_t nstreams, int channels, int device);
void opusgpu_batch_destroy(opusgpu_batch *);
/* Host half of send_packet for n (stream, packet) pairs: framing + range decoding + side-info
 * emission (packet.rs, entropy.rs, the integer control of celt/decoder.rs and silk/decoder.rs).
 * A stream may appear at most once per call.  errors[i] (optional) receives the per-packet code: a
 * bad packet never poisons the batch.  lens[i] == 0 reports that stream's packet as lost (concealment, as in
 * opusgpu_decoder_send_packet).  Returns OPUSGPU_OK or a global failure. */
int opusgpu_batch_send(opusgpu_batch *, const uint8_t *const *pkts, const uint32_t *lens,
                       const uint32_t *stream_ids, uint32_t n, int32_t *errors);
/* The same for `packets_per_stream` consecutive packets of each of `nstreams` streams in one call (packet t of the i-th stream
 * of the call is pkts[i * packets_per_stream + t], errors likewise): the stream's PCM row then holds their output back to
 * back (the rows grow to packets_per_stream x OPUSGPU_MAX_FRAME samples on first use; batch_pcm's pointer and stride change
 * with that).  When every packet of a send (this call or opusgpu_batch_send) holds the same number of frames, all of them
 * CELT-only -- no SILK part, no redundant frame -- and there is more than one frame per stream, batch_decode issues the whole
 * send as ONE launch in which each stream's frames are walked in time order by one workgroup (the layer-3b
 * frames_per_group), instead of one round of launches per frame.  No counterpart in the reference (one packet per call). */
int opusgpu_batch_send_multi(opusgpu_batch *, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                             uint32_t nstreams, uint32_t packets_per_stream, int32_t *errors);
/* opusgpu_decoder_send_packet_fec for a batch: every packet of the call is the successor of a lost packet of its stream; the
 * send holds the lost packets' audio (decode it, read it, then send the same packets with opusgpu_batch_send). */
int opusgpu_batch_send_fec(opusgpu_batch *, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                           uint32_t n, int32_t *errors);
/* Device half: H2D of the side-info, then the kernels, on the batch's HIP stream.  Asynchronous: the next batch_send
 * may follow at once (it waits for this call's H2D copies, not for its kernels, before it rewrites the staging arrays);
 * batch_pcm / batch_read results are valid after batch_sync (batch_read synchronises itself). */
int opusgpu_batch_decode(opusgpu_batch *);
int opusgpu_batch_sync(opusgpu_batch *);
/* Device-resident result of the last decode: interleaved f32, row s = stream s,
 * row stride *stride_floats, nsamples[s] valid samples per channel (host array owned by the batch,
 * valid until the next batch_send). */
int opusgpu_batch_pcm(opusgpu_batch *, const float **d_pcm, size_t *stride_floats, const uint32_t **nsamples);
/* Copy stream s's result to the host (synchronises). */
int opusgpu_batch_read(opusgpu_batch *, uint32_t stream, float *pcm, size_t cap_floats, size_t *nsamples);
int opusgpu_batch_flush(opusgpu_batch *, uint32_t stream);
/* Side-info the last batch_send produced (host memory owned by the batch; parity tests read it). */
int opusgpu_batch_sideinfo(opusgpu_batch *, const OgCeltHdr **celt, uint32_t *ncelt, const OgCeltLeaf **leaves,
                           uint32_t *nleaves, const OgSilkFrame **silk, uint32_t *nsilk);
/* The launch plan of that send (host memory owned by the batch, valid until the next batch_send): what batch_decode does with
 * the side-info, for callers that keep several sends' side-info resident and issue the layer-3 launches themselves (bench.py's
 * mixed-mode workload).  SILK frames are ordered by depth -- launch d = frames [silk_depth_off[d], silk_depth_off[d + 1]), the
 * d-th SILK frame of every stream that has one; all of them run before the CELT launch, whose records are the whole `celt`
 * array, stream run r = [run_off[r], run_off[r + 1]) (run_len != 0: every run has that many records). */
typedef struct {
    const OgLeafRef *refs;         /* K1 work list over all CELT records (opusgpu_sort_pvq_leaves' order) */
    uint32_t nrefs, class_counts[4];
    const uint32_t *run_off;       /* [nruns + 1] */
    uint32_t nruns, run_len;
    const uint32_t *silk_depth_off;/* [ndepths + 1] */
    uint32_t ndepths;
} OgBatchPlan;
int opusgpu_batch_plan(opusgpu_batch *, OgBatchPlan *out);

/* ------------------------------------------------------------------------------------------------
 * Layer 3a: host parser for one stream, no GPU involved.  What the reference's Rust host would keep
 * (north_star): Packet::from_slice (packet.rs:202-317), RangeDecoder (entropy.rs:264-543),
 * Celt::decode's parse half (celt/decoder.rs:2222-2323), Silk::decode's parse half
 * (silk/decoder.rs:2063-2127) and the send_packet driver (decoder.rs:48-127). */
typedef struct opusgpu_parser opusgpu_parser;

typedef struct {               /* one Opus frame of a packet */
    uint8_t mode;              /* 0 SILK, 1 hybrid, 2 CELT (packet.rs:13-18) */
    uint8_t bandwidth;         /* 0 NB .. 4 FB (packet.rs:23-30) */
    uint8_t stereo;            /* packet stereo flag */
    uint8_t nsilk;             /* SILK frames emitted for this Opus frame (0-3) */
    uint16_t frame_size;       /* samples per channel @ 48 kHz */
    uint16_t has_celt;         /* CELT records of this frame (0-3), consecutive from celt_index in decode order: a redundant
                                  frame in front / an overlap flush, the frame's own CELT part, a redundant frame behind
                                  (OgCeltHdr.spread holds each one's launch round and cross-fade) */
    uint32_t celt_index;       /* index into the celt[] output */
    uint32_t silk_index;       /* index of the first SILK frame in silk[] */
    uint32_t final_range;      /* range decoder state after the frame (celt/decoder.rs:2322) */
    uint32_t tell_bits;        /* bits the range decoder had consumed at the end of the frame (entropy.rs:375-378 tell) */
    uint32_t budget_bits;      /* bits the frame held for it (8 * frame bytes, less a redundant frame's): a parse that
                                  follows the bitstream never ends with tell_bits > budget_bits */
} OgFrameInfo;

int opusgpu_parser_create(opusgpu_parser **out, int channels);
void opusgpu_parser_destroy(opusgpu_parser *);
void opusgpu_parser_reset(opusgpu_parser *);
/* Parses one packet.  Appends to the caller's arrays (capacities in, counts in/out through the
 * n* pointers).  Returns the number of Opus frames (1-48) or a negative error. */
int opusgpu_parser_parse(opusgpu_parser *, const uint8_t *pkt, size_t len, OgFrameInfo *frames /* [48] */,
                         OgCeltHdr *celt, uint32_t celt_cap, uint32_t *ncelt,
                         OgCeltLeaf *leaves, uint32_t leaf_cap, uint32_t *nleaves,
                         OgSilkFrame *silk, uint32_t silk_cap, uint32_t *nsilk);
/* The same for in-band FEC recovery (opusgpu_decoder_send_packet_fec): pkt follows a lost packet; the records are the lost
 * packet's audio as far as pkt's LBRR frames hold it. */
int opusgpu_parser_parse_fec(opusgpu_parser *, const uint8_t *pkt, size_t len, OgFrameInfo *frames /* [48] */,
                             OgCeltHdr *celt, uint32_t celt_cap, uint32_t *ncelt,
                             OgCeltLeaf *leaves, uint32_t leaf_cap, uint32_t *nleaves,
                             OgSilkFrame *silk, uint32_t silk_cap, uint32_t *nsilk);
/* Packet::from_slice alone (packet.rs:202-317). */
typedef struct {
    int32_t mode, bandwidth, frame_size, stereo, code, nframes, padding;
    int32_t off[48], len[48];
} OgPacket;
int opusgpu_packet_parse(const uint8_t *pkt, size_t len, OgPacket *out);
/* RFC 6716 Appendix B: the self-delimiting packet at pkt (every stream but the last of a multistream packet, RFC 7845 5.1.1;
 * no counterpart in the reference, whose multistream configuration ends in unimplemented!(), decoder.rs:151-157) rewritten
 * as the ordinary packet with the same frames: out[0 .. *out_len), and *consumed = the bytes it took at pkt. */
int opusgpu_packet_undelimit(const uint8_t *pkt, size_t len, uint8_t *out, size_t out_cap, size_t *out_len, int *consumed);

/* ------------------------------------------------------------------------------------------------
 * Layer 3b: raw kernel launches on caller-owned device memory.  `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  No allocation, no synchronisation.  All pointers are device
 * pointers unless noted. */

typedef struct {               /* per-stream CELT synthesis state in HBM (celt/decoder.rs:34-43 `buf`,
                                  `deemph_coeff`; the reference never fills them) */
    float hist[2][1024];       /* ring of post-filtered, pre-de-emphasis output; index = sample & 1023 */
    float overlap[2][60];      /* IMDCT tail awaiting the next frame's TDAC mirror */
    float deemph[2];
    uint32_t pos;              /* absolute sample position of the next frame (ring phase) */
    uint32_t pad_;
} OgCeltState;

typedef struct {               /* optional stage dumps for parity tests; any pointer may be NULL; zero-fill them first */
    int16_t *pulses;           /* [nframes][2][960]  PVQ pulse vectors at their coefficient positions */
    float *norm;               /* [nframes][2][960]  normalised coefficients after band decoding */
    float *freq;               /* [nframes][2][960]  denormalised MDCT coefficients */
    uint8_t *collapse;         /* [nframes][2][21]   collapse masks */
    uint64_t *cycles;          /* [nframes][64] + [177*4096][4]: shader-clock stamps, K2/K3 stages per frame at 0-21, then K1 stages
                                  per workgroup at ((NMAX*4096 + block)*4) (profiling aid, tools/stage_profile.py) */
} OgCeltDebug;

/* One-time per-device set-up of the constant tables (twiddles, window, PVQ U table). Called lazily by
 * the launches too; exposed so a caller can pay the allocation before graph capture. */
int opusgpu_init(int device);

/* CELT, four kernel stages per launch:
 *   K1  og_pvq_all_kernel (one launch; its workgroups each serve one leaf size class): one lane per PVQ leaf (refs[], sorted by size
 *       across the whole batch so that the lanes of a wavefront run the same trip counts): CWRS index -> pulses, normalise, spreading
 *       rotation, collapse mask (celt/decoder.rs:541-786); writes the shaped coefficients to the `work` scratch.
 *   plan  og_plan_kernel: one lane per frame: the integer half of decode_bands (celt/decoder.rs:2037-2220): which
 *       zero-pulse leaves fold / make noise and with which seed, collapse masks, dependency levels of the bands.
 *   K2  og_celt_kernel ("spectrum"): one workgroup per frame, stateless: fold / noise fill, Haar + Hadamard reordering,
 *       stereo merge (celt/decoder.rs:453-538, :788-827, :1601-1663, :1994-2032), then the first part of the synthesis
 *       the reference lacks (anti-collapse, denormalise, IMDCT up to its post-twiddle); leaves the frame's
 *       time-domain samples (before the window) in the frame's `work` row.
 *   K3  og_celt_tail_kernel: one workgroup per group of frames, in stream order, carrying the decoder state: TDAC window
 *       + overlap-add, comb post-filter, de-emphasis (RFC 6716 4.3.7), PCM out, state write-back.
 * Frames [i*frames_per_group, (i+1)*frames_per_group) must belong to one stream, in time order.
 * `work` = opusgpu_celt_work_bytes(nframes) bytes of device scratch (one block per frame, so a buffer sized for the
 * largest launch serves smaller ones), zero-filled ONCE by the caller; the kernels hand the parts that must start at
 * zero back zero-filled.  Launches that share `work` must be ordered on one stream. */
size_t opusgpu_celt_work_bytes(uint32_t nframes);
/* Host helper: lists the PVQ leaves (K > 0) of hdr[0..nframes) in decreasing size N and counts them per size class
 * (N > 64, 24 < N <= 64, 8 < N <= 24, N <= 8): every K1 workgroup serves one class, so that the many small leaves do not
 * pay the LDS of the few large ones.  Returns the total (<= cap; 0 if cap is too small). */
uint32_t opusgpu_sort_pvq_leaves(const OgCeltHdr *hdr, uint32_t nframes, const OgCeltLeaf *leaves, OgLeafRef *out, uint32_t cap,
                                 uint32_t class_counts[4]);
int opusgpu_launch_celt(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                        const OgLeafRef *refs, const uint32_t class_counts[4], void *work, OgCeltState *state, float *pcm,
                        size_t pcm_stride_floats, const OgCeltDebug *debug /* host struct, may be NULL */, void *stream);
/* The same launch in its two halves, for callers that pipeline consecutive batches: the first half (K1 + plan) only
 * writes `work` and touches no decoder state, so batch k+1's first half may run on another stream -- with its own
 * `work` buffer -- while batch k's frame kernel is still running; the frame kernels of a stream's batches must stay in
 * order (they carry the decoder state).  opusgpu_launch_celt = both halves on one stream. */
int opusgpu_launch_celt_pvq(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, const OgLeafRef *refs,
                            const uint32_t class_counts[4], void *work, const OgCeltDebug *debug, void *stream);
int opusgpu_launch_celt_frames(const OgCeltHdr *hdr, uint32_t nframes, uint32_t frames_per_group, void *work, OgCeltState *state,
                               float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream);
/* opusgpu_launch_celt_frames as its two kernels (the thin per-kernel FFI of SURVEY 8(b): ..._launch_band / _celt_synth):
 * K2 alone touches no decoder state, so any number of a stream's frames may be in one K2 launch; K3 must follow it on
 * the same stream (or behind an event).  bench.py times K2, the dominant kernel, through this entry point. */
int opusgpu_launch_celt_spectrum(const OgCeltHdr *hdr, uint32_t nframes, void *work, const OgCeltDebug *debug, void *stream);
int opusgpu_launch_celt_tail(const OgCeltHdr *hdr, uint32_t nframes, uint32_t frames_per_group, void *work, OgCeltState *state,
                             float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream);
/* The same with groups of different sizes -- a mixed send, where the number of CELT records (frames of a packet, redundant
 * frames, overlap flushes: everything decoder.rs:72-123 walks for one packet, for every packet of the send) differs from stream
 * to stream: run r = records [run_off[r], run_off[r + 1]) of ONE stream, in the order they must run; run_off is a device
 * array of nruns + 1 offsets.  K1, the plan and K2 are stateless, so all records of a send go through them in one launch
 * whatever their order in time; only the tail walks a stream's records one after the other. */
int opusgpu_launch_celt_tail_runs(const OgCeltHdr *hdr, uint32_t nframes, const uint32_t *run_off, uint32_t nruns, void *work,
                                  OgCeltState *state, float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream);
int opusgpu_launch_celt_runs(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, const uint32_t *run_off, uint32_t nruns,
                             const OgLeafRef *refs, const uint32_t class_counts[4], void *work, OgCeltState *state, float *pcm,
                             size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream);

/* opusgpu_launch_celt for callers that cannot allocate device memory or bring a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (which has no device at
 * all).  The `work` scratch is a static __device__ array inside the library (7.6 MiB per GPU, held until the process
 * ends) instead of an argument; inputs, state and outputs are still the caller's.  Limits: nframes <=
 * opusgpu_noalloc_max_frames() (256) per call, else OPUSGPU_BAD_ARG with opusgpu_last_error() set; opusgpu_init(device)
 * must have run once on the device (that call makes the library's only allocation, the constant tables).  Every call on
 * a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams, two threads) race -- order them on
 * one stream or behind events.  Calls hipMemsetAsync + the four kernels on `stream`; no hipMalloc / hipFree, no
 * synchronisation. */
uint32_t opusgpu_noalloc_max_frames(void);
int opusgpu_launch_celt_noalloc(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                                const OgLeafRef *refs, const uint32_t class_counts[4], OgCeltState *state, float *pcm,
                                size_t pcm_stride_floats, void *stream);

/* SILK: K4 LTP + LPC synthesis with one lane per (stream, channel) over struct-of-arrays state (replaces
 * silk/decoder.rs:1775-1898), then K5 + K6: stereo unmix / mono copy (silk/decoder.rs:2012-2061, :2104-2113) and the
 * internal-rate -> 48 kHz resampler the reference lacks, writing interleaved PCM at OgSilkFrame.pcm_off.
 * `state` = opusgpu_silk_state_bytes(nstreams) bytes of zero-initialised device memory, layout private to the
 * kernels.  A stream may appear at most once per launch (its frames are sequential in time). */
size_t opusgpu_silk_state_bytes(uint32_t nstreams);
int opusgpu_launch_silk(const OgSilkFrame *frames, uint32_t nframes, void *state, uint32_t nstreams, float *pcm,
                        size_t pcm_stride_floats, float *debug_internal /* [nframes][2][320] internal-rate L/R, or NULL */,
                        void *stream);

/* Bytes of OgCeltState[] / PCM a caller must provide are sizeof-derived; helpers for FFI users: */
size_t opusgpu_sizeof_celt_state(void);
size_t opusgpu_sizeof_celt_hdr(void);
size_t opusgpu_sizeof_celt_leaf(void);
size_t opusgpu_sizeof_silk_frame(void);

#ifdef __cplusplus
}
#endif
#endif
