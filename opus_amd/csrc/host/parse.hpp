// Host-side parsers of the opus_amd engine: packet framing, CELT and SILK side-info emission.
#pragma once
#include <cstdint>
#include <vector>
#include "../../../include/opusgpu_sideinfo.h"
#include "rangedec.hpp"

namespace og {

// ---- packet framing: replaces Packet::from_slice (lu-zero/opus src/packet.rs:202-317) ----------
enum Mode { MODE_SILK = 0, MODE_HYBRID = 1, MODE_CELT = 2 };          // packet.rs:13-18
enum Bandwidth { BW_NB = 0, BW_MB = 1, BW_WB = 2, BW_SWB = 3, BW_FB = 4 };   // packet.rs:23-30

struct Packet {
    int mode, bandwidth, frame_size, stereo, code, nframes, padding;
    int off[48], len[48];
    int celt_end_band() const { static const int e[5] = {13, 17, 17, 19, 21}; return e[bandwidth]; }  // packet.rs:33-42
};
// 0 on success, negative OPUSGPU_* error otherwise (never aborts; packet.rs:215-225 panics instead)
int parse_packet(const uint8_t *buf, int len, Packet *p);

// RFC 6716 Appendix B self-delimiting packet at buf -> the ordinary packet with the same frames; *consumed = its size
int undelimit_packet(const uint8_t *buf, int len, std::vector<uint8_t> &out, int *consumed);

// ---- CELT: the sequential half of Celt::decode (src/celt/decoder.rs:2222-2323) ------------------
struct CeltHostState {          // what must survive between frames on the host (celt/decoder.rs:34-88)
    int channels = 2;
    float oldE[2][21], logE1[2][21], logE2[2][21];
    uint16_t pf_period = 0, pf_period_old = 0;
    uint8_t pf_gain_q = 0, pf_gain_q_old = 0, pf_tapset = 0, pf_tapset_old = 0;
    uint32_t rng = 0;
    bool pending_reset = true;
    CeltHostState() { reset(); }
    void reset();
};

// Parses one CELT frame from rd (positioned after any SILK part), fills hdr and appends its leaves.
// N = 120/240/480/960, [start,end) band range, C coded channels.  Returns 0, or <0 if the decoder
// ran past the end of the frame (corrupt data; the frame is still emitted, as the reference would).
// conceal_decay >= 0 makes it the concealment of a lost frame (no counterpart in the reference; the noise branch of libopus'
// celt_decode_lost): rd is an empty payload, the frame is not taken for silence, and the band energies are the last frame's
// lowered by conceal_decay (log2 units) instead of being read.  With no bits every band gets no pulses, so on the device the
// spectrum is the decoder's own noise filling, shaped by the decaying energies; the noise seed moves on by one LCG step.
int celt_parse_frame(CeltHostState &st, RangeDec &rd, int N, int start, int end, int C, OgCeltHdr *hdr,
                     std::vector<OgCeltLeaf> &leaves, float conceal_decay = -1.f);

// ---- SILK: the entropy half of Silk::decode (src/silk/decoder.rs:2063-2127) ---------------------
struct SilkChanHost {           // silk/decoder.rs:1296-1307 (the parse-side fields of SilkFrame)
    int log_gain = 0, coded = 0, prev_voiced = 0, previous_lag = 0;
    int16_t nlsfs[16] = {0};
    float lpc[16] = {0};
    bool flushed = true;        // device history must be cleared before the next coded frame
    // what concealment continues from (silk_conceal): the last subframe's parameters and the frame's excitation
    float last_gain = 0.f, last_taps[5] = {0}, last_lpc[16] = {0};
    int32_t last_exc[320] = {0};
    int last_lag = 0, last_n = 0;
    uint32_t plc_seed = 0;
    void flush();
};
struct SilkHostState {
    int stereo_out = 1;
    SilkChanHost mid, side;
    float w0 = 0.f, w1 = 0.f;          // stereo prediction weights (silk/decoder.rs:23-24)
    float prev_w0 = 0.f, prev_w1 = 0.f; // previous frame's weights (silk/decoder.rs:25-26, cleared by flush :1938-1939)
    bool stereo_reset = true;
    bool resampler_reset = true;        // device resampler history must be cleared (build's own stage)
    void flush();
    void reset();                       // both channels, coded or not (Decoder::flush; a side channel may hold a flush that never reached the device)
};
// Concealment of a lost Opus frame of duration_ms (10 | 20 | 40 | 60) at the bandwidth and channel coding of the last decoded one
// (no counterpart in the reference; after libopus' silk/PLC.c, simplified): appends one OgSilkFrame per 20 ms whose parameters
// continue the last frame's -- voiced: no excitation, the long-term predictor repeats the last pitch period (taps normalised into
// 0.7 .. 0.95, then attenuated); unvoiced: samples drawn at random from the last excitation, the gain attenuated; LPC bandwidth-
// expanded by 0.99 per frame -- so the device synthesises it by the regular path.  `lost` = frames concealed so far since the
// last decoded one.  Returns the number of frames; 0 when there is nothing to continue from.
int silk_conceal(SilkHostState &st, int bandwidth, int duration_ms, int stereo_pkt, int lost, std::vector<OgSilkFrame> &out);
// Parses the SILK part of one Opus frame (1-3 internal 20 ms frames); appends one OgSilkFrame each.
// Returns number of internal frames, or <0.  LBRR frames (where the reference returns Unsupported, silk/decoder.rs:2070) are
// parsed and dropped -- or, with `fec`, decoded IN PLACE of the regular frames, which are then not read: the packet's LBRR frames
// are the previous packet's audio (RFC 6716 4.2.4-4.2.7; libopus' decode_fec), recovered into the running decoder state; a 20 ms
// frame without an LBRR copy comes out as an OG_SF_CONCEAL record.
int silk_parse(SilkHostState &st, RangeDec &rd, int bandwidth, int duration_ms, int stereo_pkt,
               std::vector<OgSilkFrame> &out, bool fec = false);

}  // namespace og
