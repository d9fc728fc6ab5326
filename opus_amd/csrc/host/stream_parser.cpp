// Per-stream packet driver of the host layer: the sequential half of Dec::send_packet
// (lu-zero/opus src/decoder.rs:48-127).  Frames a packet, runs the range decoder over every frame and
// emits side-info records; touches no sample data and no GPU.
#include <cstring>
#include <new>
#include "parse.hpp"
#include "stream_parser.hpp"
#include "../../../include/opusgpu.h"

namespace og {

StreamParser::StreamParser(int channels) : channels_(channels)
{
    celt_.channels = channels;
    silk_.stereo_out = channels > 1;
    reset();
}

void StreamParser::reset()
{
    celt_.reset();
    silk_.reset();
    silk_.resampler_reset = true;
    prev_mode_ = -1;
    last_packet_samples_ = last_frame_samples_ = 960;
    last_bandwidth_ = 4;
    last_stereo_ = 0;
    lost_silk_ = lost_celt_ = 0;
    prev_redundancy_ = false;
}

// A packet refused after its parsing has begun (the reference returns its error from the middle of the frame as well, e.g.
// decoder.rs:104, with its state as far as it got; what follows such a packet is not defined there): the records of the packet
// are dropped by the caller, so the device has seen none of it, and the host state must not run ahead of the device's -- the
// stream starts again, as after flush(): the next frames carry the reset flags.
int StreamParser::refuse(int rc)
{
    reset();
    return rc;
}

int StreamParser::parse(const uint8_t *pkt, size_t len, OgFrameInfo *frames, std::vector<OgCeltHdr> &celt,
                        std::vector<OgCeltLeaf> &leaves, std::vector<OgSilkFrame> &silk)
{
    Packet p;
    if (len > 0x7FFFFFFF) return OPUSGPU_INVALID_DATA;
    if (len > 0) {                                      // a packet of empty frames (TOC only: DTX, RFC 6716 3.2.1) is concealed as well,
        int rc0 = parse_packet(pkt, int(len), &p);      // over the span its TOC names
        if (rc0) return rc0;
        bool empty = true;
        for (int f = 0; f < p.nframes; f++) empty = empty && p.len[f] == 0;
        if (empty) { last_packet_samples_ = p.nframes * p.frame_size; last_frame_samples_ = p.frame_size; len = 0; }
    }
    if (len == 0) {
        // A lost packet (or a DTX gap) is concealed over the span of the last packet; no counterpart in the reference (it panics
        // on an empty packet, packet.rs:215-225).  In the mode of the last packet, as libopus does it:
        //   SILK    the SILK frames continue from the last one's parameters (silk_conceal: pitch repetition / noise);
        //   CELT    noise shaped by the last band energies, which decay by 1.5 (first frame) and 0.5 per frame in log2 units;
        //   hybrid  both, the CELT part from band 17.
        // Decoder state runs on, so the next packet continues it.  Before any packet, and when SILK has nothing to continue
        // from: silent CELT frames (on the device the MDCT overlap fades out over 2.5 ms, then zeros) and SILK restarts.
        static const uint8_t kSilence[2] = {0xFF, 0xFF};
        static const int kCeltEnd[5] = {13, 17, 17, 19, 21};
        const int N = last_frame_samples_, mode = prev_mode_;
        const bool silk_ok = mode >= 0 && mode != MODE_CELT && N >= 480 && silk_.mid.coded;
        const bool celt_noise = mode == MODE_CELT || (mode == MODE_HYBRID && silk_ok);
        int n = 0;
        for (int o = 0; o < last_packet_samples_; o += N, n++) {
            OgFrameInfo &fi = frames[n];
            std::memset(&fi, 0, sizeof fi);
            fi.mode = uint8_t(mode < 0 ? MODE_CELT : mode);
            fi.bandwidth = uint8_t(last_bandwidth_);
            fi.stereo = uint8_t(last_stereo_);
            fi.frame_size = uint16_t(N);
            fi.celt_index = uint32_t(celt.size());
            fi.silk_index = uint32_t(silk.size());
            if (silk_ok) {
                fi.nsilk = uint8_t(silk_conceal(silk_, std::min(last_bandwidth_, int(BW_WB)), N / 48, last_stereo_, lost_silk_, silk));
                lost_silk_ += fi.nsilk;
            }
            if (silk_ok && mode == MODE_SILK) continue;
            for (int q = 0, M; q < N; q += M) {                            // in CELT frame sizes, largest first
                const int left = N - q;
                M = left >= 960 ? 960 : left >= 480 ? 480 : left >= 240 ? 240 : 120;
                celt.emplace_back();
                if (celt_noise) {
                    RangeDec r(kSilence, 0);
                    (void)celt_parse_frame(celt_, r, M, mode == MODE_HYBRID ? 17 : 0, kCeltEnd[last_bandwidth_], last_stereo_ + 1,
                                           &celt.back(), leaves, lost_celt_++ ? 0.5f : 1.5f);
                } else {
                    RangeDec r(kSilence, 2);
                    (void)celt_parse_frame(celt_, r, M, 0, 21, channels_, &celt.back(), leaves);
                }
                if (silk_ok) celt.back().flags |= OG_F_ACCUM;
                celt.back().pcm_off = uint32_t(q);
                celt.back().spread = uint8_t((celt.back().spread & OG_SPREAD_MASK) | OG_ROUND_MAIN << OG_ROUND_SHIFT);
                fi.has_celt++;
            }
        }
        if (!silk_ok) { silk_.flush(); silk_.resampler_reset = true; }
        prev_redundancy_ = false;
        return n;
    }
    int rc = parse_packet(pkt, int(len), &p);
    if (rc) return rc;

    // Mode transitions (decoder.rs:61-67, :79 and, where the reference parses and drops, RFC 6716 4.5): SILK restarts after
    // CELT-only packets; CELT restarts when the mode changes -- unless the previous frame ended in a SILK -> CELT redundant
    // frame, from which CELT continues -- and only after a CELT -> SILK redundant frame has been parsed from the old state.
    bool celt_restart = p.mode != prev_mode_ && prev_mode_ >= 0 && !prev_redundancy_ && p.mode != MODE_SILK;
    if (p.mode != prev_mode_ && prev_mode_ == MODE_CELT) { silk_.flush(); silk_.resampler_reset = true; }

    auto tag = [](OgCeltHdr &h, int xfade, int round) { h.spread = uint8_t((h.spread & OG_SPREAD_MASK) | xfade << OG_X_SHIFT | round << OG_ROUND_SHIFT); };
    for (int f = 0; f < p.nframes; f++) {
        RangeDec rd(pkt + p.off[f], uint32_t(p.len[f]));
        const int flen = p.len[f], N = p.frame_size;
        OgFrameInfo &fi = frames[f];
        std::memset(&fi, 0, sizeof fi);
        fi.mode = uint8_t(p.mode);
        fi.bandwidth = uint8_t(p.bandwidth);
        fi.stereo = uint8_t(p.stereo);
        fi.frame_size = uint16_t(N);
        fi.celt_index = uint32_t(celt.size());
        fi.silk_index = uint32_t(silk.size());

        if (p.mode != MODE_CELT) {                                          // decoder.rs:75-77
            int n = silk_parse(silk_, rd, std::min(p.bandwidth, int(BW_WB)), N / 48, p.stereo, silk);
            if (n < 0) return refuse(n);
            fi.nsilk = uint8_t(n);
        } else {
            silk_.flush();                                                  // decoder.rs:79
        }

        // redundancy side information (decoder.rs:82-115; RFC 6716 4.5.1)
        const int consumed = rd.tell();
        bool redundancy = false, celt_to_silk = false;
        int rbytes = 0;
        const uint8_t *rdata = nullptr;
        if (p.mode == MODE_HYBRID && consumed + 37 <= flen * 8) redundancy = rd.bit_logp(12);
        else if (p.mode == MODE_SILK && consumed + 17 <= flen * 8) redundancy = true;
        if (redundancy) {
            celt_to_silk = rd.bit_logp(1);                                  // position: 1 = the redundant frame belongs in front
            rbytes = p.mode == MODE_HYBRID ? int(rd.uint(256)) + 2 : flen - ((rd.tell() + 7) >> 3);
            if (rbytes >= flen) return refuse(OPUSGPU_INVALID_DATA);        // decoder.rs:104
            int keep = flen - rbytes;
            if (keep * 8 < rd.tell()) { keep = 0; redundancy = false; rbytes = 0; }
            else rdata = pkt + p.off[f] + keep;
            rd.set_storage(uint32_t(keep));
        }
        // The redundant frame: 5 ms of CELT from band 0 (the reference drops it, decoder.rs:84-115).  CELT -> SILK: parsed first,
        // from the running CELT state, and it opens the Opus frame.
        uint32_t red_range = 0;
        if (redundancy && celt_to_silk) {
            RangeDec r2(rdata, uint32_t(rbytes));
            celt.emplace_back();
            if (celt_parse_frame(celt_, r2, 240, 0, p.celt_end_band(), p.stereo + 1, &celt.back(), leaves) < 0) return refuse(OPUSGPU_INVALID_DATA);
            tag(celt.back(), OG_X_RED_HEAD, OG_ROUND_HEAD);
            red_range = r2.rng();
            fi.has_celt++;
        }
        if (f == 0 && celt_restart) celt_.reset();
        celt_restart = false;

        if (p.mode != MODE_SILK && N <= 960) {                              // decoder.rs:117-123
            const int start = p.mode == MODE_HYBRID ? 17 : 0;
            celt.emplace_back();
            if (celt_parse_frame(celt_, rd, N, start, p.celt_end_band(), p.stereo + 1, &celt.back(), leaves) < 0) return refuse(OPUSGPU_INVALID_DATA);
            if (p.mode == MODE_HYBRID) celt.back().flags |= OG_F_ACCUM;
            tag(celt.back(), redundancy && celt_to_silk ? OG_X_AFTER_RED_HEAD : OG_X_NONE, OG_ROUND_MAIN);
            fi.has_celt++;
        } else if (p.mode == MODE_SILK && prev_mode_ == MODE_HYBRID && !(redundancy && celt_to_silk && prev_redundancy_)) {
            // hybrid -> SILK: a silent 2.5 ms CELT frame lets the MDCT overlap of the last hybrid frame fade out (RFC 6716 4.5.2)
            static const uint8_t kSilence[2] = {0xFF, 0xFF};
            RangeDec r3(kSilence, 2);
            celt.emplace_back();
            (void)celt_parse_frame(celt_, r3, 120, 0, p.celt_end_band(), p.stereo + 1, &celt.back(), leaves);     // (a silent frame: cannot fail)
            celt.back().flags |= OG_F_ACCUM;
            // One record per stream per round (the tail kernel reads and writes the stream's state): behind a redundant frame in
            // front (round HEAD) the flush takes the round after it and stores no PCM -- its 2.5 ms are the redundant frame's
            // (orc_opus.c: the flush is added, then overwritten); the CELT state runs redundant frame -> flush, as in the oracle.
            if (redundancy && celt_to_silk) tag(celt.back(), OG_X_RED_TAIL /* a 120-sample record of this kind stores nothing */, OG_ROUND_FLUSH);
            else tag(celt.back(), OG_X_NONE, OG_ROUND_HEAD);
            fi.has_celt++;
        }
        if (redundancy && !celt_to_silk) {
            // SILK -> CELT: parsed last, from a fresh CELT state, which the next (CELT) packet continues; it closes the Opus frame
            RangeDec r2(rdata, uint32_t(rbytes));
            celt_.reset();
            celt.emplace_back();
            if (celt_parse_frame(celt_, r2, 240, 0, p.celt_end_band(), p.stereo + 1, &celt.back(), leaves) < 0) return refuse(OPUSGPU_INVALID_DATA);
            celt.back().pcm_off = uint32_t(N - 240);
            tag(celt.back(), OG_X_RED_TAIL, OG_ROUND_TAIL);
            red_range = r2.rng();
            fi.has_celt++;
        }
        fi.final_range = rd.rng() ^ red_range;                              // RFC 6716 4.5.1.4
        fi.tell_bits = uint32_t(rd.tell());
        fi.budget_bits = uint32_t(rd.storage_bits());
        prev_mode_ = p.mode;
        prev_redundancy_ = redundancy && !celt_to_silk;
    }
    prev_mode_ = p.mode;
    last_packet_samples_ = p.nframes * p.frame_size;
    last_frame_samples_ = p.frame_size;
    last_bandwidth_ = p.bandwidth;
    last_stereo_ = p.stereo;
    lost_silk_ = lost_celt_ = 0;
    return p.nframes;
}

int StreamParser::parse_fec(const uint8_t *pkt, size_t len, OgFrameInfo *frames, std::vector<OgCeltHdr> &celt,
                            std::vector<OgCeltLeaf> &leaves, std::vector<OgSilkFrame> &silk)
{
    Packet p;
    if (len == 0 || len > 0x7FFFFFFF) return parse(nullptr, 0, frames, celt, leaves, silk);
    int rc = parse_packet(pkt, int(len), &p);
    if (rc) return rc;
    bool any = false;                                   // does any frame of the packet carry LBRR frames?  (the flag behind the VAD flags)
    for (int f = 0; f < p.nframes && p.mode != MODE_CELT; f++) {
        if (p.len[f] == 0) continue;
        RangeDec rd(pkt + p.off[f], uint32_t(p.len[f]));
        const int nsf = p.frame_size <= 960 ? 1 : p.frame_size / 960;
        for (int i = 0; i < nsf; i++) (void)rd.bit_logp(1);
        bool a = rd.bit_logp(1);
        if (p.stereo) {
            for (int i = 0; i < nsf; i++) (void)rd.bit_logp(1);
            a = rd.bit_logp(1) || a;
        }
        any = any || a;
    }
    if (!any) {                                         // nothing to recover from: the span of this packet is concealed
        last_packet_samples_ = p.nframes * p.frame_size;
        last_frame_samples_ = p.frame_size;
        return parse(nullptr, 0, frames, celt, leaves, silk);
    }
    // as a packet of this mode would start (see parse()); no redundancy information is available for the lost packet
    if (p.mode != prev_mode_ && prev_mode_ == MODE_CELT) { silk_.flush(); silk_.resampler_reset = true; }
    if (p.mode != prev_mode_ && prev_mode_ >= 0 && !prev_redundancy_ && p.mode != MODE_SILK) celt_.reset();
    static const uint8_t kSilence[2] = {0xFF, 0xFF};
    for (int f = 0; f < p.nframes; f++) {
        const int N = p.frame_size;
        OgFrameInfo &fi = frames[f];
        std::memset(&fi, 0, sizeof fi);
        fi.mode = uint8_t(p.mode);
        fi.bandwidth = uint8_t(p.bandwidth);
        fi.stereo = uint8_t(p.stereo);
        fi.frame_size = uint16_t(N);
        fi.celt_index = uint32_t(celt.size());
        fi.silk_index = uint32_t(silk.size());
        RangeDec rd(pkt + p.off[f], uint32_t(p.len[f]));
        const int n = silk_parse(silk_, rd, std::min(p.bandwidth, int(BW_WB)), N / 48, p.stereo, silk, true);
        if (n < 0) return refuse(n);
        fi.nsilk = uint8_t(n);
        if (p.mode == MODE_HYBRID || prev_mode_ != MODE_SILK) {
            // the CELT part of a hybrid frame is not in the LBRR copy, and after CELT / hybrid audio the MDCT overlap has to fade
            // out: silent CELT frames over the span, added onto the SILK output (as after a loss)
            for (int o = 0, M; o < N; o += M) {
                M = N - o >= 960 ? 960 : N - o >= 480 ? 480 : N - o >= 240 ? 240 : 120;
                RangeDec r(kSilence, 2);
                celt.emplace_back();
                (void)celt_parse_frame(celt_, r, M, 0, 21, channels_, &celt.back(), leaves);
                celt.back().flags |= OG_F_ACCUM;
                celt.back().pcm_off = uint32_t(o);
                celt.back().spread = uint8_t((celt.back().spread & OG_SPREAD_MASK) | OG_ROUND_MAIN << OG_ROUND_SHIFT);
                fi.has_celt++;
            }
        }
        fi.tell_bits = uint32_t(rd.tell());
        fi.budget_bits = uint32_t(rd.storage_bits());
        prev_mode_ = p.mode;
    }
    prev_redundancy_ = false;
    last_packet_samples_ = p.nframes * p.frame_size;
    last_frame_samples_ = p.frame_size;
    last_bandwidth_ = p.bandwidth;
    last_stereo_ = p.stereo;
    lost_silk_ = lost_celt_ = 0;
    return p.nframes;
}

}  // namespace og