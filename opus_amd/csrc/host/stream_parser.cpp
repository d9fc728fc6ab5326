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
    silk_.flush();
    silk_.resampler_reset = true;
    prev_mode_ = -1;
}

int StreamParser::parse(const uint8_t *pkt, size_t len, OgFrameInfo *frames, std::vector<OgCeltHdr> &celt,
                        std::vector<OgCeltLeaf> &leaves, std::vector<OgSilkFrame> &silk)
{
    Packet p;
    if (len > 0x7FFFFFFF) return OPUSGPU_INVALID_DATA;
    int rc = parse_packet(pkt, int(len), &p);
    if (rc) return rc;

    // mode transitions: decoder.rs:61-67, :79 (silk.flush() on CELT packets); CELT state restarts when
    // CELT (re)joins after a SILK-only packet (RFC 6716 4.5.2; the reference keeps stale state)
    if (p.mode != prev_mode_ && prev_mode_ >= 0) {
        if (p.mode != MODE_SILK) celt_.reset();
        if (prev_mode_ == MODE_CELT) { silk_.flush(); silk_.resampler_reset = true; }
    }

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
            if (n < 0) return n;
            fi.nsilk = uint8_t(n);
        } else {
            silk_.flush();                                                  // decoder.rs:79
        }

        // redundancy side information: parsed, then only used to shorten the CELT budget
        // (decoder.rs:82-115; RFC 6716 4.5.1).  The 5 ms redundant frame itself is not decoded.
        const int consumed = rd.tell();
        bool redundancy = false;
        if (p.mode == MODE_HYBRID && consumed + 37 <= flen * 8) redundancy = rd.bit_logp(12);
        else if (p.mode == MODE_SILK && consumed + 17 <= flen * 8) redundancy = true;
        if (redundancy) {
            (void)rd.bit_logp(1);                                           // celt_to_silk position
            int rbytes = p.mode == MODE_HYBRID ? int(rd.uint(256)) + 2 : flen - ((rd.tell() + 7) >> 3);
            if (rbytes >= flen) return OPUSGPU_INVALID_DATA;                // decoder.rs:104
            int keep = flen - rbytes;
            if (keep * 8 < rd.tell()) keep = 0;
            rd.set_storage(uint32_t(keep));
        }

        if (p.mode != MODE_SILK && N <= 960) {                              // decoder.rs:117-123
            const int start = p.mode == MODE_HYBRID ? 17 : 0;
            celt.emplace_back();
            (void)celt_parse_frame(celt_, rd, N, start, p.celt_end_band(), p.stereo + 1, &celt.back(), leaves);
            if (p.mode == MODE_HYBRID) celt.back().flags |= OG_F_ACCUM;
            fi.has_celt = 1;
        }
        fi.final_range = rd.rng();
        fi.tell_bits = uint32_t(rd.tell());
        fi.budget_bits = uint32_t(rd.storage_bits());
    }
    prev_mode_ = p.mode;
    return p.nframes;
}

}  // namespace og
