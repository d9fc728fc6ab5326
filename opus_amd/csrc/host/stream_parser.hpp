// One stream's host-side decoder state: what `struct Dec` holds in the reference (src/decoder.rs:16-30),
// minus every sample buffer (those live in HBM).
#pragma once
#include <cstddef>
#include <vector>
#include "parse.hpp"
#include "../../../include/opusgpu.h"

namespace og {

class StreamParser {
public:
    explicit StreamParser(int channels);
    void reset();                                   // Decoder::flush (decoder.rs:184-187)
    int channels() const { return channels_; }
    // Parses one packet, appending records.  Returns the number of Opus frames or a negative OPUSGPU_* code.
    int parse(const uint8_t *pkt, size_t len, OgFrameInfo *frames /*[48]*/, std::vector<OgCeltHdr> &celt,
              std::vector<OgCeltLeaf> &leaves, std::vector<OgSilkFrame> &silk);
    // In-band FEC recovery (RFC 6716 4.2.4-4.2.7; libopus' decode_fec; the reference returns Unsupported at an LBRR flag,
    // silk/decoder.rs:2070-2072): pkt is the packet FOLLOWING a lost one.  Emits the lost packet's audio -- this packet's LBRR
    // frames decoded into the running state, over this packet's duration; the CELT part of a hybrid packet, and everything of a
    // packet without LBRR frames, is concealed as after a loss.  The caller then parses pkt itself with parse().
    int parse_fec(const uint8_t *pkt, size_t len, OgFrameInfo *frames /*[48]*/, std::vector<OgCeltHdr> &celt,
                  std::vector<OgCeltLeaf> &leaves, std::vector<OgSilkFrame> &silk);

private:
    int refuse(int rc);                             // a packet refused midway: the stream starts again (stream_parser.cpp)
    int channels_;
    int prev_mode_ = -1;
    int last_packet_samples_ = 960;                 // samples per channel of the last packet (what a lost packet is taken to span)
    int last_frame_samples_ = 960;                  // ... of one Opus frame of the last packet
    int last_bandwidth_ = 4, last_stereo_ = 0;      // the last packet's TOC: what a concealed frame is coded as
    int lost_silk_ = 0, lost_celt_ = 0;             // SILK / CELT frames concealed since the last decoded packet
    bool prev_redundancy_ = false;                  // the previous frame ended in a SILK -> CELT redundant frame: CELT continues from it
    CeltHostState celt_;
    SilkHostState silk_;
};

}  // namespace og
