// C ABI, layer 3a (include/opusgpu.h): host parser entry points.  No GPU calls in this file.
#include <cstring>
#include <new>
#include <vector>
#include "stream_parser.hpp"

struct opusgpu_parser {
    og::StreamParser sp;
    std::vector<OgCeltHdr> celt;
    std::vector<OgCeltLeaf> leaves;
    std::vector<OgSilkFrame> silk;
    explicit opusgpu_parser(int ch) : sp(ch) {}
};

extern "C" {

const char *opusgpu_version(void) { return "opus_amd 0.1 (gfx950)"; }

// Descriptor::describe (decoder.rs:37-39) of OPUS_DESCR (decoder.rs:190-197)
const OgDescr *opusgpu_describe(void)
{
    static const OgDescr d = {"opus", "opus", "opus_amd batched opus decoder (MI355X / gfx950)", "audio/OPUS"};
    return &d;
}

const char *opusgpu_strerror(int code)
{
    switch (code) {
    case OPUSGPU_OK: return "ok";
    case OPUSGPU_INVALID_DATA: return "invalid data";
    case OPUSGPU_CONFIG_INVALID: return "configuration invalid";
    case OPUSGPU_CONFIG_INCOMPLETE: return "configuration incomplete";
    case OPUSGPU_UNSUPPORTED: return "unsupported";
    case OPUSGPU_MORE_DATA_NEEDED: return "more data needed";
    case OPUSGPU_DEVICE: return "device error";
    case OPUSGPU_BAD_ARG: return "bad argument";
    default: return "unknown error";
    }
}

int opusgpu_packet_undelimit(const uint8_t *pkt, size_t len, uint8_t *out, size_t out_cap, size_t *out_len, int *consumed)
{
    if (!pkt || !out || !out_len || !consumed) return OPUSGPU_BAD_ARG;
    std::vector<uint8_t> v;
    const int rc = og::undelimit_packet(pkt, len > 0x7FFFFFFF ? 0x7FFFFFFF : int(len), v, consumed);
    if (rc) return rc;
    if (v.size() > out_cap) return OPUSGPU_BAD_ARG;
    std::memcpy(out, v.data(), v.size());
    *out_len = v.size();
    return OPUSGPU_OK;
}

int opusgpu_packet_parse(const uint8_t *pkt, size_t len, OgPacket *out)
{
    if (!out) return OPUSGPU_BAD_ARG;
    og::Packet p;
    int rc = og::parse_packet(pkt, len > 0x7FFFFFFF ? -1 : int(len), &p);
    out->mode = p.mode; out->bandwidth = p.bandwidth; out->frame_size = p.frame_size; out->stereo = p.stereo;
    out->code = p.code; out->nframes = p.nframes; out->padding = p.padding;
    std::memcpy(out->off, p.off, sizeof p.off);
    std::memcpy(out->len, p.len, sizeof p.len);
    return rc;
}

int opusgpu_parser_create(opusgpu_parser **out, int channels)
{
    if (!out || channels < 1 || channels > 2) return channels > 2 ? OPUSGPU_UNSUPPORTED : OPUSGPU_BAD_ARG;
    *out = new (std::nothrow) opusgpu_parser(channels);
    return *out ? OPUSGPU_OK : OPUSGPU_BAD_ARG;
}
void opusgpu_parser_destroy(opusgpu_parser *p) { delete p; }
void opusgpu_parser_reset(opusgpu_parser *p) { if (p) p->sp.reset(); }

static int parser_parse(opusgpu_parser *p, bool fec, const uint8_t *pkt, size_t len, OgFrameInfo *frames,
                        OgCeltHdr *celt, uint32_t celt_cap, uint32_t *ncelt,
                        OgCeltLeaf *leaves, uint32_t leaf_cap, uint32_t *nleaves,
                        OgSilkFrame *silk, uint32_t silk_cap, uint32_t *nsilk)
{
    if (!p || !frames || !ncelt || !nleaves || !nsilk) return OPUSGPU_BAD_ARG;
    p->celt.clear(); p->leaves.clear(); p->silk.clear();
    int n = fec ? p->sp.parse_fec(pkt, len, frames, p->celt, p->leaves, p->silk) : p->sp.parse(pkt, len, frames, p->celt, p->leaves, p->silk);
    if (n < 0) return n;
    if (*ncelt + p->celt.size() > celt_cap || *nleaves + p->leaves.size() > leaf_cap || *nsilk + p->silk.size() > silk_cap) {
        p->sp.reset();                                  // the records are not handed over: the stream starts again (as a refused packet's)
        return OPUSGPU_BAD_ARG;
    }
    for (int f = 0; f < n; f++) { frames[f].celt_index += *ncelt; frames[f].silk_index += *nsilk; }
    for (auto &h : p->celt) h.leaf_begin += *nleaves;
    if (!p->celt.empty()) std::memcpy(celt + *ncelt, p->celt.data(), p->celt.size() * sizeof(OgCeltHdr));
    if (!p->leaves.empty()) std::memcpy(leaves + *nleaves, p->leaves.data(), p->leaves.size() * sizeof(OgCeltLeaf));
    if (!p->silk.empty()) std::memcpy(silk + *nsilk, p->silk.data(), p->silk.size() * sizeof(OgSilkFrame));
    *ncelt += uint32_t(p->celt.size()); *nleaves += uint32_t(p->leaves.size()); *nsilk += uint32_t(p->silk.size());
    return n;
}

int opusgpu_parser_parse(opusgpu_parser *p, const uint8_t *pkt, size_t len, OgFrameInfo *frames,
                         OgCeltHdr *celt, uint32_t celt_cap, uint32_t *ncelt,
                         OgCeltLeaf *leaves, uint32_t leaf_cap, uint32_t *nleaves,
                         OgSilkFrame *silk, uint32_t silk_cap, uint32_t *nsilk)
{
    return parser_parse(p, false, pkt, len, frames, celt, celt_cap, ncelt, leaves, leaf_cap, nleaves, silk, silk_cap, nsilk);
}

int opusgpu_parser_parse_fec(opusgpu_parser *p, const uint8_t *pkt, size_t len, OgFrameInfo *frames,
                             OgCeltHdr *celt, uint32_t celt_cap, uint32_t *ncelt,
                             OgCeltLeaf *leaves, uint32_t leaf_cap, uint32_t *nleaves,
                             OgSilkFrame *silk, uint32_t silk_cap, uint32_t *nsilk)
{
    return parser_parse(p, true, pkt, len, frames, celt, celt_cap, ncelt, leaves, leaf_cap, nleaves, silk, silk_cap, nsilk);
}

size_t opusgpu_sizeof_celt_state(void) { return sizeof(OgCeltState); }
size_t opusgpu_sizeof_celt_hdr(void) { return sizeof(OgCeltHdr); }
size_t opusgpu_sizeof_celt_leaf(void) { return sizeof(OgCeltLeaf); }
size_t opusgpu_sizeof_silk_frame(void) { return sizeof(OgSilkFrame); }

}  // extern "C"
