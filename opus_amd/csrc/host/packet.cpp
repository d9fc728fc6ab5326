// Opus packet framing for the opus_amd host layer.
//
// Replaces Packet::from_slice and its helpers (lu-zero/opus src/packet.rs:77-317): TOC byte ->
// mode / bandwidth / frame duration / stereo, then frame slicing for codes 0-3.  Written from
// RFC 6716 section 3.2 as a cursor over [p, end); it never aborts (the reference panics on an empty
// packet, packet.rs:215-225) and fixes the reference's code-1 and code-3-VBR slicing defects
// (packet.rs:133, :177-188; SURVEY.md section 8 notes).
#include "parse.hpp"
#include "../../../include/opusgpu.h"

namespace og {
namespace {

struct Cursor {
    const uint8_t *base, *p, *end;
    int left() const { return int(end - p); }
    // RFC 6716 3.2.1 frame length coding (packet.rs:77-89): 0..251 one byte, else second byte * 4 + first
    bool frame_length(int *out)
    {
        if (left() < 1) return false;
        int b0 = *p++;
        if (b0 < 252) { *out = b0; return true; }
        if (left() < 1) return false;
        *out = 4 * int(*p++) + b0;
        return true;
    }
};

}  // namespace

int parse_packet(const uint8_t *buf, int len, Packet *pk)
{
    *pk = Packet{};
    if (!buf || len < 1) return OPUSGPU_INVALID_DATA;
    const int toc = buf[0];
    const int config = toc >> 3;                                   // packet.rs:219-221
    pk->stereo = (toc >> 2) & 1;
    pk->code = toc & 3;
    if (config >= 16) {                                            // CELT-only: NB WB SWB FB x 2.5/5/10/20 ms
        static const int bw[4] = {BW_NB, BW_WB, BW_SWB, BW_FB};
        pk->mode = MODE_CELT;
        pk->bandwidth = bw[(config >> 2) & 3];
        pk->frame_size = 120 << (config & 3);
    } else if (config >= 12) {                                     // hybrid: SWB FB x 10/20 ms
        pk->mode = MODE_HYBRID;
        pk->bandwidth = config >= 14 ? BW_FB : BW_SWB;
        pk->frame_size = (config & 1) ? 960 : 480;
    } else {                                                       // SILK-only: NB MB WB x 10/20/40/60 ms
        static const int dur[4] = {480, 960, 1920, 2880};
        pk->mode = MODE_SILK;
        pk->bandwidth = config >> 2;
        pk->frame_size = dur[config & 3];
    }

    Cursor c{buf, buf + 1, buf + len};
    auto place = [&](int i, int n) { pk->off[i] = int(c.p - c.base); pk->len[i] = n; c.p += n; };
    switch (pk->code) {
    case 0:                                                        // one frame
        pk->nframes = 1;
        place(0, c.left());
        break;
    case 1: {                                                      // two equal frames
        if (c.left() & 1) return OPUSGPU_INVALID_DATA;
        int half = c.left() / 2;
        pk->nframes = 2;
        place(0, half);
        place(1, half);
        break;
    }
    case 2: {                                                      // two frames, first length coded
        int n0;
        if (!c.frame_length(&n0) || n0 > c.left()) return OPUSGPU_INVALID_DATA;
        pk->nframes = 2;
        place(0, n0);
        place(1, c.left());
        break;
    }
    default: {                                                     // arbitrary count (packet.rs:155-200)
        if (c.left() < 1) return OPUSGPU_INVALID_DATA;
        const int fc = *c.p++;
        const bool vbr = fc & 0x80, padded = fc & 0x40;
        const int count = fc & 0x3F;
        if (count == 0 || count * pk->frame_size > 5760) return OPUSGPU_INVALID_DATA;   // > 120 ms
        if (padded) {                                              // packet.rs:91-111
            int b;
            do {
                if (c.left() < 1) return OPUSGPU_INVALID_DATA;
                b = *c.p++;
                pk->padding += b == 255 ? 254 : b;
            } while (b == 255);
            if (pk->padding > c.left()) return OPUSGPU_INVALID_DATA;
            c.end -= pk->padding;
        }
        pk->nframes = count;
        if (vbr) {
            int lens[48], sum = 0;
            for (int i = 0; i + 1 < count; i++) {
                if (!c.frame_length(&lens[i])) return OPUSGPU_INVALID_DATA;
                sum += lens[i];
            }
            if (sum > c.left()) return OPUSGPU_INVALID_DATA;
            lens[count - 1] = c.left() - sum;
            for (int i = 0; i < count; i++) place(i, lens[i]);
        } else {
            if (c.left() % count) return OPUSGPU_INVALID_DATA;
            int each = c.left() / count;
            for (int i = 0; i < count; i++) place(i, each);
        }
        break;
    }
    }
    for (int i = 0; i < pk->nframes; i++)
        if (pk->len[i] > 1275) return OPUSGPU_INVALID_DATA;
    return OPUSGPU_OK;
}

// RFC 6716 Appendix B: a self-delimiting packet (every stream but the last of a multistream packet, RFC 7845 5.1.1) carries
// the length of its last frame as well, so that its end can be found.  Rewrites the one at buf as the ordinary packet with
// the same frames into `out` and reports how many bytes it took.  0 on success.
int undelimit_packet(const uint8_t *buf, int len, std::vector<uint8_t> &out, int *consumed)
{
    out.clear();
    if (!buf || len < 2) return OPUSGPU_INVALID_DATA;
    Cursor c{buf, buf + 1, buf + len};
    const int code = buf[0] & 3;
    out.push_back(buf[0]);
    auto copy = [&](int n) { if (n < 0 || n > c.left()) return false; out.insert(out.end(), c.p, c.p + n); c.p += n; return true; };
    int n0, n1;
    switch (code) {
    case 0:
        if (!c.frame_length(&n0) || !copy(n0)) return OPUSGPU_INVALID_DATA;
        break;
    case 1:
        if (!c.frame_length(&n0) || !copy(2 * n0)) return OPUSGPU_INVALID_DATA;
        break;
    case 2: {
        const uint8_t *l0 = c.p;
        if (!c.frame_length(&n0)) return OPUSGPU_INVALID_DATA;
        out.insert(out.end(), l0, c.p);                                    // the first frame's length stays
        if (!c.frame_length(&n1) || !copy(n0 + n1)) return OPUSGPU_INVALID_DATA;
        break;
    }
    default: {
        if (c.left() < 1) return OPUSGPU_INVALID_DATA;
        const int fc = *c.p;
        const bool vbr = fc & 0x80, padded = fc & 0x40;
        const int count = fc & 0x3F;
        if (count == 0) return OPUSGPU_INVALID_DATA;
        if (!copy(1)) return OPUSGPU_INVALID_DATA;
        int padding = 0;
        if (padded) {
            int b;
            do {
                if (c.left() < 1) return OPUSGPU_INVALID_DATA;
                b = *c.p;
                padding += b == 255 ? 254 : b;
                copy(1);
            } while (b == 255);
        }
        int total = 0;
        if (vbr) {
            for (int i = 0; i + 1 < count; i++) {
                const uint8_t *l = c.p;
                if (!c.frame_length(&n0)) return OPUSGPU_INVALID_DATA;
                out.insert(out.end(), l, c.p);
                total += n0;
            }
            if (!c.frame_length(&n0)) return OPUSGPU_INVALID_DATA;          // the last frame's length: the added field
            total += n0;
        } else {
            if (!c.frame_length(&n0)) return OPUSGPU_INVALID_DATA;          // every frame's length: the added field
            total = n0 * count;
        }
        if (!copy(total) || !copy(padding)) return OPUSGPU_INVALID_DATA;
        break;
    }
    }
    *consumed = int(c.p - c.base);
    return OPUSGPU_OK;
}

}  // namespace og
