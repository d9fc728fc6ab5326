// C ABI, container layer (include/opusgpu.h): OpusHead parsing and Matroska / Ogg demux of one Opus track.
//
// The reference has no container code of its own: its integration tests pull packets out of the RFC 6716 vectors
// with the `matroska` and `av-format` crates (src/decoder.rs:202-232: MkvDemuxer -> Event::NewPacket -> send_packet)
// and read OpusHead fields in Dec::configure (src/decoder.rs:133-182: channels byte 9, gain 16..17, mapping family 18,
// stream / coupled counts and the mapping table behind them).  This file gives a caller of the C ABI the same two
// things without those crates: (a) OpusHead -> struct, (b) a file image -> extradata + packets.  SURVEY.md 8(f) rank 1.
//
// Everything is bounds-checked against the caller's buffer; malformed input yields OPUSGPU_INVALID_DATA, never a crash.
#include <cstring>
#include <new>
#include <vector>
#include "../../../include/opusgpu.h"

extern "C" int opusgpu_opus_head_parse(const uint8_t *e, size_t len, OgOpusHead *out)
{
    if (!e || !out) return OPUSGPU_BAD_ARG;
    std::memset(out, 0, sizeof *out);
    if (len < 19 || std::memcmp(e, "OpusHead", 8) != 0) return OPUSGPU_CONFIG_INVALID;
    out->version = e[8];
    if ((out->version & 0xF0) != 0) return OPUSGPU_UNSUPPORTED;          // RFC 7845 5.1: major version must be 0
    out->channels = e[9];                                                // decoder.rs:145
    out->pre_skip = uint16_t(e[10] | e[11] << 8);
    out->input_sample_rate = uint32_t(e[12]) | uint32_t(e[13]) << 8 | uint32_t(e[14]) << 16 | uint32_t(e[15]) << 24;
    out->output_gain_q8 = int16_t(uint16_t(e[16] | e[17] << 8));         // decoder.rs:148
    out->mapping_family = e[18];                                         // decoder.rs:149
    if (out->channels < 1) return OPUSGPU_CONFIG_INVALID;
    if (out->mapping_family == 0) {                                      // decoder.rs:158-165: implicit mono / stereo
        if (out->channels > 2) return OPUSGPU_CONFIG_INVALID;
        out->stream_count = 1;
        out->coupled_count = uint8_t(out->channels - 1);
        out->mapping[0] = 0;
        out->mapping[1] = 1;
        return OPUSGPU_OK;
    }
    if (out->mapping_family == 1 && out->channels > 8) return OPUSGPU_CONFIG_INVALID;   // RFC 7845 5.1.1.2: Vorbis order, 1..8 channels
    if (len < size_t(21) + out->channels) return OPUSGPU_CONFIG_INCOMPLETE;   // decoder.rs:151-157, :167
    out->stream_count = e[19];
    out->coupled_count = e[20];
    if (out->stream_count < 1 || out->coupled_count > out->stream_count ||
        int(out->stream_count) + int(out->coupled_count) > 255)
        return OPUSGPU_CONFIG_INVALID;
    for (int c = 0; c < out->channels; c++) {
        out->mapping[c] = e[21 + c];
        if (out->mapping[c] != 255 && out->mapping[c] >= out->stream_count + out->coupled_count) return OPUSGPU_CONFIG_INVALID;
    }
    return OPUSGPU_OK;
}

namespace {

struct Pkt { size_t off, len; int64_t pts; };

struct Cursor {                         // bounds-checked big-endian reader over [p, end)
    const uint8_t *b;
    size_t p, end;
    bool ok = true;
    uint8_t u8() { if (p >= end) { ok = false; return 0; } return b[p++]; }
    // EBML variable-length integer; keep_marker keeps the length marker bit (element ids)
    uint64_t vint(bool keep_marker, bool *unknown = nullptr, int *width = nullptr)
    {
        const uint8_t first = u8();
        if (!ok || first == 0) { ok = false; return 0; }
        int n = 1;
        uint8_t mask = 0x80;
        while (!(first & mask)) { mask >>= 1; n++; }
        uint64_t v = keep_marker ? first : uint64_t(first & (mask - 1));
        bool all_ones = (first & (mask - 1)) == (mask - 1);
        for (int i = 1; i < n; i++) {
            const uint8_t c = u8();
            if (c != 0xFF) all_ones = false;
            v = v << 8 | c;
        }
        if (unknown) *unknown = all_ones;
        if (width) *width = n;
        return v;
    }
    uint64_t uint_be(size_t n)
    {
        uint64_t v = 0;
        for (size_t i = 0; i < n && i < 8; i++) v = v << 8 | u8();
        return v;
    }
};

}  // namespace

struct opusgpu_demux {
    int container = 0;                  // 0 Matroska, 1 Ogg
    std::vector<uint8_t> extra;         // OpusHead
    std::vector<uint8_t> arena;         // packet bytes, back to back
    std::vector<Pkt> pkts;

    // ---- Matroska
    uint64_t timecode_scale = 1000000;  // ns per tick (Info/TimecodeScale default)
    uint64_t opus_track = 0;
    bool have_track = false;
    int64_t cluster_tc = 0;

    void add_packet(const uint8_t *p, size_t n, int64_t pts)
    {
        pkts.push_back({arena.size(), n, pts});
        arena.insert(arena.end(), p, p + n);
    }

    bool block(const uint8_t *b, size_t p, size_t end)          // (Simple)Block payload
    {
        Cursor c{b, p, end};
        const uint64_t track = c.vint(false);
        const int16_t rel = int16_t(c.uint_be(2));
        const uint8_t flags = c.u8();
        if (!c.ok) return false;
        if (!have_track || track != opus_track) return true;    // another track: skip
        const int64_t pts = int64_t((uint64_t(cluster_tc) + uint64_t(int64_t(rel))) * timecode_scale);    // (a hostile scale wraps: unsigned)
        const int lacing = (flags >> 1) & 3;
        if (lacing == 0) { add_packet(b + c.p, end - c.p, pts); return true; }
        const int nfr = int(c.u8()) + 1;
        if (!c.ok) return false;
        std::vector<size_t> sz(nfr, 0);
        if (lacing == 1) {                                      // Xiph: 255-runs, last size implied
            for (int i = 0; i + 1 < nfr; i++) {
                uint8_t v;
                do { v = c.u8(); sz[i] += v; } while (c.ok && v == 255);
            }
        } else if (lacing == 3) {                               // EBML: first size, then signed deltas
            int w = 0;
            int64_t cur = int64_t(c.vint(false, nullptr, &w));
            if (nfr > 1) sz[0] = size_t(cur);
            for (int i = 1; i + 1 < nfr; i++) {
                const uint64_t raw = c.vint(false, nullptr, &w);
                if (!c.ok) return false;                        // (w is not set then)
                cur += int64_t(raw) - ((int64_t(1) << (7 * w - 1)) - 1);
                if (cur < 0) return false;
                sz[i] = size_t(cur);
            }
        }
        if (!c.ok) return false;
        const size_t left = end - c.p;
        if (lacing == 2) {                                      // fixed size
            if (left % size_t(nfr)) return false;
            for (int i = 0; i < nfr; i++) sz[i] = left / size_t(nfr);
        } else {
            size_t used = 0;
            for (int i = 0; i + 1 < nfr; i++) used += sz[i];
            if (used > left) return false;
            sz[nfr - 1] = left - used;
        }
        size_t q = c.p;
        for (int i = 0; i < nfr; i++) { add_packet(b + q, sz[i], pts); q += sz[i]; }
        return true;
    }

    bool track_entry(const uint8_t *b, size_t p, size_t end)
    {
        uint64_t number = 0;
        bool is_opus = false;
        size_t priv_off = 0, priv_len = 0;
        Cursor c{b, p, end};
        while (c.p < end) {
            const uint64_t id = c.vint(true);
            bool unk = false;
            const uint64_t sz = c.vint(false, &unk);
            if (!c.ok || unk || sz > end - c.p) return false;
            if (id == 0xD7) { Cursor d{b, c.p, c.p + size_t(sz)}; number = d.uint_be(size_t(sz)); }
            else if (id == 0x86) is_opus = sz == 6 && std::memcmp(b + c.p, "A_OPUS", 6) == 0;
            else if (id == 0x63A2) { priv_off = c.p; priv_len = size_t(sz); }
            c.p += size_t(sz);
        }
        if (is_opus && !have_track) {
            have_track = true;
            opus_track = number;
            extra.assign(b + priv_off, b + priv_off + priv_len);
        }
        return true;
    }

    // Walks the children of a master element.  `parent` says which one, and each master id is entered only below its
    // proper parent (Segment at the top, Tracks / Info / Cluster in the Segment, BlockGroup in a Cluster), so the
    // recursion is at most four frames deep whatever the file nests; a master element anywhere else is malformed.
    // An unknown-size Cluster (live muxers) ends where the next Segment-level element starts: the walk returns that
    // position in *resume and the Segment loop goes on from there (no recursion from cluster to cluster).
    enum Parent { TOP, SEGMENT, TRACKS, INFO, CLUSTER, BLOCKGROUP };
    static bool segment_level_id(uint64_t id)
    {
        switch (id) {
        case 0x1F43B675: case 0x1654AE6B: case 0x1549A966: case 0x114D9B74: case 0x1C53BB6B: case 0x1254C367:
        case 0x1941A469: case 0x1043A770:               // Cluster, Tracks, Info, SeekHead, Cues, Tags, Attachments, Chapters
            return true;
        default: return false;
        }
    }
    bool walk(const uint8_t *b, size_t p, size_t end, Parent parent, bool open_ended, size_t *resume)
    {
        Cursor c{b, p, end};
        while (c.p < end) {
            const size_t at = c.p;
            const uint64_t id = c.vint(true);
            bool unk = false;
            uint64_t sz = c.vint(false, &unk);
            if (!c.ok) return false;
            if (open_ended && parent == CLUSTER && (segment_level_id(id) || id == 0x18538067)) { *resume = at; return true; }
            if (unk) {                                          // unknown size: only Segment / Cluster may (live muxers)
                if (id != 0x18538067 && id != 0x1F43B675) return false;
                sz = end - c.p;
            }
            if (sz > end - c.p) return false;
            const size_t q = c.p;
            size_t qe = c.p + size_t(sz);
            switch (id) {
            case 0x18538067:                                    // Segment
                if (parent != TOP || !walk(b, q, qe, SEGMENT, false, nullptr)) return false;
                break;
            case 0x1654AE6B:                                    // Tracks
                if (parent != SEGMENT || !walk(b, q, qe, TRACKS, false, nullptr)) return false;
                break;
            case 0x1549A966:                                    // Info
                if (parent != SEGMENT || !walk(b, q, qe, INFO, false, nullptr)) return false;
                break;
            case 0x1F43B675: {                                  // Cluster
                if (parent != SEGMENT) return false;
                cluster_tc = 0;
                size_t stop = qe;
                if (!walk(b, q, qe, CLUSTER, unk, &stop)) return false;
                qe = stop;
                break;
            }
            case 0xA0:                                          // BlockGroup
                if (parent != CLUSTER || !walk(b, q, qe, BLOCKGROUP, false, nullptr)) return false;
                break;
            case 0xAE: if (parent == TRACKS && !track_entry(b, q, qe)) return false; break;
            case 0x2AD7B1: if (parent == INFO) { Cursor d{b, q, qe}; timecode_scale = d.uint_be(size_t(sz)); } break;
            case 0xE7: if (parent == CLUSTER) { Cursor d{b, q, qe}; cluster_tc = int64_t(d.uint_be(size_t(sz))); } break;
            case 0xA3:                                          // SimpleBlock
                if (parent == CLUSTER && !block(b, q, qe)) return false;
                break;
            case 0xA1:                                          // Block
                if (parent == BLOCKGROUP && !block(b, q, qe)) return false;
                break;
            default: break;                                     // EBML header, SeekHead, Cues, Tags, Void, ...: skipped
            }
            c.p = qe;
        }
        if (resume) *resume = end;
        return true;
    }

    // ---- Ogg (RFC 3533 pages, RFC 7845 mapping)
    bool ogg(const uint8_t *b, size_t len)
    {
        static uint32_t crc_table[256];
        static bool crc_init = false;
        if (!crc_init) {
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t r = i << 24;
                for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1;
                crc_table[i] = r;
            }
            crc_init = true;
        }
        bool have_serial = false, in_packet = false;
        uint32_t serial = 0;
        int npk = 0;                                            // packets of the chosen stream seen so far
        std::vector<uint8_t> cur;
        size_t p = 0;
        while (p < len) {
            if (len - p < 27 || std::memcmp(b + p, "OggS", 4) != 0 || b[p + 4] != 0) return false;
            const uint8_t htype = b[p + 5];
            int64_t granule = 0;
            for (int i = 7; i >= 0; i--) granule = int64_t(uint64_t(granule) << 8 | b[p + 6 + i]);
            const uint32_t ser = uint32_t(b[p + 14]) | uint32_t(b[p + 15]) << 8 | uint32_t(b[p + 16]) << 16 | uint32_t(b[p + 17]) << 24;
            const uint32_t want = uint32_t(b[p + 22]) | uint32_t(b[p + 23]) << 8 | uint32_t(b[p + 24]) << 16 | uint32_t(b[p + 25]) << 24;
            const int nseg = b[p + 26];
            if (len - p < size_t(27 + nseg)) return false;
            size_t body = 0;
            for (int i = 0; i < nseg; i++) body += b[p + 27 + i];
            const size_t total = size_t(27 + nseg) + body;
            if (len - p < total) return false;
            uint32_t crc = 0;
            for (size_t i = 0; i < total; i++) {
                const uint8_t v = (i >= 22 && i < 26) ? 0 : b[p + i];
                crc = (crc << 8) ^ crc_table[((crc >> 24) ^ v) & 0xFF];
            }
            if (crc != want) return false;
            const uint8_t *seg = b + p + 27, *data = seg + nseg;
            p += total;
            if (have_serial && ser != serial) continue;         // another logical stream
            if (!have_serial) {
                if (!(htype & 2)) continue;                     // wait for a beginning-of-stream page
                if (body < 8 || std::memcmp(data, "OpusHead", 8) != 0) continue;
                have_serial = true;
                serial = ser;
            }
            if (!(htype & 1) && in_packet) { cur.clear(); in_packet = false; }     // lost continuation: drop the stub
            bool skip_cont = (htype & 1) && !in_packet;          // continuation of a packet whose start was not seen
            size_t q = 0;
            // packets that END on this page take its granule position; pts of the others is -1
            int last_complete = -1;
            for (int i = 0; i < nseg; i++) if (seg[i] < 255) last_complete = i;
            for (int i = 0; i < nseg; i++) {
                if (!skip_cont) cur.insert(cur.end(), data + q, data + q + seg[i]);
                q += seg[i];
                in_packet = true;
                if (seg[i] < 255) {
                    if (!skip_cont) {
                        if (npk == 0) extra = cur;
                        else if (npk == 1) { if (cur.size() < 8 || std::memcmp(cur.data(), "OpusTags", 8) != 0) return false; }
                        else add_packet(cur.data(), cur.size(), i == last_complete ? granule : -1);
                        npk++;
                    }
                    cur.clear();
                    in_packet = false;
                    skip_cont = false;
                }
            }
        }
        return have_serial && npk >= 1;
    }
};

extern "C" {

int opusgpu_demux_open(opusgpu_demux **out, const uint8_t *data, size_t len)
{
    if (!out || !data) return OPUSGPU_BAD_ARG;
    *out = nullptr;
    if (len < 4) return OPUSGPU_INVALID_DATA;
    opusgpu_demux *d = new (std::nothrow) opusgpu_demux;
    if (!d) return OPUSGPU_BAD_ARG;
    bool ok = false;
    if (std::memcmp(data, "OggS", 4) == 0) { d->container = 1; ok = d->ogg(data, len); }
    else if (data[0] == 0x1A && data[1] == 0x45 && data[2] == 0xDF && data[3] == 0xA3) {
        d->container = 0;
        ok = d->walk(data, 0, len, opusgpu_demux::TOP, false, nullptr) && d->have_track;
    }
    OgOpusHead head;
    int hrc = ok ? opusgpu_opus_head_parse(d->extra.data(), d->extra.size(), &head) : OPUSGPU_INVALID_DATA;
    // a track whose OpusHead no decoder could be configured from is not an Opus track (an empty or short CodecPrivate,
    // a header cut short); a well-formed header this build cannot decode (version) still opens: configure reports it
    if (hrc != OPUSGPU_OK && hrc != OPUSGPU_UNSUPPORTED) {
        delete d;
        return OPUSGPU_INVALID_DATA;
    }
    *out = d;
    return OPUSGPU_OK;
}

void opusgpu_demux_close(opusgpu_demux *d) { delete d; }

int opusgpu_demux_info(const opusgpu_demux *d, int *container, const uint8_t **extradata, size_t *extralen, uint32_t *npackets)
{
    if (!d) return OPUSGPU_BAD_ARG;
    if (container) *container = d->container;
    if (extradata) *extradata = d->extra.data();
    if (extralen) *extralen = d->extra.size();
    if (npackets) *npackets = uint32_t(d->pkts.size());
    return OPUSGPU_OK;
}

int opusgpu_demux_packet(const opusgpu_demux *d, uint32_t i, const uint8_t **pkt, size_t *len, int64_t *pts)
{
    if (!d || !pkt || !len) return OPUSGPU_BAD_ARG;
    if (i >= d->pkts.size()) return OPUSGPU_MORE_DATA_NEEDED;
    *pkt = d->arena.data() + d->pkts[i].off;
    *len = d->pkts[i].len;
    if (pts) *pts = d->pkts[i].pts;
    return OPUSGPU_OK;
}

}  // extern "C"
