// Container layer under AddressSanitizer + UBSan: valid Matroska / Ogg images (written by tests/test_container.py's writers),
// mutated -- bytes overwritten, runs of 0x00 / 0xFF, truncations, spliced copies of their own pieces -- through
// opusgpu_demux_open / _info / _packet and opusgpu_opus_head_parse; every byte of every packet handed back is read.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -o /tmp/demux_fuzz_asan tools/demux_fuzz_asan.cpp \
//       opus_amd/csrc/host/container.cpp
//   /tmp/demux_fuzz_asan <seed> <iterations> <image files ...>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/opusgpu.h"

static uint64_t x = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return uint32_t(x >> 24); }

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    x ^= strtoull(argv[1], nullptr, 10) * 0x2545F4914F6CDD1Dull;
    const long iters = atol(argv[2]);
    std::vector<std::vector<uint8_t>> base;
    for (int i = 3; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) { std::printf("cannot open %s\n", argv[i]); return 2; }
        std::vector<uint8_t> b;
        uint8_t buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f);
        base.push_back(b);
    }
    long opened = 0, refused = 0, packets = 0;
    uint64_t sum = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> img = base[rnd() % base.size()];
        const int nm = it % 50 == 0 ? 0 : 1 + rnd() % 6;            // (now and then the image as it is)
        for (int m = 0; m < nm && !img.empty(); m++) {
            const uint32_t kind = rnd() % 8, at = rnd() % img.size();
            if (kind < 3) img[at] = uint8_t(rnd());
            else if (kind == 3) img[at] = rnd() & 1 ? 0xFF : 0x00;
            else if (kind == 4) { const size_t n = std::min<size_t>(1 + rnd() % 16, img.size() - at); memset(&img[at], rnd() & 1 ? 0xFF : 0x00, n); }
            else if (kind == 5) img.resize(at);                       // truncated
            else if (kind == 6) {                                     // a piece of the image copied over another place
                const size_t from = rnd() % img.size(), n = std::min<size_t>(1 + rnd() % 64, std::min(img.size() - from, img.size() - at));
                memmove(&img[at], &img[from], n);
            } else {                                                  // a piece inserted
                const size_t from = rnd() % img.size(), n = std::min<size_t>(1 + rnd() % 64, img.size() - from);
                std::vector<uint8_t> piece(img.begin() + from, img.begin() + from + n);
                img.insert(img.begin() + at, piece.begin(), piece.end());
            }
        }
        std::vector<uint8_t> exact(img);                              // (an exact-size heap block: reads past the end are reported)
        opusgpu_demux *d = nullptr;
        const int rc = opusgpu_demux_open(&d, exact.data(), exact.size());
        if (rc != 0) { refused++; if (d) { std::printf("a handle came back with an error\n"); return 1; } continue; }
        opened++;
        int container = -1;
        const uint8_t *extra = nullptr;
        size_t extralen = 0;
        uint32_t np = 0;
        if (opusgpu_demux_info(d, &container, &extra, &extralen, &np) != 0) { std::printf("info failed on an open handle\n"); return 1; }
        for (size_t i = 0; i < extralen; i++) sum += extra[i];
        OgOpusHead head;
        (void)opusgpu_opus_head_parse(extra, extralen, &head);
        for (uint32_t i = 0; i < np + 2; i++) {                       // (two past the end: must be refused)
            const uint8_t *p = nullptr;
            size_t len = 0;
            int64_t pts = 0;
            const int r = opusgpu_demux_packet(d, i, &p, &len, &pts);
            if (i >= np) { if (r == 0) { std::printf("a packet past the end\n"); return 1; } continue; }
            if (r != 0) { std::printf("packet %u of %u refused\n", i, np); return 1; }
            for (size_t k = 0; k < len; k++) sum += p[k];
            packets++;
        }
        opusgpu_demux_close(d);
    }
    // OpusHead on arbitrary bytes
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> e(rnd() % 64);
        for (auto &b : e) b = uint8_t(rnd());
        if (e.size() >= 8 && rnd() % 2) memcpy(e.data(), "OpusHead", 8);
        OgOpusHead head;
        (void)opusgpu_opus_head_parse(e.data(), e.size(), &head);
    }
    std::printf("demux_fuzz_asan: %ld images opened (%ld packets read, checksum %llu), %ld refused, no sanitizer report\n", opened, packets,
                (unsigned long long)sum, refused);
    return 0;
}
