// Host parser under AddressSanitizer + UBSan: arbitrary bytes (not only well-formed packets) through StreamParser::parse /
// parse_fec and the Ogg / container reader -- nothing may read or write out of bounds, whatever the packet says.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -fno-strict-aliasing -o /tmp/parse_fuzz_asan \
//       tools/parse_fuzz_asan.cpp opus_amd/csrc/host/{celt_parse,packet,stream_parser,silk_parse,silk_tables,celt_tables,capi_parse}.cpp
//   /tmp/parse_fuzz_asan [seed] [packets]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../opus_amd/csrc/host/stream_parser.hpp"

static uint64_t x = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return uint32_t(x >> 24); }

int main(int argc, char **argv)
{
    x ^= argc > 1 ? strtoull(argv[1], nullptr, 10) * 0x2545F4914F6CDD1Dull : 0;
    const long N = argc > 2 ? atol(argv[2]) : 200000;
    std::vector<OgCeltHdr> celt;
    std::vector<OgCeltLeaf> leaves;
    std::vector<OgSilkFrame> silk;
    OgFrameInfo fi[48];
    long ok = 0, refused = 0, frames = 0;
    for (int ch = 1; ch <= 2; ch++) {
        og::StreamParser sp(ch);
        for (long i = 0; i < N; i++) {
            // lengths: mostly short, sometimes long, sometimes empty; contents: random, with runs of 0x00 / 0xFF now and then
            const uint32_t kind = rnd() % 16;
            size_t len = kind == 0 ? 0 : kind < 10 ? 1 + rnd() % 64 : kind < 15 ? 1 + rnd() % 400 : 1 + rnd() % 1600;
            std::vector<uint8_t> p(len);                 // (an exact-size heap block: any read past the end is reported)
            const uint32_t fill = rnd() % 8;
            for (auto &b : p) b = fill == 0 ? 0x00 : fill == 1 ? 0xFF : uint8_t(rnd());
            if (len && rnd() % 4) p[0] = uint8_t(rnd());    // a random TOC in front of the runs
            celt.clear(); leaves.clear(); silk.clear();
            int n;
            if (rnd() % 10 == 0) {
                n = sp.parse_fec(p.data(), len, fi, celt, leaves, silk);
                if (n > 0) frames += n;
                celt.clear(); leaves.clear(); silk.clear();
            }
            n = sp.parse(len ? p.data() : nullptr, len, fi, celt, leaves, silk);
            if (n < 0) refused++; else { ok++; frames += n; }
            if (n > 48) { std::printf("more than 48 frames: %d\n", n); return 1; }
            for (const auto &h : celt)
                if (size_t(h.leaf_begin) + h.nleaves > leaves.size()) { std::printf("leaf range past the array\n"); return 1; }
            if (rnd() % 5000 == 0) sp.reset();
        }
    }
    // the same kind of input through the C ABI's parser entry points, into record arrays of random (often too small) capacity,
    // and through the framing helpers (RFC 6716 3.2 packets, self-delimited framing of appendix B)
    long abi_ok = 0, abi_refused = 0;
    for (int ch = 1; ch <= 2; ch++) {
        opusgpu_parser *ps = nullptr;
        if (opusgpu_parser_create(&ps, ch) != 0) { std::printf("parser_create failed\n"); return 1; }
        for (long i = 0; i < N / 4; i++) {
            const size_t len = rnd() % 3 == 0 ? rnd() % 8 : 1 + rnd() % 300;
            std::vector<uint8_t> p(len);
            for (auto &b : p) b = uint8_t(rnd());
            const uint32_t ccap = rnd() % 50, lcap = rnd() % 4 == 0 ? rnd() % 64 : 8192, scap = rnd() % 8;
            std::vector<OgCeltHdr> c(ccap);
            std::vector<OgCeltLeaf> l(lcap);
            std::vector<OgSilkFrame> sf(scap);
            uint32_t nc = 0, nl = 0, ns = 0;
            const int n = (rnd() % 8 ? opusgpu_parser_parse : opusgpu_parser_parse_fec)(ps, len ? p.data() : nullptr, len, fi, c.data(), ccap, &nc, l.data(), lcap, &nl,
                                                                                         sf.data(), scap, &ns);
            if (n < 0) abi_refused++; else abi_ok++;
            if (nc > ccap || nl > lcap || ns > scap) { std::printf("more records than room\n"); return 1; }
            OgPacket pk;
            (void)opusgpu_packet_parse(len ? p.data() : nullptr, len, &pk);
            std::vector<uint8_t> out(rnd() % 400);
            size_t out_len = 0;
            int consumed = 0;
            const int rc = opusgpu_packet_undelimit(p.data(), len, out.data(), out.size(), &out_len, &consumed);
            if (rc == 0 && (out_len > out.size() || consumed < 0 || size_t(consumed) > len)) { std::printf("undelimit out of range\n"); return 1; }
        }
        opusgpu_parser_destroy(ps);
    }
    std::printf("C ABI parser: %ld accepted, %ld refused\n", abi_ok, abi_refused);
    std::printf("parse_fuzz_asan: %ld packets accepted (%ld frames), %ld refused, no sanitizer report\n", ok, frames, refused);
    return 0;
}
