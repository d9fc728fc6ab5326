// Host parser microbenchmark: one thread, the bench workload's packets (TOC 0xFC + 160 random payload bytes, one stream's
// frames in order).  Build and run:
//   g++ -O3 -std=c++17 -fno-strict-aliasing -o /tmp/parse_bench tools/parse_bench.cpp opus_amd/csrc/host/{celt_parse,packet,stream_parser,silk_parse,silk_tables}.cpp opus_amd/csrc/rfc6716_tables.cpp
//   /tmp/parse_bench [streams] [frames]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../opus_amd/csrc/host/stream_parser.hpp"

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 64, F = argc > 2 ? atoi(argv[2]) : 200, P = 160;
    std::vector<uint8_t> pk(size_t(S) * F * (P + 1));
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < pk.size(); i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        pk[i] = (i % (P + 1)) == 0 ? 0xFC : uint8_t(x >> 24);
    }
    std::vector<OgCeltHdr> celt;
    std::vector<OgCeltLeaf> leaves;
    std::vector<OgSilkFrame> silk;
    OgFrameInfo fi[48];
    size_t nl = 0, nf = 0;
    double best = 1e30;
    for (int rep = 0; rep < 25; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        nl = nf = 0;
        for (int s = 0; s < S; s++) {
            og::StreamParser sp(2);
            for (int f = 0; f < F; f++) {
                celt.clear(); leaves.clear(); silk.clear();
                const int n = sp.parse(pk.data() + (size_t(s) * F + f) * (P + 1), P + 1, fi, celt, leaves, silk);
                if (n != 1) { std::printf("parse failed %d\n", n); return 1; }
                nl += leaves.size(); nf++;
            }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
    }
    std::printf("%.2f us/frame, %.1f leaves/frame, %.0f audio-s/s per thread\n", best / nf * 1e6, double(nl) / nf, nf * 0.02 / best);
    return 0;
}
