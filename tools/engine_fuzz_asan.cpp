// The engine's host side (batch sends, merge / sort / launch plan, multi-packet sends, FEC sends, flushes, the one-stream
// decoder with multistream extradata) under AddressSanitizer, on the GPU: the host code of the whole library is built with
// -fsanitize=address (the device code is not: no GPU sanitizer on this pool) and linked into this driver, which feeds random
// packets under every TOC.  No oracle here (tests/test_gpu_opus.py and tools/fuzz_gpu.py compare PCM): this run is about memory.
//   build (CPU container):  bash tools/build_engine_fuzz_asan.sh      -> variants/engine_fuzz_asan
//   run (GPU box):          ASAN_OPTIONS=detect_leaks=0 variants/engine_fuzz_asan <seed> <seconds>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/opusgpu.h"

static uint64_t x = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return uint32_t(x >> 24); }
static uint32_t below(uint32_t n) { return n ? rnd() % n : 0; }

// tests/test_emu_silk.py::random_packet: TOC (config, stereo, code) and random frame contents
static std::vector<uint8_t> random_packet(int config, int stereo, int code)
{
    const double ms = config < 12 ? (double[]){10, 20, 40, 60}[config & 3] : config < 16 ? (double[]){10, 20}[config & 1] : (double[]){2.5, 5, 10, 20}[config & 3];
    auto size = [&] { const int hi = std::max(4, int(ms * 6)); return 2 + int(below(uint32_t(hi - 2))); };
    std::vector<uint8_t> p{uint8_t(config << 3 | stereo << 2 | code)};
    auto body = [&](int n) { for (int i = 0; i < n; i++) p.push_back(uint8_t(rnd())); };
    if (code == 0) body(size());
    else if (code == 1) body(2 * size());
    else if (code == 2) { const int n1 = std::min(size(), 251); p.push_back(uint8_t(n1)); body(n1); body(size()); }
    else {
        int m = 1 + int(below(uint32_t(std::max(1, int(120 / ms)))));
        m = std::min(m, 6);
        if (rnd() & 1) { const int n = size(); p.push_back(uint8_t(m)); body(m * n); }
        else {
            std::vector<int> sz(m);
            for (auto &s : sz) s = std::min(size(), 251);
            p.push_back(uint8_t(0x80 | m));
            for (int i = 0; i + 1 < m; i++) p.push_back(uint8_t(sz[i]));
            for (int s : sz) body(s);
        }
    }
    return p;
}

static bool g_trace = false;
#define MARK(what) { if (g_trace) { std::printf("  . %s\n", what); std::fflush(stdout); } }
#define CHECK(call) { MARK(#call) const int rc_ = (call); if (rc_ != 0) { std::printf("%s -> %d (%s)\n", #call, rc_, opusgpu_last_error()); return 1; } }

int main(int argc, char **argv)
{
    x ^= (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1) * 0x2545F4914F6CDD1Dull;
    const double budget = argc > 2 ? atof(argv[2]) : 60;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    CHECK(opusgpu_init(0));
    long sends = 0, packets = 0, refused = 0, rounds = 0;
    double last = 0;
    while (elapsed() < budget) {
        rounds++;
        g_trace = getenv("ENGINE_FUZZ_TRACE_AFTER") && elapsed() > atof(getenv("ENGINE_FUZZ_TRACE_AFTER"));
        MARK("round")
        // ---- a batch of streams of random size
        const uint32_t ns = 1 + below(160);
        opusgpu_batch *b = nullptr;
        CHECK(opusgpu_batch_create(&b, ns, 1 + int(rnd() & 1), 0));
        std::vector<float> pcm(5760 * 2 * 8);
        for (int step = 0; step < 40; step++) {
            std::vector<uint32_t> ids;
            for (uint32_t s = 0; s < ns; s++) if (below(5)) ids.push_back(s);
            if (ids.empty()) continue;
            const uint32_t T = step % 5 == 2 ? 1 : (uint32_t[]){1, 1, 2, 3, 4, 8}[below(6)];
            const bool uniform = step % 3 == 0;
            const int ucfg = 16 + int(below(16)), ucode = int(below(3));
            std::vector<std::vector<uint8_t>> store;
            for (size_t i = 0; i < ids.size() * T; i++) {
                if (uniform) store.push_back(random_packet(ucfg, int(rnd() & 1), ucode));
                else if (below(25) == 0) store.emplace_back();                 // lost
                else if (below(40) == 0) { store.emplace_back(1 + below(40)); for (auto &v : store.back()) v = uint8_t(rnd()); }   // arbitrary bytes
                else store.push_back(random_packet(int(below(32)), int(rnd() & 1), int(below(4))));
            }
            std::vector<const uint8_t *> ptr;
            std::vector<uint32_t> lens;
            for (auto &p : store) { ptr.push_back(p.empty() ? nullptr : p.data()); lens.push_back(uint32_t(p.size())); }
            std::vector<int32_t> errs(store.size());
            if (T == 1 && step % 5 == 2) {
                CHECK(opusgpu_batch_send_fec(b, ptr.data(), lens.data(), ids.data(), uint32_t(ids.size()), errs.data()));
                CHECK(opusgpu_batch_decode(b));
                sends++;
            }
            if (T == 1) { CHECK(opusgpu_batch_send(b, ptr.data(), lens.data(), ids.data(), uint32_t(ids.size()), rnd() & 7 ? errs.data() : nullptr)); }
            else CHECK(opusgpu_batch_send_multi(b, ptr.data(), lens.data(), ids.data(), uint32_t(ids.size()), T, errs.data()));
            for (int32_t e : errs) refused += e != 0;
            CHECK(opusgpu_batch_decode(b));
            sends++;
            packets += long(store.size());
            if (pcm.size() < size_t(5760) * 2 * T) pcm.resize(size_t(5760) * 2 * T);
            for (int k = 0; k < 3; k++) {
                size_t n = 0;
                CHECK(opusgpu_batch_read(b, ids[below(uint32_t(ids.size()))], pcm.data(), pcm.size(), &n));
            }
            if (below(6) == 0) CHECK(opusgpu_batch_flush(b, below(ns)));
            if (below(4) == 0) { OgBatchPlan pl; CHECK(opusgpu_batch_plan(b, &pl)); }
        }
        CHECK(opusgpu_batch_sync(b));
        MARK("batch_destroy")
        opusgpu_batch_destroy(b);
        MARK("batch destroyed")
        // ---- the one-stream decoder: plain, or multistream with a random (valid) mapping
        opusgpu_decoder *d = nullptr;
        CHECK(opusgpu_decoder_create(&d, 0));
        int nstreams = 1, ncoupled = 0;
        const uint32_t fam = below(3);                                          // 0: default stereo; 1: mapping family 1; 2: family 255
        if (fam) {
            nstreams = 1 + int(below(4));
            ncoupled = int(below(uint32_t(nstreams + 1)));
            const int channels = fam == 1 ? std::min(8, nstreams + ncoupled) : 1 + int(below(12));
            std::vector<uint8_t> e{'O', 'p', 'u', 's', 'H', 'e', 'a', 'd', 1, uint8_t(channels), 0, 0, 0x80, 0xBB, 0, 0, 0, 0, uint8_t(fam == 1 ? 1 : 255),
                                   uint8_t(nstreams), uint8_t(ncoupled)};
            for (int c = 0; c < channels; c++) e.push_back(below(8) == 0 ? 255 : uint8_t(below(uint32_t(nstreams + ncoupled))));
            if (opusgpu_decoder_set_extradata(d, e.data(), e.size()) != 0) { nstreams = 1; refused++; }
        }
        if (opusgpu_decoder_configure(d) == 0) {
            std::vector<float> out(5760 * 16);
            for (int i = 0; i < 60; i++) {
                std::vector<uint8_t> pkt;
                for (int s = 0; s < nstreams; s++) {                            // RFC 7845 5.1.1: all but the last self-delimited
                    std::vector<uint8_t> p = random_packet(int(below(32)), int(rnd() & 1), 0);
                    if (s + 1 < nstreams) {                                     // code 0 self-delimited: TOC, length, frame
                        const size_t n = p.size() - 1;
                        pkt.push_back(p[0]);
                        if (n < 252) pkt.push_back(uint8_t(n)); else { pkt.push_back(uint8_t(252 + (n & 3))); pkt.push_back(uint8_t((n - (252 + (n & 3))) >> 2)); }
                        pkt.insert(pkt.end(), p.begin() + 1, p.end());
                    } else pkt.insert(pkt.end(), p.begin(), p.end());
                }
                if (below(20) == 0) pkt.resize(below(uint32_t(pkt.size())));   // truncated / lost
                if (below(15) == 0 && !pkt.empty()) (void)opusgpu_decoder_send_packet_fec(d, pkt.data(), pkt.size());
                size_t n = 0;
                int ch = 0;
                if (below(15) == 0) (void)opusgpu_decoder_receive_frame(d, out.data(), out.size(), &n, &ch);
                MARK("decoder_send_packet")
                if (opusgpu_decoder_send_packet(d, pkt.empty() ? nullptr : pkt.data(), pkt.size()) != 0) { refused++; continue; }
                MARK("decoder_receive_frame")
                packets++;
                (void)opusgpu_decoder_receive_frame(d, out.data(), below(10) == 0 ? 100 : out.size(), &n, &ch);
                if (below(30) == 0) (void)opusgpu_decoder_flush(d);
            }
        }
        MARK("decoder_destroy")
        opusgpu_decoder_destroy(d);
        MARK("decoder destroyed")
        if (elapsed() - last > 30) { last = elapsed(); std::printf("%.0f s: %ld rounds, %ld sends, %ld packets, %ld refused\n", last, rounds, sends, packets, refused); fflush(stdout); }
    }
    std::printf("engine_fuzz_asan: %ld rounds, %ld batch sends, %ld packets, %ld refused, no sanitizer report\n", rounds, sends, packets, refused);
    std::fflush(stdout);
    std::_Exit(0);                                      // (not through the HIP runtime's exit-time teardown: the sanitizer's allocator checks the order it unloads in)
}
