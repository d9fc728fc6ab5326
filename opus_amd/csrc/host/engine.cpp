// C ABI, layers 1 and 2 (include/opusgpu.h): the batch engine and the one-stream decoder on top of it.
//
// One opusgpu_batch owns, for `nstreams` independent streams on one GPU: a host parser per stream (what
// `struct Dec` keeps in the reference, src/decoder.rs:16-30, minus sample buffers), per-stream
// synthesis state in HBM, pinned side-info staging and one HIP stream.  batch_send is the host half of
// Dec::send_packet (decoder.rs:48-127), batch_decode the device half.
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sched.h>
#include <algorithm>
#include <cctype>
#include <string>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "leaf_sort.hpp"
#include "stream_parser.hpp"
#include "../hip/device_util.hpp"


namespace {

template <class T>
struct DeviceBuf {                       // grow-only device array
    T *p = nullptr;
    size_t cap = 0;
    // Kernels of the previous decode (asynchronous) may still be reading the old block: it is freed only after `stream`
    // has drained (the contents are not carried over: every user refills the array before it launches).
    int reserve(size_t n, hipStream_t stream)
    {
        if (n <= cap) return OPUSGPU_OK;
        size_t want = std::max(n, cap * 2);
        T *q = nullptr;
        OG_HIP(hipMalloc((void **)&q, want * sizeof(T)));
        if (p) {
            hipError_t e = hipStreamSynchronize(stream);
            if (e == hipSuccess) e = hipFree(p);
            if (e != hipSuccess) { (void)hipFree(q); return og::device_fail(e, "DeviceBuf::reserve"); }
        }
        p = q; cap = want;
        return OPUSGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class T>
struct PinnedBuf {                       // grow-only pinned host array (H2D source)
    T *p = nullptr;
    size_t cap = 0, n = 0;
    int reserve(size_t want)
    {
        if (want <= cap) return OPUSGPU_OK;
        size_t c = std::max(want, cap * 2);
        T *q = nullptr;
        OG_HIP(hipHostMalloc((void **)&q, c * sizeof(T), hipHostMallocDefault));
        if (p) { std::memcpy(q, p, n * sizeof(T)); (void)hipHostFree(p); }
        p = q; cap = c;
        return OPUSGPU_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
};

struct ParsedPacket {                    // one thread-local parse result
    int rc = 0;                          // number of frames or error
    uint32_t stream = 0;
    int worker = 0;                      // whose local arrays fi[].celt_index / silk_index point into
    OgFrameInfo fi[48];
    size_t celt0 = 0, leaf0 = 0, silk0 = 0;   // offsets into the worker's local arrays
    size_t ncelt = 0, nleaf = 0, nsilk = 0;
};

struct Worker {
    std::vector<OgCeltHdr> celt;
    std::vector<OgCeltLeaf> leaves;
    std::vector<OgSilkFrame> silk;
    std::vector<uint32_t> scount;            // SILK frames per depth (k-th SILK frame of its stream in this send) among this worker's
                                             // packets, then: where its first of that depth goes
    uint32_t nruns = 0;                      // streams of this worker with at least one CELT record in this send
    void reach(size_t depth) { if (scount.size() < depth) scount.resize(depth, 0); }
};

// Parse threads of a batch: started once, parked on a condition variable between batch_send calls (a send of 4096 packets
// is ~5 ms of parsing on 16 threads: creating and joining the threads every call showed up next to it).
// CPUs local to the GPU (a rank's parse threads stay on its GPU's NUMA node, so that eight ranks on one host do
// not drag each other's side-info across sockets): the PCI device's local_cpulist from sysfs, cut to what the process may
// run on.  Empty (no pinning) when anything is missing or OPUSGPU_PIN=0.
static std::vector<int> gpu_local_cpus(int device)
{
    std::vector<int> cpus;
    if (const char *e = std::getenv("OPUSGPU_PIN")) if (e[0] == '0') return cpus;
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return cpus;
    for (char *c = bdf; *c; c++) *c = char(std::tolower(*c));
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    std::FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return cpus;
    char line[4096] = {0};
    const bool got = std::fgets(line, sizeof line, f) != nullptr;
    std::fclose(f);
    if (!got) return cpus;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return cpus;
    for (const char *q = line; *q && *q != '\n';) {       // "0-47,96-143"
        char *end = nullptr;
        long a = std::strtol(q, &end, 10), z = a;
        if (end == q) break;
        if (*end == '-') { const char *r = end + 1; z = std::strtol(r, &end, 10); if (end == r) break; }
        for (long c = a; c <= z && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET(int(c), &allowed)) cpus.push_back(int(c));
        q = *end == ',' ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return cpus;
}

class ParsePool {
public:
    ParsePool(int n, int device) : n_(n)
    {
        const std::vector<int> cpus = gpu_local_cpus(device);
        for (int w = 1; w < n_; w++) {
            threads_.emplace_back([this, w] { loop(w); });
            if (!cpus.empty()) {                             // anywhere on the GPU's node: the scheduler places them inside it
                cpu_set_t set;
                CPU_ZERO(&set);
                for (int c : cpus) CPU_SET(c, &set);
                (void)pthread_setaffinity_np(threads_.back().native_handle(), sizeof set, &set);
            }
        }
        pinned_ = int(cpus.size());
    }
    int pinned_cpus() const { return pinned_; }
    ~ParsePool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    int size() const { return n_; }
    // job(w) for w in [0, nw), nw <= size(); the caller runs w = 0 itself and returns when all are done
    void run(int nw, const std::function<void(int)> &job)
    {
        if (nw <= 1) { job(0); return; }
        { std::lock_guard<std::mutex> lk(mu_); job_ = &job; nw_ = nw; left_ = nw - 1; gen_++; }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return left_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int w)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (w < nw_) job = job_;
            }
            if (job) {
                (*job)(w);
                std::lock_guard<std::mutex> lk(mu_);
                if (--left_ == 0) done_.notify_one();
            }
        }
    }
    int n_, nw_ = 0, left_ = 0, pinned_ = 0;
    bool stop_ = false;
    uint64_t gen_ = 0;
    const std::function<void(int)> *job_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
};

int worker_count()
{
    int n = 0;
    if (const char *e = std::getenv("OPUSGPU_THREADS")) n = std::atoi(e);
    if (n <= 0) n = int(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    return std::min(n, 64);
}

}  // namespace

struct opusgpu_batch {
    int device = 0, channels = 2;
    uint32_t nstreams = 0;
    hipStream_t stream = nullptr;
    std::vector<og::StreamParser> parsers;
    std::vector<uint32_t> nsamples;
    std::vector<uint8_t> seen;
    std::unique_ptr<ParsePool> pool;                        // parse threads and their arenas, kept between batch_send calls
    std::vector<Worker> workers;
    std::vector<ParsedPacket> parsed;
    std::vector<uint32_t> key_count;                        // [worker][og::K1_NKEYS + 256]: the threaded K1 sort's histograms

    // side-info of the last batch_send, frames ordered by round (frame index inside the packet)
    PinnedBuf<OgCeltHdr> h_celt;
    PinnedBuf<OgCeltLeaf> h_leaves;
    PinnedBuf<OgSilkFrame> h_silk;
    PinnedBuf<OgLeafRef> h_refs;                            // PVQ leaves per round, sorted by size (K1's work list)
    uint32_t ref_classes[4] = {0, 0, 0, 0};                 // leaves per K1 size class
    std::vector<uint32_t> silk_round_off;                   // [depths + 1]: SILK frames by depth (a stream's k-th SILK frame of the send)
    PinnedBuf<uint32_t> h_runs;                             // CELT records of stream run r: [h_runs[r], h_runs[r + 1])
    uint32_t nruns = 0, run_len = 0;                        // run_len != 0: every run has that many records (no offsets needed)

    DeviceBuf<OgCeltHdr> d_celt;
    DeviceBuf<OgCeltLeaf> d_leaves;
    DeviceBuf<OgSilkFrame> d_silk;
    DeviceBuf<OgLeafRef> d_refs;
    DeviceBuf<uint32_t> d_runs;
    DeviceBuf<uint8_t> d_work;                              // K1 -> K2 scratch, kept zero-filled between launches
    OgCeltState *d_celt_state = nullptr;
    void *d_silk_state = nullptr;
    float *d_pcm = nullptr;
    size_t pcm_stride = 0;
    bool pending = false;
    hipStream_t side = nullptr;                             // the stateless CELT kernels of a mixed send run here, beside its SILK launches
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t h2d_done = nullptr;                          // recorded behind the last H2D copy of batch_decode
    bool h2d_in_flight = false;                             // the pinned side-info may still be read by the copy engine

    ~opusgpu_batch()
    {
        (void)hipSetDevice(device);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        if (h2d_done) (void)hipEventDestroy(h2d_done);
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        h_celt.release(); h_leaves.release(); h_silk.release(); h_refs.release(); h_runs.release();
        d_celt.release(); d_leaves.release(); d_silk.release(); d_refs.release(); d_runs.release(); d_work.release();
        if (d_celt_state) (void)hipFree(d_celt_state);
        if (d_silk_state) (void)hipFree(d_silk_state);
        if (d_pcm) (void)hipFree(d_pcm);
    }
};

// The second half of a send (batch_send_impl), on the parse threads.  What the device gets:
//   * the send's CELT records as ONE launch, laid out [stream][time] -- the order in which the workers hold them -- with the
//     stream runs' offsets (K1, the plan and K2 are stateless: only the tail kernel walks a stream's records in order);
//   * its SILK frames by depth: the k-th SILK frame of every stream forms launch k (a stream's SILK frames are sequential in
//     time; none depends on a CELT record, whereas a hybrid frame's CELT part adds onto the SILK output: all SILK launches
//     run before the CELT one).
// Every worker copies its leaves and records to their places in the pinned arrays and counts its PVQ leaves by key; the K1
// work list (opusgpu_sort_pvq_leaves' order exactly: key descending, then frame, then leaf) then comes from one prefix pass
// over the (key, worker) counts and a scatter by every worker of its own leaves.  Returns the number of refs.
static uint32_t merge_send(opusgpu_batch *b, uint32_t n, uint32_t T, int nw, const uint32_t *stream_ids, int32_t *errors,
                           const std::vector<size_t> &leaf_base, const std::vector<size_t> &celt_base, const std::vector<uint32_t> &run_base)
{
    using namespace og;
    constexpr size_t ROW = K1_NKEYS + 256;                   // a worker's histogram: keys, then leaves per dimension n
    std::vector<Worker> &workers = b->workers;
    if (b->key_count.size() < ROW * size_t(nw)) b->key_count.resize(ROW * size_t(nw));
    const std::function<void(int)> place = [&](int w) {
        Worker &W = workers[w];
        uint32_t *cnt = b->key_count.data() + ROW * size_t(w);
        std::memset(cnt, 0, ROW * sizeof(uint32_t));
        if (!W.leaves.empty()) std::memcpy(b->h_leaves.p + leaf_base[w], W.leaves.data(), W.leaves.size() * sizeof(OgCeltLeaf));
        const uint32_t lo = uint32_t(uint64_t(n) * w / nw), hi = uint32_t(uint64_t(n) * (w + 1) / nw);
        OgCeltHdr *out = b->h_celt.p + celt_base[w];
        uint32_t *sfill = W.scount.data();
        uint32_t *runs = b->h_runs.p + run_base[w];
        size_t j = 0;                                        // records of this worker placed so far (= index into W.celt)
        for (uint32_t i = lo; i < hi; i++) {
            uint32_t pcm_off = 0, depth = 0;
            const uint32_t stream = stream_ids ? stream_ids[i] : i;
            const size_t j0 = j;
            for (uint32_t t = 0; t < T; t++) {
                const ParsedPacket &pp = b->parsed[size_t(i) * T + t];
                if (errors) errors[size_t(i) * T + t] = pp.rc < 0 ? pp.rc : OPUSGPU_OK;
                for (int f0 = 0; f0 < pp.rc; f0++) {
                    const OgFrameInfo &fi = pp.fi[f0];
                    uint32_t sub = 0;                        // SILK frames of one Opus frame are consecutive 20 ms pieces
                    for (uint32_t k = 0; k < fi.nsilk; k++, depth++) {
                        OgSilkFrame &sf = b->h_silk.p[sfill[depth]++];
                        sf = W.silk[fi.silk_index + k];
                        sf.stream = stream;
                        sf.pcm_off = pcm_off + sub;
                        sub += uint32_t(sf.f_size) * (sf.bandwidth == 0 ? 6u : sf.bandwidth == 1 ? 4u : 3u);
                    }
                    for (uint32_t c = 0; c < fi.has_celt; c++, j++) {      // (records were emitted in the order they must run)
                        OgCeltHdr &hd = out[j];
                        hd = W.celt[fi.celt_index + c];      // (== j: records are placed in the order they were emitted)
                        hd.stream = stream;
                        hd.pcm_off += pcm_off;               // the parser left the offset inside the Opus frame
                        const OgCeltLeaf *lf = W.leaves.data() + hd.leaf_begin;
                        for (uint32_t l = 0; l < hd.nleaves; l++)
                            if (k1_wanted(hd, lf[l])) { cnt[k1_key(lf[l])]++; cnt[K1_NKEYS + lf[l].n]++; }
                        hd.leaf_begin = uint32_t(hd.leaf_begin + leaf_base[w]);
                    }
                    pcm_off += fi.frame_size;
                }
            }
            b->nsamples[stream] = pcm_off;
            if (j > j0) *runs++ = uint32_t(celt_base[w] + j0);
        }
    };
    b->pool->run(nw, place);
    // counts -> first output index of every (key, worker): keys from the largest down, inside a key the workers in order
    uint32_t total = 0;
    for (int d = 255; d >= 0; d--) {
        uint32_t any = 0;                                    // few of the 256 dimensions occur: skip the keys of the others
        for (int w = 0; w < nw; w++) any |= b->key_count[ROW * size_t(w) + K1_NKEYS + size_t(d)];
        if (!any) continue;
        for (int k = d * K1_KQ + K1_KQ - 1; k >= d * K1_KQ; k--)
            for (int w = 0; w < nw; w++) {
                uint32_t &c = b->key_count[ROW * size_t(w) + size_t(k)];
                const uint32_t here = c;
                c = total;
                total += here;
            }
    }
    for (int c = 0; c < 4; c++) b->ref_classes[c] = 0;
    for (int w = 0; w < nw; w++)
        for (int d = 0; d < 256; d++) b->ref_classes[k1_class(d)] += b->key_count[ROW * size_t(w) + K1_NKEYS + d];
    const std::function<void(int)> scatter = [&](int w) {
        uint32_t *at = b->key_count.data() + ROW * size_t(w);
        const OgCeltHdr *hd = b->h_celt.p + celt_base[w];
        const size_t nc = workers[w].celt.size();
        for (size_t j = 0; j < nc; j++) {
            const OgCeltLeaf *lf = b->h_leaves.p + hd[j].leaf_begin;
            for (uint32_t l = 0; l < hd[j].nleaves; l++)
                if (k1_wanted(hd[j], lf[l])) b->h_refs.p[at[k1_key(lf[l])]++] = k1_ref(hd[j], hd[j].leaf_begin + l, uint32_t(celt_base[w] + j));
        }
    };
    b->pool->run(nw, scatter);
    return total;
}

extern "C" {

int opusgpu_batch_create(opusgpu_batch **out, uint32_t nstreams, int channels, int device)
{
    if (!out || nstreams == 0) return OPUSGPU_BAD_ARG;
    if (channels > 2) return OPUSGPU_UNSUPPORTED;       // decoder.rs:170-171
    if (channels < 1) return OPUSGPU_CONFIG_INVALID;
    *out = nullptr;
    int ndev = 0;
    OG_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return OPUSGPU_BAD_ARG;
    OG_HIP(hipSetDevice(device));
    int rc = opusgpu_init(device);
    if (rc) return rc;
    opusgpu_batch *b = new (std::nothrow) opusgpu_batch;
    if (!b) return OPUSGPU_BAD_ARG;
    b->device = device; b->channels = channels; b->nstreams = nstreams;
    b->parsers.reserve(nstreams);
    for (uint32_t s = 0; s < nstreams; s++) b->parsers.emplace_back(channels);
    b->nsamples.assign(nstreams, 0);
    b->seen.assign(nstreams, 0);
    b->pcm_stride = size_t(OPUSGPU_MAX_FRAME) * channels;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b->h2d_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b->join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_celt_state, size_t(nstreams) * sizeof(OgCeltState));
    if (e == hipSuccess) e = hipMemsetAsync(b->d_celt_state, 0, size_t(nstreams) * sizeof(OgCeltState), b->stream);
    if (e == hipSuccess) e = hipMalloc(&b->d_silk_state, opusgpu_silk_state_bytes(nstreams));
    if (e == hipSuccess) e = hipMemsetAsync(b->d_silk_state, 0, opusgpu_silk_state_bytes(nstreams), b->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_pcm, size_t(nstreams) * b->pcm_stride * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(b->d_pcm, 0, size_t(nstreams) * b->pcm_stride * sizeof(float), b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) { delete b; return og::device_fail(e, "batch_create"); }
    *out = b;
    return OPUSGPU_OK;
}

void opusgpu_batch_destroy(opusgpu_batch *b) { delete b; }

// n streams, T consecutive packets each: packet t of the i-th stream of the call is pkts[i * T + t]
static int batch_send_impl(opusgpu_batch *b, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                           uint32_t n, uint32_t T, int32_t *errors, bool fec = false)
{
    if (!b || T == 0 || (n && (!pkts || !lens))) return OPUSGPU_BAD_ARG;
    std::fill(b->seen.begin(), b->seen.end(), 0);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t s = stream_ids ? stream_ids[i] : i;
        if (s >= b->nstreams || b->seen[s]) return OPUSGPU_BAD_ARG;
        b->seen[s] = 1;
    }
    // batch_decode is asynchronous: its H2D copies read the pinned side-info arrays this call is about to rewrite.  Wait
    // for the copies (not for the kernels) of the previous decode before touching them.
    if (b->h2d_in_flight) {
        OG_HIP(hipSetDevice(b->device));
        OG_HIP(hipEventSynchronize(b->h2d_done));
        b->h2d_in_flight = false;
    }
    std::fill(b->nsamples.begin(), b->nsamples.end(), 0);
    if (T > 1) {                                        // a stream's PCM row holds the whole send: T packets of up to 120 ms
        const size_t need = size_t(T) * OPUSGPU_MAX_FRAME * b->channels;
        if (need > b->pcm_stride) {
            OG_HIP(hipSetDevice(b->device));
            OG_HIP(hipStreamSynchronize(b->stream));
            float *np = nullptr;
            OG_HIP(hipMalloc((void **)&np, size_t(b->nstreams) * need * sizeof(float)));
            OG_HIP(hipMemsetAsync(np, 0, size_t(b->nstreams) * need * sizeof(float), b->stream));
            (void)hipFree(b->d_pcm);
            b->d_pcm = np;
            b->pcm_stride = need;
        }
    }

    // OPUSGPU_TIMING=1: one line per send on stderr, the milliseconds of its three phases (parse on the pool / merge / K1 sort)
    static const bool timing = std::getenv("OPUSGPU_TIMING") != nullptr;
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    const auto t_begin = now();

    // ---- parse: static partition of the streams over the workers, one stream per task
    if (!b->pool) b->pool.reset(new ParsePool(worker_count(), b->device));
    const int nw = std::max(1, std::min<int>(b->pool->size(), int(n)));
    std::vector<Worker> &workers = b->workers;
    std::vector<ParsedPacket> &parsed = b->parsed;
    if (int(workers.size()) < nw) workers.resize(nw);
    if (parsed.size() < size_t(n) * T) parsed.resize(size_t(n) * T);
    for (auto &W : workers) { W.celt.clear(); W.leaves.clear(); W.silk.clear(); W.scount.clear(); W.nruns = 0; }   // capacity stays
    const std::function<void(int)> run = [&](int w) {
        Worker &W = workers[w];
        const uint32_t lo = uint32_t(uint64_t(n) * w / nw), hi = uint32_t(uint64_t(n) * (w + 1) / nw);
        for (uint32_t i = lo; i < hi; i++) {
            uint32_t depth = 0;                         // SILK frames of this stream so far in the send
            const size_t c0 = W.celt.size();
            for (uint32_t t = 0; t < T; t++) {
                ParsedPacket &pp = parsed[size_t(i) * T + t];
                pp.stream = stream_ids ? stream_ids[i] : i;
                pp.worker = w;
                pp.celt0 = W.celt.size(); pp.leaf0 = W.leaves.size(); pp.silk0 = W.silk.size();
                og::StreamParser &sp = b->parsers[pp.stream];
                pp.rc = fec ? sp.parse_fec(pkts[size_t(i) * T + t], lens[size_t(i) * T + t], pp.fi, W.celt, W.leaves, W.silk)
                            : sp.parse(pkts[size_t(i) * T + t], lens[size_t(i) * T + t], pp.fi, W.celt, W.leaves, W.silk);
                if (pp.rc < 0) {                        // a bad packet never poisons the batch
                    W.celt.resize(pp.celt0); W.leaves.resize(pp.leaf0); W.silk.resize(pp.silk0);
                } else {
                    for (int f = 0; f < pp.rc; f++) {
                        W.reach(depth + pp.fi[f].nsilk);
                        for (uint32_t k = 0; k < pp.fi[f].nsilk; k++) W.scount[depth++]++;
                    }
                }
                pp.ncelt = W.celt.size() - pp.celt0; pp.nleaf = W.leaves.size() - pp.leaf0; pp.nsilk = W.silk.size() - pp.silk0;
            }
            if (W.celt.size() > c0) W.nruns++;
        }
    };
    b->pool->run(nw, run);
    const auto t_parsed = now();

    // ---- merge: where every worker's leaves, CELT records, stream runs and SILK frames (by depth) go
    std::vector<size_t> leaf_base(nw + 1, 0), celt_base(nw + 1, 0);
    std::vector<uint32_t> run_base(nw + 1, 0);
    size_t tot_silk = 0, depths = 0;
    for (int w = 0; w < nw; w++) {
        leaf_base[w + 1] = leaf_base[w] + workers[w].leaves.size();
        celt_base[w + 1] = celt_base[w] + workers[w].celt.size();
        run_base[w + 1] = run_base[w] + workers[w].nruns;
        tot_silk += workers[w].silk.size();
        depths = std::max(depths, workers[w].scount.size());
    }
    const size_t tot_celt = celt_base[nw];
    if (tot_celt > OG_REF_FRAME_MASK) return OPUSGPU_BAD_ARG;
    OG_HIP(hipSetDevice(b->device));
    int rc;
    if ((rc = b->h_leaves.reserve(leaf_base[nw] + 1))) return rc;
    if ((rc = b->h_celt.reserve(tot_celt + 1))) return rc;
    if ((rc = b->h_silk.reserve(tot_silk + 1))) return rc;
    if ((rc = b->h_refs.reserve(leaf_base[nw] + 1))) return rc;
    if ((rc = b->h_runs.reserve(size_t(run_base[nw]) + 2))) return rc;
    b->silk_round_off.assign(depths + 1, 0);
    for (size_t d = 0; d < depths; d++) {               // counts -> every worker's first index at every depth
        uint32_t k = b->silk_round_off[d];
        for (int w = 0; w < nw; w++) {
            Worker &W = workers[w];
            W.reach(depths);
            const uint32_t wk = W.scount[d];
            W.scount[d] = k;
            k += wk;
        }
        b->silk_round_off[d + 1] = k;
    }
    const uint32_t nref = merge_send(b, n, T, nw, stream_ids, errors, leaf_base, celt_base, run_base);
    b->h_leaves.n = leaf_base[nw];
    b->h_celt.n = tot_celt;
    b->h_silk.n = tot_silk;
    b->h_refs.n = nref;
    b->nruns = run_base[nw];
    b->h_runs.p[b->nruns] = uint32_t(tot_celt);
    b->h_runs.n = size_t(b->nruns) + 1;
    // runs of one length (every steady batch: one CELT frame per stream, or T F of them -- several packets per stream and / or
    // code 1-3 packets) need no offsets: the tail kernel's fixed groups
    b->run_len = b->nruns ? uint32_t(tot_celt / b->nruns) : 0;
    if (b->run_len * size_t(b->nruns) != tot_celt) b->run_len = 0;
    for (uint32_t r = 0; r < b->nruns && b->run_len; r++)
        if (b->h_runs.p[r] != r * b->run_len) b->run_len = 0;
    b->pending = true;
    if (std::getenv("OPUSGPU_CHECK_SORT") && tot_celt) {    // self-check (tests): the one-thread sort of the same records must agree
        std::vector<OgLeafRef> want(b->h_leaves.n + 1);
        uint32_t cls[4];
        const uint32_t nw2 = opusgpu_sort_pvq_leaves(b->h_celt.p, uint32_t(tot_celt), b->h_leaves.p, want.data(), uint32_t(want.size()), cls);
        if (nw2 != nref || std::memcmp(want.data(), b->h_refs.p, size_t(nref) * sizeof(OgLeafRef)) != 0 ||
            std::memcmp(cls, b->ref_classes, sizeof cls) != 0) {
            std::fprintf(stderr, "opusgpu: the threaded K1 sort differs from opusgpu_sort_pvq_leaves\n");
            return OPUSGPU_DEVICE;
        }
    }
    if (timing)
        std::fprintf(stderr, "opusgpu send: %u x %u packets, %d threads: parse %.3f ms, merge + sort on the threads %.3f ms\n", n, T, nw,
                     ms(t_begin, t_parsed), ms(t_parsed, now()));
    return OPUSGPU_OK;
}

extern "C" int opusgpu_batch_send(opusgpu_batch *b, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                                  uint32_t n, int32_t *errors)
{
    return batch_send_impl(b, pkts, lens, stream_ids, n, 1, errors);
}

extern "C" int opusgpu_batch_send_fec(opusgpu_batch *b, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                                      uint32_t n, int32_t *errors)
{
    return batch_send_impl(b, pkts, lens, stream_ids, n, 1, errors, true);
}

extern "C" int opusgpu_batch_send_multi(opusgpu_batch *b, const uint8_t *const *pkts, const uint32_t *lens, const uint32_t *stream_ids,
                                        uint32_t nstreams, uint32_t packets_per_stream, int32_t *errors)
{
    return batch_send_impl(b, pkts, lens, stream_ids, nstreams, packets_per_stream, errors);
}

int opusgpu_batch_decode(opusgpu_batch *b)
{
    if (!b) return OPUSGPU_BAD_ARG;
    if (!b->pending) return OPUSGPU_OK;
    OG_HIP(hipSetDevice(b->device));
    int rc;
    const bool offsets = b->nruns && !b->run_len;       // runs of different lengths: the tail kernel needs their offsets
    // (every device buffer grows before the first launch of this call: growing frees the old block, see DeviceBuf)
    const size_t work_need = opusgpu_celt_work_bytes(uint32_t(b->h_celt.n)), work_had = b->d_work.cap;
    if ((rc = b->d_celt.reserve(b->h_celt.n + 1, b->stream)) || (rc = b->d_leaves.reserve(b->h_leaves.n + 1, b->stream)) ||
        (rc = b->d_silk.reserve(b->h_silk.n + 1, b->stream)) || (rc = b->d_refs.reserve(b->h_refs.n + 1, b->stream)) ||
        (rc = b->d_runs.reserve(b->h_runs.n + 1, b->stream)) || (rc = b->d_work.reserve(work_need, b->stream)))
        return rc;
    if (b->h_celt.n) OG_HIP(hipMemcpyAsync(b->d_celt.p, b->h_celt.p, b->h_celt.n * sizeof(OgCeltHdr), hipMemcpyHostToDevice, b->stream));
    if (b->h_leaves.n) OG_HIP(hipMemcpyAsync(b->d_leaves.p, b->h_leaves.p, b->h_leaves.n * sizeof(OgCeltLeaf), hipMemcpyHostToDevice, b->stream));
    if (b->h_silk.n) OG_HIP(hipMemcpyAsync(b->d_silk.p, b->h_silk.p, b->h_silk.n * sizeof(OgSilkFrame), hipMemcpyHostToDevice, b->stream));
    if (b->h_refs.n) OG_HIP(hipMemcpyAsync(b->d_refs.p, b->h_refs.p, b->h_refs.n * sizeof(OgLeafRef), hipMemcpyHostToDevice, b->stream));
    if (offsets) OG_HIP(hipMemcpyAsync(b->d_runs.p, b->h_runs.p, b->h_runs.n * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
    OG_HIP(hipEventRecord(b->h2d_done, b->stream));
    b->h2d_in_flight = true;
    if (b->d_work.cap != work_had) OG_HIP(hipMemsetAsync(b->d_work.p, 0, b->d_work.cap, b->stream));   // zero once per (re)allocation
    // SILK by depth (a stream's frames stay in order), and every CELT record of the send in one launch: decoder.rs:72-123's
    // per-frame loop, cut where the data dependences are instead of where the packet boundaries are.  K1, the plan and the
    // spectrum kernel touch neither decoder state nor PCM: in a send that also has SILK frames they run on a side stream beside
    // the SILK launches (which are latency-bound and leave most of the chip idle); the tail kernel, which adds hybrid frames'
    // CELT part onto the SILK output, follows both.
    const uint32_t cn = uint32_t(b->h_celt.n);
    const bool beside = cn && b->h_silk.n;
    if (beside) {
        OG_HIP(hipEventRecord(b->fork, b->stream));
        OG_HIP(hipStreamWaitEvent(b->side, b->fork, 0));
        if ((rc = opusgpu_launch_celt_pvq(b->d_celt.p, b->d_leaves.p, cn, b->d_refs.p, b->ref_classes, b->d_work.p, nullptr, b->side))) return rc;
        if ((rc = opusgpu_launch_celt_spectrum(b->d_celt.p, cn, b->d_work.p, nullptr, b->side))) return rc;
        OG_HIP(hipEventRecord(b->join, b->side));
    }
    for (size_t d = 0; d + 1 < b->silk_round_off.size(); d++) {
        const uint32_t s0 = b->silk_round_off[d], sn = b->silk_round_off[d + 1] - s0;
        if (sn && (rc = opusgpu_launch_silk(b->d_silk.p + s0, sn, b->d_silk_state, b->nstreams, b->d_pcm, b->pcm_stride, nullptr, b->stream)))
            return rc;
    }
    if (cn) {
        if (beside) OG_HIP(hipStreamWaitEvent(b->stream, b->join, 0));
        else {
            if ((rc = opusgpu_launch_celt_pvq(b->d_celt.p, b->d_leaves.p, cn, b->d_refs.p, b->ref_classes, b->d_work.p, nullptr, b->stream))) return rc;
            if ((rc = opusgpu_launch_celt_spectrum(b->d_celt.p, cn, b->d_work.p, nullptr, b->stream))) return rc;
        }
        if (offsets)
            rc = opusgpu_launch_celt_tail_runs(b->d_celt.p, cn, b->d_runs.p, b->nruns, b->d_work.p, b->d_celt_state, b->d_pcm, b->pcm_stride,
                                               nullptr, b->stream);
        else
            rc = opusgpu_launch_celt_tail(b->d_celt.p, cn, b->run_len, b->d_work.p, b->d_celt_state, b->d_pcm, b->pcm_stride, nullptr, b->stream);
        if (rc) return rc;
    }
    b->pending = false;
    return OPUSGPU_OK;
}

int opusgpu_batch_sync(opusgpu_batch *b)
{
    if (!b) return OPUSGPU_BAD_ARG;
    OG_HIP(hipSetDevice(b->device));
    OG_HIP(hipStreamSynchronize(b->stream));
    return OPUSGPU_OK;
}

int opusgpu_batch_pcm(opusgpu_batch *b, const float **d_pcm, size_t *stride_floats, const uint32_t **nsamples)
{
    if (!b) return OPUSGPU_BAD_ARG;
    if (d_pcm) *d_pcm = b->d_pcm;
    if (stride_floats) *stride_floats = b->pcm_stride;
    if (nsamples) *nsamples = b->nsamples.data();
    return OPUSGPU_OK;
}

int opusgpu_batch_read(opusgpu_batch *b, uint32_t stream, float *pcm, size_t cap_floats, size_t *nsamples)
{
    if (!b || stream >= b->nstreams || !pcm) return OPUSGPU_BAD_ARG;
    const size_t n = b->nsamples[stream], fl = n * b->channels;
    if (fl > cap_floats) return OPUSGPU_BAD_ARG;
    OG_HIP(hipSetDevice(b->device));
    if (fl) OG_HIP(hipMemcpyAsync(pcm, b->d_pcm + size_t(stream) * b->pcm_stride, fl * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    OG_HIP(hipStreamSynchronize(b->stream));
    if (nsamples) *nsamples = n;
    return OPUSGPU_OK;
}

int opusgpu_batch_flush(opusgpu_batch *b, uint32_t stream)
{
    if (!b || stream >= b->nstreams) return OPUSGPU_BAD_ARG;
    b->parsers[stream].reset();                         // next frames carry the RESET flags; device state follows
    b->nsamples[stream] = 0;
    return OPUSGPU_OK;
}

int opusgpu_batch_sideinfo(opusgpu_batch *b, const OgCeltHdr **celt, uint32_t *ncelt, const OgCeltLeaf **leaves,
                           uint32_t *nleaves, const OgSilkFrame **silk, uint32_t *nsilk)
{
    if (!b) return OPUSGPU_BAD_ARG;
    if (celt) *celt = b->h_celt.p;
    if (ncelt) *ncelt = uint32_t(b->h_celt.n);
    if (leaves) *leaves = b->h_leaves.p;
    if (nleaves) *nleaves = uint32_t(b->h_leaves.n);
    if (silk) *silk = b->h_silk.p;
    if (nsilk) *nsilk = uint32_t(b->h_silk.n);
    return OPUSGPU_OK;
}

int opusgpu_batch_plan(opusgpu_batch *b, OgBatchPlan *out)
{
    if (!b || !out) return OPUSGPU_BAD_ARG;
    out->refs = b->h_refs.p;
    out->nrefs = uint32_t(b->h_refs.n);
    for (int c = 0; c < 4; c++) out->class_counts[c] = b->ref_classes[c];
    out->run_off = b->h_runs.p;
    out->nruns = b->nruns;
    out->run_len = b->run_len;
    out->silk_depth_off = b->silk_round_off.data();
    out->ndepths = b->silk_round_off.empty() ? 0 : uint32_t(b->silk_round_off.size() - 1);
    return OPUSGPU_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Layer 1: the reference's one-stream Decoder (src/decoder.rs:16-188) = a batch of one + an output queue.
struct opusgpu_decoder {
    int device = 0, channels = 2;
    bool configured = false;
    std::vector<uint8_t> extradata;
    opusgpu_batch *batch = nullptr;
    std::deque<std::vector<float>> queue;
    uint32_t skip_left = 0;             // OpusHead pre-skip still to drop (48 kHz samples)
    uint32_t pre_skip = 0;
    float out_gain = 1.f;               // 10^(output_gain_q8 / (20 * 256))
    // multistream (RFC 7845 5.1.1, mapping families 1 and 255; decoder.rs:151-157, :170-171 stop at unimplemented!()): the
    // packet's `streams` Opus streams decode side by side -- the first `coupled` ones as the streams of a two-channel batch,
    // the others as the streams of a one-channel batch (`batch_mono`; a stereo-coded packet in a mono stream is down-mixed, as
    // by any mono decoder); output channel c is channel `mapping[c]` of the decoded set -- 2s, 2s+1 for coupled stream s,
    // then one per mono stream -- or silence for 255
    uint32_t streams = 1, coupled = 0;
    bool multi = false;                 // decode through the mapping table (false: family 0's implicit mono / stereo layout)
    std::vector<uint8_t> mapping;
    opusgpu_batch *batch_mono = nullptr;
    ~opusgpu_decoder() { delete batch; delete batch_mono; }
};

extern "C" {

int opusgpu_decoder_create(opusgpu_decoder **out, int device)
{
    if (!out) return OPUSGPU_BAD_ARG;
    *out = new (std::nothrow) opusgpu_decoder;
    if (!*out) return OPUSGPU_BAD_ARG;
    (*out)->device = device;
    return OPUSGPU_OK;
}

void opusgpu_decoder_destroy(opusgpu_decoder *d) { delete d; }

int opusgpu_decoder_set_extradata(opusgpu_decoder *d, const uint8_t *extra, size_t len)
{
    if (!d || (len && !extra)) return OPUSGPU_BAD_ARG;
    d->extradata.assign(extra, extra + len);            // decoder.rs:45-47
    return OPUSGPU_OK;
}

int opusgpu_decoder_configure(opusgpu_decoder *d)
{
    if (!d) return OPUSGPU_BAD_ARG;
    int channels = 2;                                   // decoder.rs:136-145
    const std::vector<uint8_t> &e = d->extradata;
    d->pre_skip = 0;
    d->out_gain = 1.f;
    if (!e.empty()) {
        OgOpusHead head;
        const int hrc = opusgpu_opus_head_parse(e.data(), e.size(), &head);
        // decoder.rs:151-171: explicit stream maps / > 2 channels end in unimplemented!() in the reference; here mapping
        // families 1 and 255 configure a multistream decoder
        if (hrc == OPUSGPU_CONFIG_INVALID && e.size() >= 19 && std::memcmp(e.data(), "OpusHead", 8) == 0 && e[9] > 2 && e[18] == 0)
            return OPUSGPU_UNSUPPORTED;                 // family 0 with > 2 channels: not a valid header (RFC 7845 5.1.1.1)
        if (hrc) return hrc;
        channels = head.channels;
        d->streams = head.stream_count;
        d->coupled = head.coupled_count;
        d->mapping.assign(head.mapping, head.mapping + head.channels);
        d->pre_skip = head.pre_skip;
        d->out_gain = head.output_gain_q8 ? std::pow(10.f, float(head.output_gain_q8) / (20.f * 256.f)) : 1.f;
    }
    delete d->batch;
    d->batch = nullptr;
    d->queue.clear();
    d->skip_left = d->pre_skip;
    d->channels = channels;
    if (e.empty()) { d->streams = 1; d->coupled = 1; d->mapping = {0, 1}; }
    delete d->batch_mono;
    d->batch_mono = nullptr;
    // the plain path is mapping family 0's implicit layout (one stream; mono, or one coupled pair in order); anything else --
    // family 1 / 255 with one stream and a mapping of its own included -- goes through the mapping table
    d->multi = !(d->streams == 1 && channels <= 2 && d->coupled == uint32_t(channels - 1) && d->mapping[0] == 0 &&
                 (channels == 1 || d->mapping[1] == 1));
    const bool multi = d->multi;
    int rc = OPUSGPU_OK;
    if (!multi) rc = opusgpu_batch_create(&d->batch, 1, channels, d->device);
    else {
        if (d->coupled) rc = opusgpu_batch_create(&d->batch, d->coupled, 2, d->device);
        if (!rc && d->streams > d->coupled) rc = opusgpu_batch_create(&d->batch_mono, d->streams - d->coupled, 1, d->device);
    }
    d->configured = rc == OPUSGPU_OK;
    return rc;
}

// One multistream packet: `streams` Opus packets back to back, all but the last self-delimiting (RFC 7845 5.1.1 / RFC 6716
// Appendix B), decoded side by side as the streams of the batch and dealt to the output channels by the mapping table.
static int send_multistream(opusgpu_decoder *d, const uint8_t *pkt, size_t len, std::vector<float> &out, size_t *nsamples)
{
    const uint32_t S = d->streams;
    std::vector<std::vector<uint8_t>> sub(S);
    std::vector<const uint8_t *> pk(S, nullptr);
    std::vector<uint32_t> ln(S, 0);
    std::vector<int32_t> err(S, 0);
    if (len) {                                          // (len == 0: the whole packet is lost -- every stream conceals)
        size_t pos = 0;
        for (uint32_t s = 0; s < S; s++) {
            if (s + 1 < S) {
                int used = 0;
                const int rc = og::undelimit_packet(pkt + pos, int(std::min<size_t>(len - pos, 0x7FFFFFFF)), sub[s], &used);
                if (rc) return rc;
                pos += size_t(used);
                if (pos >= len) return OPUSGPU_INVALID_DATA;
            } else {
                sub[s].assign(pkt + pos, pkt + len);
            }
            pk[s] = sub[s].data();
            ln[s] = uint32_t(sub[s].size());
        }
    }
    const uint32_t M = d->coupled;
    int rc = OPUSGPU_OK;
    if (M && (rc = opusgpu_batch_send(d->batch, pk.data(), ln.data(), nullptr, M, err.data()))) return rc;
    if (S > M && (rc = opusgpu_batch_send(d->batch_mono, pk.data() + M, ln.data() + M, nullptr, S - M, err.data() + M))) return rc;
    int bad = OPUSGPU_OK;
    for (uint32_t s = 0; s < S && !bad; s++) bad = err[s];
    if (bad) {
        // One stream's packet did not parse: the streams that did have advanced their host state, so they are decoded (their
        // device state follows) and the bad ones conceal -- host and device stay in step on every stream; the caller gets the
        // error and no frame for this packet.
        std::fill(ln.begin(), ln.end(), 0u);
        std::vector<uint32_t> ids;
        for (int half = 0; half < 2; half++) {
            opusgpu_batch *bt = half ? d->batch_mono : d->batch;
            const uint32_t lo = half ? M : 0, hi = half ? S : M;
            if (!bt || hi == lo) continue;
            (void)opusgpu_batch_decode(bt);
            ids.clear();
            for (uint32_t s = lo; s < hi; s++) if (err[s]) ids.push_back(s - lo);
            if (ids.empty()) continue;
            std::vector<int32_t> e2(ids.size(), 0);
            if (opusgpu_batch_send(bt, pk.data(), ln.data(), ids.data(), uint32_t(ids.size()), e2.data()) == OPUSGPU_OK) (void)opusgpu_batch_decode(bt);
        }
        return bad;
    }
    if (M && (rc = opusgpu_batch_decode(d->batch))) return rc;             // both batches' kernels are in flight before the first read
    if (S > M && (rc = opusgpu_batch_decode(d->batch_mono))) return rc;
    std::vector<float> dec(size_t(OPUSGPU_MAX_FRAME) * 2);
    size_t n = 0;
    for (uint32_t s = 0; s < S; s++) {
        size_t ns = 0;
        const int cs = s < M ? 2 : 1;                   // channels of this stream's decoder
        if ((rc = opusgpu_batch_read(s < M ? d->batch : d->batch_mono, s < M ? s : s - M, dec.data(), dec.size(), &ns))) return rc;
        if (s == 0) { n = ns; out.assign(n * d->channels, 0.f); }
        else if (ns != n) return OPUSGPU_INVALID_DATA;  // RFC 7845 5.1.1: every stream of a packet has the same duration
        for (int c = 0; c < d->channels; c++) {
            const uint32_t m = d->mapping[c];
            if (m == 255) continue;
            const uint32_t src = m < 2 * M ? m >> 1 : m - M, chn = m < 2 * M ? m & 1 : 0;
            if (src != s) continue;
            for (size_t i = 0; i < n; i++) out[i * d->channels + c] = dec[cs * i + chn];
        }
    }
    *nsamples = n;
    return OPUSGPU_OK;
}

int opusgpu_decoder_send_packet(opusgpu_decoder *d, const uint8_t *pkt, size_t len)
{
    if (!d) return OPUSGPU_BAD_ARG;
    if (!d->configured) return OPUSGPU_CONFIG_INCOMPLETE;   // the reference unwraps None here (decoder.rs:49-50)
    if (len > 0xFFFFFFFFu) return OPUSGPU_INVALID_DATA;
    std::vector<float> out;
    size_t n = 0;
    int rc;
    if (d->multi) {
        if ((rc = send_multistream(d, pkt, len, out, &n))) return rc;
    } else {
        const uint8_t *pk[1] = {pkt};
        uint32_t ln[1] = {uint32_t(len)};
        int32_t err = 0;
        rc = opusgpu_batch_send(d->batch, pk, ln, nullptr, 1, &err);
        if (rc) return rc;
        if (err) return err;
        if ((rc = opusgpu_batch_decode(d->batch))) return rc;
        out.resize(size_t(OPUSGPU_MAX_FRAME) * d->channels);
        if ((rc = opusgpu_batch_read(d->batch, 0, out.data(), out.size(), &n))) return rc;
        out.resize(n * d->channels);
    }
    if (d->skip_left) {                                 // RFC 7845 4.2: the first pre_skip samples are decoder warm-up
        const size_t drop = std::min<size_t>(d->skip_left, n);
        out.erase(out.begin(), out.begin() + drop * d->channels);
        d->skip_left -= uint32_t(drop);
    }
    if (d->out_gain != 1.f) for (float &v : out) v *= d->out_gain;
    d->queue.push_back(std::move(out));
    return OPUSGPU_OK;
}

int opusgpu_decoder_send_packet_fec(opusgpu_decoder *d, const uint8_t *pkt, size_t len)
{
    if (!d) return OPUSGPU_BAD_ARG;
    if (!d->configured) return OPUSGPU_CONFIG_INCOMPLETE;
    if (len > 0xFFFFFFFFu) return OPUSGPU_INVALID_DATA;
    if (d->multi) return OPUSGPU_UNSUPPORTED;           // (multistream: send the lost packet as len 0)
    const uint8_t *pk[1] = {pkt};
    uint32_t ln[1] = {uint32_t(len)};
    int32_t err = 0;
    int rc = opusgpu_batch_send_fec(d->batch, pk, ln, nullptr, 1, &err);
    if (rc) return rc;
    if (err) return err;
    if ((rc = opusgpu_batch_decode(d->batch))) return rc;
    std::vector<float> out(size_t(OPUSGPU_MAX_FRAME) * d->channels);
    size_t n = 0;
    if ((rc = opusgpu_batch_read(d->batch, 0, out.data(), out.size(), &n))) return rc;
    out.resize(n * d->channels);
    if (d->skip_left) {
        const size_t drop = std::min<size_t>(d->skip_left, n);
        out.erase(out.begin(), out.begin() + drop * d->channels);
        d->skip_left -= uint32_t(drop);
    }
    if (d->out_gain != 1.f) for (float &v : out) v *= d->out_gain;
    d->queue.push_back(std::move(out));
    return OPUSGPU_OK;
}

int opusgpu_decoder_receive_frame(opusgpu_decoder *d, float *pcm, size_t cap_floats, size_t *nsamples, int *channels)
{
    if (!d || !pcm) return OPUSGPU_BAD_ARG;
    if (d->queue.empty()) return OPUSGPU_MORE_DATA_NEEDED;
    std::vector<float> &f = d->queue.front();
    if (f.size() > cap_floats) return OPUSGPU_BAD_ARG;
    std::memcpy(pcm, f.data(), f.size() * sizeof(float));
    if (nsamples) *nsamples = f.size() / d->channels;
    if (channels) *channels = d->channels;
    d->queue.pop_front();
    return OPUSGPU_OK;
}

int opusgpu_decoder_flush(opusgpu_decoder *d)
{
    if (!d) return OPUSGPU_BAD_ARG;
    d->queue.clear();
    d->skip_left = d->pre_skip;
    int rc = OPUSGPU_OK;                                // every stream of both batches (multistream: coupled, then mono)
    for (opusgpu_batch *bt : {d->batch, d->batch_mono})
        for (uint32_t s = 0; bt && s < bt->nstreams && !rc; s++) rc = opusgpu_batch_flush(bt, s);
    return rc;
}

}  // extern "C"
