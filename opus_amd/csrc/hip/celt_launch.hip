// C ABI, layer 3b (include/opusgpu.h): raw CELT kernel launch on caller-owned device memory.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include "celt_frame.h"
#include "../host/celt_tables.hpp"
#include "device_util.hpp"

__global__ __launch_bounds__(OG_WG, 4) void og_celt_kernel(const OgCeltHdr *__restrict__ hdr, const OgCeltLeaf *__restrict__ leaves,
                                                        uint32_t nframes, uint32_t frames_per_group,
                                                        const OgCeltTables *__restrict__ T, OgCeltState *state, float *pcm,
                                                        size_t pcm_stride, OgCeltDebug dbg)
{
    __shared__ OgCeltLds L;
    const uint32_t f0 = blockIdx.x * frames_per_group;
    for (uint32_t t = 0; t < frames_per_group; t++) {
        const uint32_t f = f0 + t;
        if (f >= nframes) break;
        OgCeltDebug d;
        d.pulses = dbg.pulses ? dbg.pulses + (size_t)f * 1920 : nullptr;
        d.norm = dbg.norm ? dbg.norm + (size_t)f * 1920 : nullptr;
        d.freq = dbg.freq ? dbg.freq + (size_t)f * 1920 : nullptr;
        d.collapse = dbg.collapse ? dbg.collapse + (size_t)f * 42 : nullptr;
        d.cycles = dbg.cycles ? dbg.cycles + (size_t)f * 64 : nullptr;
        og_celt_frame(hdr + f, leaves, T, state, pcm, pcm_stride, L, d);
        __syncthreads();                                // frame t's state stores before frame t+1's loads
    }
}

namespace og {

thread_local std::string g_last_error;

int device_fail(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return OPUSGPU_DEVICE;
}

static std::mutex g_tables_mu;
static OgCeltTables *g_celt_tables[64];

int celt_tables_for_device(int device, const OgCeltTables **out)
{
    if (device < 0 || device >= 64) return OPUSGPU_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    if (!g_celt_tables[device]) {
        int prev = 0;
        OG_HIP(hipGetDevice(&prev));
        OG_HIP(hipSetDevice(device));
        OgCeltTables *host = new OgCeltTables, *dev = nullptr;
        og_fill_celt_tables(host);
        hipError_t e = hipMalloc(&dev, sizeof(OgCeltTables));
        if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(OgCeltTables), hipMemcpyHostToDevice);
        delete host;
        (void)hipSetDevice(prev);
        if (e != hipSuccess) return device_fail(e, "celt tables upload");
        g_celt_tables[device] = dev;
    }
    *out = g_celt_tables[device];
    return OPUSGPU_OK;
}

}  // namespace og

extern "C" {

const char *opusgpu_last_error(void) { return og::g_last_error.c_str(); }

int opusgpu_init(int device)
{
    const OgCeltTables *t;
    return og::celt_tables_for_device(device, &t);
}

int opusgpu_launch_celt(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                        OgCeltState *state, float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!hdr || !leaves || !state || !pcm || frames_per_group == 0) return OPUSGPU_BAD_ARG;
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    const OgCeltTables *T;
    int rc = og::celt_tables_for_device(device, &T);
    if (rc) return rc;
    OgCeltDebug dbg = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (debug) dbg = *debug;
    const uint32_t groups = (nframes + frames_per_group - 1) / frames_per_group;
    hipLaunchKernelGGL(og_celt_kernel, dim3(groups), dim3(OG_WG), 0, (hipStream_t)stream, hdr, leaves, nframes,
                       frames_per_group, T, state, pcm, pcm_stride_floats, dbg);
    OG_HIP(hipGetLastError());
    return OPUSGPU_OK;
}

}  // extern "C"
