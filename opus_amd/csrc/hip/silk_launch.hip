// C ABI, layer 3b (include/opusgpu.h): SILK kernel launches on caller-owned device memory.
#include <hip/hip_runtime.h>
#include <cmath>
#include <mutex>
#include "silk_synth.h"
#include "device_util.hpp"

__global__ __launch_bounds__(64) void og_silk_synth_kernel(const OgSilkFrame *__restrict__ frames, uint32_t nframes, OgSilkArrays A)
{
    __shared__ OgSilkLds L;
    og_silk_synth_wave(frames, nframes, A, blockIdx.x * OG_SK_SLOTS, L);
}

__global__ __launch_bounds__(OG_WG) void og_silk_output_kernel(const OgSilkFrame *__restrict__ frames, OgSilkArrays A,
                                                               const OgSilkTables *__restrict__ T, float *pcm, size_t pcm_stride,
                                                               float *dbg_internal)
{
    __shared__ OgSilkOutLds L;
    og_silk_output(frames + blockIdx.x, A, T, pcm, pcm_stride, L, dbg_internal ? dbg_internal + (size_t)blockIdx.x * 640 : nullptr);
}

void og_fill_silk_tables(OgSilkTables *t);

namespace og {

static std::mutex g_silk_mu;
static OgSilkTables *g_silk_tables[64];

static int silk_tables_for_device(int device, const OgSilkTables **out)
{
    if (device < 0 || device >= 64) return OPUSGPU_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_silk_mu);
    if (!g_silk_tables[device]) {
        OgSilkTables host, *dev = nullptr;
        og_fill_silk_tables(&host);
        hipError_t e = hipMalloc(&dev, sizeof host);
        if (e == hipSuccess) e = hipMemcpy(dev, &host, sizeof host, hipMemcpyHostToDevice);
        if (e != hipSuccess) return device_fail(e, "silk tables upload");
        g_silk_tables[device] = dev;
    }
    *out = g_silk_tables[device];
    return OPUSGPU_OK;
}

OgSilkArrays silk_arrays(void *state, uint32_t nstreams)
{
    OgSilkArrays A;
    const size_t ns = 2 * size_t(nstreams);
    float *p = (float *)state;
    A.slots = p; p += size_t(OG_SILK_SLOT) * ns;
    A.rs_hist = p;
    A.nslots = uint32_t(ns);
    A.nstreams = nstreams;
    return A;
}

}  // namespace og

extern "C" {

size_t opusgpu_silk_state_bytes(uint32_t nstreams)
{
    return sizeof(float) * (size_t(OG_SILK_SLOT) * 2 * size_t(nstreams) + size_t(nstreams) * 64);
}

int opusgpu_launch_silk(const OgSilkFrame *frames, uint32_t nframes, void *state, uint32_t nstreams, float *pcm,
                        size_t pcm_stride_floats, float *debug_internal, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!frames || !state || !pcm || nstreams == 0) return OPUSGPU_BAD_ARG;
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    const OgSilkTables *T;
    int rc = og::silk_tables_for_device(device, &T);
    if (rc) return rc;
    OgSilkArrays A = og::silk_arrays(state, nstreams);
    hipLaunchKernelGGL(og_silk_synth_kernel, dim3((2 * nframes + OG_SK_SLOTS - 1) / OG_SK_SLOTS), dim3(64), 0, (hipStream_t)stream, frames, nframes, A);
    hipLaunchKernelGGL(og_silk_output_kernel, dim3(nframes), dim3(OG_WG), 0, (hipStream_t)stream, frames, A, T, pcm,
                       pcm_stride_floats, debug_internal);
    OG_HIP(hipGetLastError());
    return OPUSGPU_OK;
}

}  // extern "C"
