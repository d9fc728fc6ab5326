// SILK kernels: placeholder until K4/K5/K6 land.
#include <hip/hip_runtime.h>
#include "device_util.hpp"
extern "C" size_t opusgpu_sizeof_silk_state(void) { return 16; }
extern "C" int opusgpu_launch_silk(const OgSilkFrame *, uint32_t nframes, void *, float *, size_t, int, void *)
{
    return nframes ? OPUSGPU_UNSUPPORTED : OPUSGPU_OK;
}
