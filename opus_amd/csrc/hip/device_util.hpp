#pragma once
#include <hip/hip_runtime.h>
#include "../../../include/opusgpu.h"
namespace og {
int device_fail(hipError_t e, const char *what);   // records opusgpu_last_error(), returns OPUSGPU_DEVICE
}
#define OG_HIP(call)                                                    \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return og::device_fail(e_, #call);        \
    } while (0)
