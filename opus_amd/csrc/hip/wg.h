// Workgroup programming frame shared by the opus_amd kernels.
//
// Kernel bodies are written as a sequence of "lane regions" separated by workgroup barriers:
//
//     uniform code (identical in every lane; reads LDS/global, never writes LDS)
//     OG_LANES_BEGIN(tid)  per-lane code, tid = 0 .. OG_WG-1  OG_LANES_END      <- barrier here
//
// hipcc sees plain HIP: the region is the body every lane executes, OG_LANES_END is __syncthreads().
// The same text also builds with g++ for tests/emu (CPU CI of the kernel logic on a box without a
// GPU): there a region is a loop over tid.  The emulation is a test fixture only; libopusgpu.so never
// contains it and nothing under opus_amd/ can reach it.
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define OG_DEV __device__ __forceinline__
#define OG_LANES_BEGIN(tid) { const int tid = (int)threadIdx.x; {
#define OG_LANES_END } } __syncthreads();
#define OG_ATOMIC_OR(p, v) atomicOr((p), (v))
#define OG_LANES_BEGIN_NOSYNC(tid) { const int tid = (int)threadIdx.x; {
#define OG_LANES_END_NOSYNC } }           /* regions that touch disjoint data; an explicit OG_BARRIER follows the batch */
#else
#include <math.h>
#define OG_DEV static inline
#define OG_LANES_BEGIN(tid) for (int tid = 0; tid < OG_WG; tid++) { {
#define OG_LANES_END } }
#define OG_ATOMIC_OR(p, v) (*(p) |= (v))
#define OG_LANES_BEGIN_NOSYNC(tid) for (int tid = 0; tid < OG_WG; tid++) { {
#define OG_LANES_END_NOSYNC } }
#endif
