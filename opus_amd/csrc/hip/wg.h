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
#define OG_MEMBER __device__ __forceinline__
#define OG_LANES_BEGIN(tid) { const int tid = (int)threadIdx.x; {
#define OG_LANES_END } } __syncthreads();
#define OG_ATOMIC_OR(p, v) atomicOr((p), (v))
#define OG_ATOMIC_ADD(p, v) atomicAdd((p), (v))         /* returns the value before */
// the same for the kernels that run OG_WGT-lane workgroups (the CELT tail kernel)
#define OG_TLANES_BEGIN(tid) { const int tid = (int)threadIdx.x; {
#define OG_TLANES_END } } __syncthreads();
#define OG_TWAVES_BEGIN(w) { const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); {
#define OG_TWAVES_END } } __syncthreads();
#define OG_LANES_BEGIN_NOSYNC(tid) { const int tid = (int)threadIdx.x; {
#define OG_LANES_END_NOSYNC } }           /* regions that touch disjoint data; an explicit OG_BARRIER follows the batch */
// Wavefront-granular sections.  Short dependent chains (a handful of <= 176-element vector steps) cost a full
// workgroup barrier per step when all four waves work on one of them; a single wave needs none (LDS operations of
// one wave complete in issue order).  So independent chains go one per wave:
//     OG_WAVES_BEGIN(w)   uniform-per-wave code, w = wave index 0..OG_WG/64-1, with regions
//         OG_WAVE_BEGIN(lane)  per-lane code, lane = 0..63  OG_WAVE_END     <- compiler-level fence only
//     OG_WAVES_END                                                          <- workgroup barrier
// OG_WAVE0_BEGIN .. OG_WAVE0_END is the same with only wave 0 running (the others wait at the closing barrier).
// OG_U32 tells the compiler a value is wave-uniform (it moves to a scalar register, and so does what is derived from it).
#define OG_WAVES_BEGIN(w) { const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); {
#define OG_WAVES_END } } __syncthreads();
#define OG_WAVE0_BEGIN if (threadIdx.x < 64) {
#define OG_WAVE0_END } __syncthreads();
#define OG_WAVE_BEGIN(lane) { const int lane = (int)(threadIdx.x & 63); {
#define OG_WAVE_END } } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#define OG_U32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// A per-lane value that lives across several OG_WAVE regions of one OG_WAVES section (a register on the GPU): declare it
// with OG_WAVE_VAR outside the regions, use it as OG_WV(name) inside them (the region's lane variable must be `lane`).
#define OG_WAVE_VAR(T, name) T name
#define OG_WV(name) name
// The same for a value that lives across OG_LANES regions (the regions' lane variable must be `tid`).
#define OG_LANE_VAR(T, name) T name
#define OG_LV(name) name
// dst (a plain variable declared outside the OG_WAVE region) = lane 0's value of val, for every lane
#define OG_WAVE_BCAST0(dst, val) dst = __builtin_amdgcn_readfirstlane((int)(val))
// Register-only wave regions and cross-lane moves for recurrences that would otherwise pay an LDS round trip per step.  Used
// OUTSIDE OG_WAVE regions, on OG_WAVE_VAR floats:
//   OG_WAVE_SHR1(dst, src)        dst of lane l = src of lane l - 1 (lane 0: 0)         one DPP move (wave_shr:1)
//   OG_WAVE_PULL(dst, src, IDX)   dst of lane `lane` = src of lane (IDX) & 63            one ds_bpermute; IDX may use `lane`
// OG_WAVE_REGS_BEGIN / _END is an OG_WAVE region without the trailing fence: for code whose lanes exchange nothing through LDS
// inside the region and between consecutive such regions (end the sequence with an ordinary OG_WAVE region or OG_WAVE_FENCE).
OG_DEV float og_dpp_wave_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
#define OG_WAVE_SHR1(dst, src) dst = og_dpp_wave_shr1(src);
#define OG_WAVE_PULL(dst, src, IDX) { const int lane = (int)(threadIdx.x & 63); \
    dst = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((IDX) & 63) << 2, __builtin_bit_cast(int, src))); }
#define OG_WAVE_REGS_BEGIN(lane) { const int lane = (int)(threadIdx.x & 63); {
#define OG_WAVE_REGS_END } }
#define OG_WAVE_FENCE __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
// Lane exchanges for butterflies whose two inputs sit in lanes l and l ^ MASK (MASK = 1, 2, 4, 8: inside a row of 16 lanes):
//   OG_WAVE_XCH(dst, src, MEMBER, MASK)   dst of lane l = src MEMBER of lane l ^ MASK     (src: an OG_WAVE_VAR; MEMBER: e.g. .v[3] or empty)
// MASK 1, 2: DPP quad_perm; 8: DPP row_ror:8; 4: ds_swizzle (bit mode) -- no memory is touched by any of them.
// OG_WAVE_GET(src, L): lane L's value of an int OG_WAVE_VAR, as a uniform value.  OG_WAVE_REF(T, name): an OG_WAVE_VAR as a function parameter.
template <int MASK> OG_DEV float og_lane_xor(float v)
{
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "inside a row of 16 lanes");
    const int x = __builtin_bit_cast(int, v);
    int r;
    if (MASK == 1) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
    else if (MASK == 2) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    else if (MASK == 8) r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);      // row_ror:8
    else r = __builtin_amdgcn_ds_swizzle(x, 0x101F);                                       // and 0x1F, or 0, xor 4
    return __builtin_bit_cast(float, r);
}
#define OG_WAVE_XCH(dst, src, MEMBER, MASK) dst = og_lane_xor<MASK>(src MEMBER);
#define OG_WAVE_GET(src, L) __builtin_amdgcn_readlane((int)(src), (L))
// OG_LANE_GET0(var, L): lane L's value of an OG_LANE_VAR that every wavefront of the workgroup filled alike (by lane index
// inside the wave), as a uniform value; used between OG_LANES regions (no barrier needed: a wave reads its own copy)
#define OG_LANE_GET0(var, L) ((uint32_t)__builtin_amdgcn_readlane((int)(var), (int)(L)))
#define OG_WAVE_REF(T, name) T &name
// Reductions of a float OG_WAVE_VAR, used between OG_WAVE regions: OG_WAVE_SUM(dst, var): dst (a plain float) = the sum over the
// wavefront; OG_ROW16_SUM(dstvar, var): dstvar (an OG_WAVE_VAR) of a lane = the sum over the lane's row of 16.  Registers only
// (og_wave_sum / og_row16_sum, celt_frame.h); partners 1, 2, 4, ... apart in both builds.
#define OG_WAVE_SUM(dst, var) dst = og_wave_sum(var);
#define OG_ROW16_SUM(dstvar, var) dstvar = og_row16_sum(var);
// One-instruction approximations (1 ulp) for the CELT kernels, whose results are compared with a tolerance (the library
// forms expand to a dozen or more instructions each: range checks, Newton steps, denormal scaling).  Arguments are normal
// numbers wherever these are used.  The SILK kernels, which are held to bit-exactness, do not use them.
OG_DEV float og_frsq(float x) { return __builtin_amdgcn_rsqf(x); }
OG_DEV float og_frcp(float x) { return __builtin_amdgcn_rcpf(x); }
OG_DEV float og_fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
OG_DEV float og_fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Complex numbers as packed pairs (re in .x, im in .y: one 64-bit register pair): sums, differences and scalings are single
// v_pk_* instructions as the compiler emits them; the forms below need the operand selectors (which half of a source feeds
// which half of the result) and per-half negation that it does not derive -- two instructions per complex product
// instead of six, one for a sum with a factor of +-i instead of three.
typedef float OgC2 __attribute__((ext_vector_type(2)));
OG_DEV OgC2 og_c2_mul(OgC2 a, OgC2 w)                  // a w
{
    OgC2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                          // (a.i w.i, a.i w.r)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // (a.r w.r - t.lo, a.r w.i + t.hi)
    return r;
}
OG_DEV OgC2 og_c2_mulc(OgC2 a, OgC2 w)                 // a conj(w)
{
    OgC2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                          // (a.i w.i, a.r w.i)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // (a.r w.r + t.lo, a.i w.r - t.hi)
    return r;
}
OG_DEV OgC2 og_c2_sub_i(OgC2 a, OgC2 b)                // a - i b
{
    OgC2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
OG_DEV OgC2 og_c2_add_i(OgC2 a, OgC2 b)                // a + i b
{
    OgC2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
OG_DEV OgC2 og_c2_scale(OgC2 a, float s) { return a * s; }
OG_DEV OgC2 og_c2_fma(OgC2 a, float s, OgC2 b) { return __builtin_elementwise_fma(a, OgC2{s, s}, b); }     // a s + b
#else
#include <math.h>
#define OG_DEV static inline
#define OG_MEMBER inline
#define OG_LANES_BEGIN(tid) for (int tid = 0; tid < OG_WG; tid++) { {
#define OG_LANES_END } }
#define OG_ATOMIC_OR(p, v) (*(p) |= (v))
#define OG_ATOMIC_ADD(p, v) ((*(p) += (v)) - (v))
#define OG_TLANES_BEGIN(tid) for (int tid = 0; tid < OG_WGT; tid++) { {
#define OG_TLANES_END } }
#define OG_TWAVES_BEGIN(w) for (int w = 0; w < OG_WGT / 64; w++) { {
#define OG_TWAVES_END } }
#define OG_LANES_BEGIN_NOSYNC(tid) for (int tid = 0; tid < OG_WG; tid++) { {
#define OG_LANES_END_NOSYNC } }
#define OG_WAVES_BEGIN(w) for (int w = 0; w < OG_WG / 64; w++) { {
#define OG_WAVES_END } }
#define OG_WAVE0_BEGIN {
#define OG_WAVE0_END }
#define OG_U32(x) ((uint32_t)(x))
#define OG_WAVE_BEGIN(lane) for (int lane = 0; lane < 64; lane++) { {
#define OG_WAVE_END } }
#define OG_WAVE_VAR(T, name) T name[64]
#define OG_WV(name) name[lane]
#define OG_LANE_VAR(T, name) T name[OG_WG]
#define OG_LV(name) name[tid]
#define OG_WAVE_BCAST0(dst, val) { if (lane == 0) dst = (int)(val); }
#define OG_WAVE_SHR1(dst, src) { float t_[64]; for (int l_ = 0; l_ < 64; l_++) t_[l_] = l_ ? src[l_ - 1] : 0.f; \
    for (int l_ = 0; l_ < 64; l_++) dst[l_] = t_[l_]; }
#define OG_WAVE_PULL(dst, src, IDX) { float t_[64]; for (int lane = 0; lane < 64; lane++) t_[lane] = src[(IDX) & 63]; \
    for (int l_ = 0; l_ < 64; l_++) dst[l_] = t_[l_]; }
#define OG_WAVE_REGS_BEGIN(lane) for (int lane = 0; lane < 64; lane++) { {
#define OG_WAVE_REGS_END } }
#define OG_WAVE_FENCE
#define OG_WAVE_XCH(dst, src, MEMBER, MASK) { for (int l_ = 0; l_ < 64; l_++) dst[l_] = src[l_ ^ (MASK)] MEMBER; }
#define OG_WAVE_GET(src, L) ((int)src[L])
#define OG_LANE_GET0(var, L) ((uint32_t)var[L])
#define OG_WAVE_REF(T, name) T (&name)[64]
#define OG_LANE_TREE_(var, TOP) float v_[64], w_[64]; for (int i_ = 0; i_ < 64; i_++) v_[i_] = var[i_]; \
    for (int m_ = 1; m_ < (TOP); m_ <<= 1) { for (int i_ = 0; i_ < 64; i_++) w_[i_] = v_[i_] + v_[i_ ^ m_]; for (int i_ = 0; i_ < 64; i_++) v_[i_] = w_[i_]; }
#define OG_WAVE_SUM(dst, var) { OG_LANE_TREE_(var, 64) dst = v_[0]; }
#define OG_ROW16_SUM(dstvar, var) { OG_LANE_TREE_(var, 16) for (int i_ = 0; i_ < 64; i_++) dstvar[i_] = v_[i_]; }
OG_DEV float og_frsq(float x) { return 1.f / sqrtf(x); }
OG_DEV float og_frcp(float x) { return 1.f / x; }
OG_DEV float og_fsqrt(float x) { return sqrtf(x); }
OG_DEV float og_fexp2(float x) { return exp2f(x); }
struct OgC2 { float x, y; };
OG_DEV OgC2 operator+(OgC2 a, OgC2 b) { return OgC2{a.x + b.x, a.y + b.y}; }
OG_DEV OgC2 operator-(OgC2 a, OgC2 b) { return OgC2{a.x - b.x, a.y - b.y}; }
OG_DEV OgC2 operator*(OgC2 a, OgC2 b) { return OgC2{a.x * b.x, a.y * b.y}; }      // per component, as on the packed pair
OG_DEV OgC2 og_c2_mul(OgC2 a, OgC2 w) { return OgC2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
OG_DEV OgC2 og_c2_mulc(OgC2 a, OgC2 w) { return OgC2{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }
OG_DEV OgC2 og_c2_sub_i(OgC2 a, OgC2 b) { return OgC2{a.x + b.y, a.y - b.x}; }
OG_DEV OgC2 og_c2_add_i(OgC2 a, OgC2 b) { return OgC2{a.x - b.y, a.y + b.x}; }
OG_DEV OgC2 og_c2_scale(OgC2 a, float s) { return OgC2{a.x * s, a.y * s}; }
OG_DEV OgC2 og_c2_fma(OgC2 a, float s, OgC2 b) { return OgC2{a.x * s + b.x, a.y * s + b.y}; }
#endif
// floor(a / d) for small non-negative integers (a < 4096, 0 < d <= 1024) through the float reciprocal: (a + 1/2) / d is at
// least 1 / (2d) away from an integer, far more than the product's rounding error
OG_DEV int og_idiv_small(int a, int d) { return (int)(((float)a + 0.5f) * og_frcp((float)d)); }
