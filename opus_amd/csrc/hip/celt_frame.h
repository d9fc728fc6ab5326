// CELT frame decode, device side, after K1 (celt_pvq.h: PVQ index -> pulse vector, normalise, spreading rotation,
// collapse mask) and the band plan (celt_plan.h): og_celt_spectrum -- one workgroup per frame, stateless: folding / noise
// fill, Haar + Hadamard reordering, stereo merge, anti-collapse, denormalise and the IMDCT up to its
// post-twiddle, all in LDS -- and og_celt_tail -- one workgroup per stream in time order: TDAC window + overlap-add, comb
// post-filter, de-emphasis, PCM (K3).  The two meet in the frame's scratch row in HBM (DESIGN.md section 4 has why).
//
// Reference anchors (lu-zero/opus): cwrsi celt/decoder.rs:541-660, unquantize :766-786, exp_rotation
// :676-721, extract_collapse_mask :723-736, decode_band_no_split fill paths :1633-1661, haar1 :453-466,
// (de)interleave_hadamard :472-538, stereo_merge :796-827, decode_band post-processing :1994-2032,
// decode_bands :2037-2220.  Everything after band decoding is absent from the reference and follows
// RFC 6716 4.3.5-4.3.7 (the reference only keeps the state fields, celt/decoder.rs:22-43, :90-94).
//
// HBM-bound by design (no MFMA: nothing here is a dense contraction); see DESIGN.md for the roofline.
#pragma once
#include "wg.h"
#include "../../../include/opusgpu.h"

#include "celt_frame_types.h"
#include "celt_plan.h"

struct alignas(16) OgF4 { float x, y, z, w; };

#if defined(__HIP__)
#define OG_MUL24(a, b) __mul24((a), (b))                /* index products below 2^24: full-rate multiply */
#else
#define OG_MUL24(a, b) ((a) * (b))
#endif


#ifndef OG_LDS_OPS
#define OG_LDS_OPS 16                 // fill ops staged in LDS per frame (C2 frames carry ~5); the rest stay in HBM
#endif

// 20.4 KB: eight workgroups per CU (LDS is allotted in 1280-byte granules: 16 of them).  Buffers with disjoint
// lifetimes share storage: `R` holds the folding source `norm` during the band walk and the FFT pong buffer afterwards
// (sums over a wavefront stay in registers: OG_WAVE_SUM); the band-walk tables give way to the FFT twiddles.
struct alignas(16) OgCeltLds {
    float X[2][960];                  // coefficients (bands) -> MDCT bins -> FFT ping buffer -> samples before the window
    float R[1920];                    // norm[2][800]   ->   bin -> band map (denormalisation)   ->   FFT pong buffer [2][480] complex
    union {
        struct {                      // band-walk lifetime
            alignas(16) uint32_t plan_head[OG_PLAN_HEAD_BYTES / 4];   // OgBandPlan up to (not including) .op
            OgFillOp ops[OG_LDS_OPS];
            struct {
                float lb[OG_WG / 64][176];                        // per wave: transformed copy of its task's folding source
            } s;
        } b;
        struct {                      // from denormalisation on
            OgCf W[480];              // FFT twiddles
        } t;
    } w;
    struct { float emeans[24]; uint8_t ebands[24], width[24]; } k;   // LDS copy of the small constant tables it indexes
    float gband[2][24];               // denormalisation gains
    OgCeltHdr hdr;                    // the frame's side-info, staged once (uniform control code reads it from LDS)
};
// views into R (band-walk lifetime)
OG_DEV float *og_norm(OgCeltLds &L, int row) { return L.R + 800 * row; }                // folding source row (celt/decoder.rs `norm_mid` / `norm_side`)

#include "celt_pvq.h"

// ------------------------------------------------------------------------------------------------
// Reductions evaluated in uniform code: every lane gets the same value.  On the GPU each wavefront reduces the
// (<= 176-element) vector on its own (lane-strided partials + xor butterfly), so no LDS scratch and no barrier.
#if defined(__HIP__)
#define OG_BARRIER __syncthreads();
// Sum over the wavefront, the same in every lane.  Six DPP additions (each one instruction: the lane exchange is an operand
// modifier) and a lane read, where a __shfl_xor butterfly is six LDS-pipe round trips with their address arithmetic.  Inside a
// row of 16 lanes: partners 1 and 2 away (quad_perm), then the mirrored half and the mirrored row -- which hold the same
// values as the partners 4 and 8 away, since the lanes of a quad / half already agree -- then the row sums across the rows
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) and lane 63's total to everyone: the pairing of a
// butterfly over partners 1, 2, 4, ... 32.
template <int CTRL, int ROWS> OG_DEV float og_dpp_get(float v)       // the lane's DPP partner's v (0 in rows outside ROWS)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
OG_DEV float og_row16_sum(float v)                      // sum over the lane's row of 16, in every lane of the row
{
    v += og_dpp_get<0xB1, 0xF>(v);                      // quad_perm [1,0,3,2]
    v += og_dpp_get<0x4E, 0xF>(v);                      // quad_perm [2,3,0,1]
    v += og_dpp_get<0x141, 0xF>(v);                     // row_half_mirror
    v += og_dpp_get<0x140, 0xF>(v);                     // row_mirror
    return v;
}
OG_DEV float og_wave_sum(float v)
{
    v = og_row16_sum(v);
    v += og_dpp_get<0x142, 0xA>(v);                     // row_bcast:15 -> rows 1, 3
    v += og_dpp_get<0x143, 0xC>(v);                     // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
OG_DEV float og_uniform_sumsq(const float *x, int n)
{
    float e = 0.f;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) e += x[i] * x[i];
    return og_wave_sum(e);
}
OG_DEV void og_uniform_dot2(const float *x, const float *y, int n, float *xy, float *yy)
{
    float a = 0.f, b = 0.f;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) { a += x[i] * y[i]; b += y[i] * y[i]; }
    *xy = og_wave_sum(a);
    *yy = og_wave_sum(b);
}
#else
#define OG_BARRIER
OG_DEV float og_uniform_sumsq(const float *x, int n)
{
    float e = 0.f;
    for (int i = 0; i < n; i++) e += x[i] * x[i];
    return e;
}
OG_DEV void og_uniform_dot2(const float *x, const float *y, int n, float *xy, float *yy)
{
    float a = 0.f, b = 0.f;
    for (int i = 0; i < n; i++) { a += x[i] * y[i]; b += y[i] * y[i]; }
    *xy = a;
    *yy = b;
}
#endif

// ------------------------------------------------------------------------------------------------
// K2 building blocks: workgroup-collective vector operations on LDS (each ends in a barrier).
// x *= gain / ||x||  (celt/decoder.rs:788-794)
OG_DEV void og_wg_renormalize(float *x, int n, float gain)
{
    const float g = gain * og_frsq(1e-15f + og_uniform_sumsq(x, n));
    OG_BARRIER                                          // every wave has read x before anyone scales it
    OG_LANES_BEGIN(tid)
    for (int p = tid; p < n; p += OG_WG) x[p] *= g;
    OG_LANES_END
}

#if defined(__HIP__)
#define OG_STAMP_TO(buf, k) { if ((buf) && threadIdx.x == 0) (buf)[k] = (uint64_t)__builtin_readcyclecounter(); }
#else
#define OG_STAMP_TO(buf, k) {}
#endif
#define OG_STAMP(k) OG_STAMP_TO(dbg.cycles, k)
#if defined(__HIP__)
#define OG_CYC() ((uint64_t)__builtin_readcyclecounter())
#else
#define OG_CYC() ((uint64_t)0)
#endif

struct OgBandCtx {
    const OgCeltHdr *h;               // staged in LDS
    const OgBandPlan *plan;           // the frame's plan in HBM (only fill ops beyond OG_LDS_OPS are read from there)
    const OgCeltTables *T;
    OgCeltLds *L;
    uint32_t seed;                    // uniform across the workgroup
    int lm;
    uint64_t *cycles;                 // profiling stamps (may be null)
    int stop;                         // profiling aid: the frame ends after stage `stop` (0 = run everything)
};

// celt/decoder.rs:468-470 HADAMARD_ORDERY for strides 2, 4, 8, 16, one nibble per entry (entry i at bits 4i)
OG_DEV int og_ordery(int slog, int i)
{
    const uint64_t t = slog == 1 ? 0x01ull : slog == 2 ? 0x1203ull : slog == 3 ? 0x25163407ull : 0x5A2D691E4B3C780Full;
    return (int)(t >> (4 * i)) & 15;
}

// where interleave_hadamard takes the value at position d from (= where deinterleave_hadamard puts the value of d)
OG_DEV int og_reorder(int slog, int n0p, int had, int d)
{
    if (!slog) return d;
    const int S = 1 << slog, i = d & (S - 1), j = d >> slog;
    return (had ? og_ordery(slog, i) : i) * n0p + j;
}

#if defined(__HIP__)
template <class T> OG_DEV T og_uniform(const T &v)     // load a wave-uniform record into scalar registers
{
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = OG_U32(w[k]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}
#else
template <class T> OG_DEV T og_uniform(const T &v) { return v; }
#endif

// The transform of one quant_band call: the Haar steps of tf recombination / time-division (celt/decoder.rs:453-466,
// :1708-1740, :1994-2019) are butterflies on the index bits `hm` (celt_plan.h; bits 0..3 only: tf changes are at most three
// steps, celt/decoder.rs:221-226, on top of at most eight blocks) and commute; the Hadamard (de)interleave (:472-538) is a
// permutation.  A lane group of STEP lanes (a whole wave, STEP = 64, or a quarter of one, STEP = 16) takes a task: the lane
// with index `first` in the group holds elements first, first + STEP, ... (K of them at most) in registers, so
//   * the permutation is the address the element is read from (POST: interleave on the way in) or written to (PRE:
//     deinterleave on the way out).  og_reorder(p) for p = first + STEP k is affine in k -- STEP is a multiple of the
//     permutation's stride 2^slog <= 16 -- so it costs one add per element (OgXfAddr);
//   * the butterflies pair lanes l and l ^ 2^bit of a row of 16: DPP / ds_swizzle exchanges (wg.h OG_WAVE_XCH), no LDS traffic;
//   * the 2^(-m/2) scale and lowband_out = sqrt(N0) x (celt/decoder.rs:2021-2028) are applied on the way out.
// Same operations in the same order per element as the step-by-step form (butterflies from the lowest bit up).
template <int K> struct OgXr { float v[K]; };

OG_DEV float og_xf_scale(const OgTask tk)               // 2^(-m/2) for the m butterflies
{
    const int m = (tk.rctd & 15) + (tk.rctd >> 4);
    if (!tk.hm) return 1.f;
    const uint32_t pw = (127u - (uint32_t)(m >> 1)) << 23;     // 2^-(m >> 1), exactly what the division by (1 << (m >> 1)) gives
    float f;
    __builtin_memcpy(&f, &pw, 4);
    return (m & 1 ? 0.70710678118654752440f : 1.f) * f;
}

struct OgXfAddr { int base, step; };  // og_reorder(first + STEP k) = base + step k
OG_DEV OgXfAddr og_xf_addr(const OgTask tk, int first, int STEP)
{
    const int slog = tk.slog;
    OgXfAddr a;
    if (!slog) { a.base = first; a.step = STEP; return a; }
    const int i = first & ((1 << slog) - 1);
    a.base = OG_MUL24((tk.flags & OG_T_LONG) ? og_ordery(slog, i) : i, (int)tk.n0p) + (first >> slog);
    a.step = STEP >> slog;
    return a;
}

// the butterflies of the lanes' tasks on registers: hmv = the lane's task's hm (0 for lanes without one), hm_any = the OR of
// hmv over the wave (uniform), nmax = a uniform bound on the tasks' n.  Lanes past their task's n hold zeros: their partners
// are past it too (n is a multiple of every butterfly's span).
// A butterfly as one multiply and one fused multiply-add per element: v' = ca v + cb v_partner with, per lane and index bit,
// (ca, cb) = (1, 1) in the lower lane of a pair, (-1, 1) in the upper one, (1, 0) where the lane's task has no butterfly on
// that bit -- products with 0 and +-1 are exact, so this is the sum / difference itself, without a select per element.
#define OG_XF_BFLY_BIT(B)                                                                                           \
    if (hm_any & (1 << (B))) {                                                                                      \
        OG_WAVE_REGS_BEGIN(lane)                                                                                    \
        const bool on_ = ((OG_WV(hmv) >> (B)) & 1) != 0;                                                            \
        OG_WV(ca) = (on_ && ((lane >> (B)) & 1)) ? -1.f : 1.f;                                                      \
        OG_WV(cb) = on_ ? 1.f : 0.f;                                                                                \
        OG_WAVE_REGS_END                                                                                            \
        _Pragma("unroll") for (int k = 0; k < K; k++) {                                                             \
            if (k * STEP < nmax) {                                                                                  \
                OG_WAVE_XCH(ex, rv, .v[k], 1 << (B))                                                                \
                OG_WAVE_REGS_BEGIN(lane)                                                                            \
                (void)lane;                                                                                         \
                OG_WV(rv).v[k] = fmaf(OG_WV(rv).v[k], OG_WV(ca), OG_WV(cb) * OG_WV(ex));                            \
                OG_WAVE_REGS_END                                                                                    \
            }                                                                                                       \
        }                                                                                                           \
    }
template <int K, int STEP>
OG_DEV void og_wave_bflies(OG_WAVE_REF(OgXr<K>, rv), OG_WAVE_REF(int, hmv), int hm_any, int nmax)
{
    OG_WAVE_VAR(float, ex);
    OG_WAVE_VAR(float, ca);
    OG_WAVE_VAR(float, cb);
    OG_XF_BFLY_BIT(0)
    OG_XF_BFLY_BIT(1)
    OG_XF_BFLY_BIT(2)
    OG_XF_BFLY_BIT(3)
}

// One task by one wave (inside an OG_WAVES section; tk uniform).  PRE: the folding source -> butterflies -> scale ->
// deinterleave into the wave's lowband buffer lb.  POST: the decoded band -> interleave -> butterflies -> scale, in place, and
// lowband_out.
template <bool PRE>
OG_DEV void og_task_xform(OgCeltLds &L, const OgTask tk, int slot, float *lb, int norm_offset)
{
    const int n = tk.n;
    float *x = &L.X[0][0] + slot * 960 + tk.boff;
    OG_WAVE_VAR(OgXr<3>, rv);
    OG_WAVE_VAR(int, hmv);
    OG_WAVE_BEGIN(lane)
    const OgXfAddr a = og_xf_addr(tk, lane, 64);
    const float *src = PRE ? og_norm(L, (tk.flags & OG_T_LB_ROW1) ? 1 : 0) + tk.lowpos : x;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p = lane + 64 * k;
        OG_WV(rv).v[k] = p < n ? src[PRE ? p : a.base + a.step * k] : 0.f;
    }
    OG_WV(hmv) = tk.hm;
    OG_WAVE_END                                         // every element is in a register before any is written
    if (tk.hm) og_wave_bflies<3, 64>(rv, hmv, tk.hm, n);
    const float scale = og_xf_scale(tk);
    if (PRE) {
        OG_WAVE_BEGIN(lane)
        const OgXfAddr a = og_xf_addr(tk, lane, 64);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (lane + 64 * k < n) lb[a.base + a.step * k] = OG_WV(rv).v[k] * scale;
        }
        OG_WAVE_END
    } else {
        const bool xf = tk.hm != 0, wr = xf || tk.slog;
        float *nr = tk.norm_row != 0xFF ? og_norm(L, tk.norm_row) + (tk.boff - norm_offset) : nullptr;
        const float sq = og_fsqrt((float)n);
        OG_WAVE_BEGIN(lane)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int p = lane + 64 * k;
            if (p < n) {
                float v = OG_WV(rv).v[k];
                if (xf) v = v * scale;
                if (wr) x[p] = v;
                if (nr) nr[p] = sq * v;
            }
        }
        OG_WAVE_END
    }
}

// The noise generator (celt/decoder.rs:1595-1599, seed' = 1664525 seed + 1013904223) advanced j < 4096 steps at once: the
// 2^k-step maps seed -> a_k seed + c_k (compile-time constants) applied for the set bits of j.  Integer arithmetic mod
// 2^32: the same value as j single steps.
struct OgLcgPow { uint32_t a[12], c[12]; };
constexpr OgLcgPow og_lcg_pows()
{
    OgLcgPow p = {};
    p.a[0] = 1664525u; p.c[0] = 1013904223u;
    for (int k = 0; k + 1 < 12; k++) { p.a[k + 1] = p.a[k] * p.a[k]; p.c[k + 1] = p.c[k] * (p.a[k] + 1u); }
    return p;
}
template <int BITS = 8>               // j < 2^BITS
OG_DEV uint32_t og_lcg_jump(uint32_t j, uint32_t s)
{
    constexpr OgLcgPow P = og_lcg_pows();
#pragma unroll
    for (int k = 0; k < BITS; k++)
        if ((j >> k) & 1u) s = P.a[k] * s + P.c[k];
    return s;
}

// One live fill leaf (celt/decoder.rs:1633-1661 + renormalize_vector :788-794), by wave w alone: n <= 176, at most
// three coefficients per lane, no workgroup barrier.  Ops of one level touch disjoint coefficients.
OG_DEV void og_fill_op(OgBandCtx &c, int k, int w, int task_pos)     // task_pos: slot * 960 + first bin of the op's task
{
    OgCeltLds &L = *c.L;
    const OgFillOp op = k < OG_LDS_OPS ? og_uniform(L.w.b.ops[k]) : og_uniform(c.plan->op[k]);
    float *xl = &L.X[0][0] + op.pos;
    OG_WAVE_VAR(float, ev);                             // the lane's share of the energy
    OG_WAVE_VAR(OgXr<3>, vv);                           // its coefficients (n <= 176: at most three), kept until they are scaled
    const float *lbl = L.w.b.s.lb[w] + ((int)op.pos - task_pos);
    const int nn = op.n;
    const uint32_t s0 = op.seed;
    const bool fold = (op.flags & OG_OP_FOLD) != 0;     // fold the lower band in (celt/decoder.rs:1652-1659), else noise (:1644-1651)
    OG_WAVE_BEGIN(lane)
    float e = 0.f;
    uint32_t s = og_lcg_jump<7>((uint32_t)lane + 1u, s0);       // lane + 1 <= 64
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int p = lane + 64 * j;
        float v = 0.f;
        if (64 * j < nn && p < nn) v = fold ? lbl[p] + ((s & 0x8000) ? 1.f / 256 : -1.f / 256) : (float)((int32_t)s >> 20);
        OG_WV(vv).v[j] = v;
        e += v * v;
        s = og_lcg_jump(64u, s);
    }
    OG_WV(ev) = e;
    OG_WAVE_END
    float esum;
    OG_WAVE_SUM(esum, ev)
    const float g = op.gain * og_frsq(1e-15f + esum);
    OG_WAVE_BEGIN(lane)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int p = lane + 64 * j;
        if (64 * j < nn && p < nn) xl[p] = OG_WV(vv).v[j] * g;
    }
    OG_WAVE_END
}

// celt/decoder.rs:2037-2220 with the range-decoder half removed (the host ran it) and the integer recurrences
// (fill masks, noise seeds, collapse masks, dependency levels) already resolved by the plan kernel (celt_plan.h).
// What is left is sample work on independent quant_band calls ("tasks"), dealt out to the waves level by level:
//   level 0   every call without fill leaves;
//   level l   calls with fill leaves whose folding sources are complete.
// A wave takes a task from start to end -- transformed lowband copy, fill leaves, reorder + Haar, lowband_out -- with
// no workgroup barrier (the tasks of a level touch disjoint coefficients); the workgroup meets once per level.
//   merges    the mid/side -> left/right rotations, one joint band per wave at a time (norm keeps the mid, so nothing
//             downstream of a band needs its merged form).
OG_DEV void og_celt_bands(OgBandCtx &c)
{
    const OgCeltHdr *h = c.h;
    OgCeltLds &L = *c.L;
    const OgBandPlan *PL = (const OgBandPlan *)L.w.b.plan_head;      // head fields only
    const OgTask *tasks = (const OgTask *)L.w.b.plan_head;
    const int lm = c.lm, M = 1 << lm, start = h->start, end = h->end;
    const int C = (h->flags & OG_F_STEREO) ? 2 : 1;
    const int norm_offset = M * L.k.ebands[start];

    OG_STAMP_TO(c.cycles, 16)
    const int nlevels = (int)OG_U32(PL->nlevels), merge_level = (int)OG_U32(PL->merge_level);
    for (int lvl = 0; lvl <= nlevels; lvl++) {
        if (lvl == merge_level) {                       // dual -> intensity switch (celt/decoder.rs:2133-2138)
            const int cnt = PL->merge_cnt;
            OG_LANES_BEGIN(tid)
            for (int p = tid; p < cnt; p += OG_WG) L.R[p] = 0.5f * (L.R[p] + L.R[800 + p]);
            OG_LANES_END
        }
        if (lvl == 0) {
            // every call without fill leaves (most of the frame): a call per 16-lane group, largest first, so that the 42
            // calls take three rounds per wave instead of ten
            OG_WAVES_BEGIN(w)
            const int t0 = (int)OG_U32(PL->task_start[0]), t1 = (int)OG_U32(PL->task_start[1]);
            for (int base = 0; base < t1 - t0; base += 4 * (OG_WG / 64)) {
                OG_WAVE_VAR(OgTask, tkv);                           // the group's task: fetched once per round
                OG_WAVE_VAR(int, xoffv);                            // where its band starts in X
                OG_WAVE_VAR(int, hmv);
                OG_WAVE_VAR(OgXr<11>, rv);                          // its elements l, l + 16, ... (l = lane & 15)
                int nmax = 0;                                       // widest task of the wave's four: group 0's (largest first)
                OG_WAVE_BEGIN(lane)
                const int k = base + 4 * w + (lane >> 4), l = lane & 15;
                OgTask tk;
                tk.boff = tk.n = tk.n0p = tk.lowpos = tk.ops = 0; tk.flags = tk.hm = tk.slog = tk.rctd = tk.level = 0; tk.norm_row = 0xFF;
                int slot = 0;
                if (k < t1 - t0) {
                    const int id = PL->order[t1 - 1 - k];
                    tk = tasks[id];
                    slot = id & 1;
                }
                OG_WAVE_BCAST0(nmax, tk.n);
                const int xoff = slot * 960 + tk.boff, n = tk.n;
                const float *x = &L.X[0][0] + xoff;
                const OgXfAddr a = og_xf_addr(tk, l, 16);
#pragma unroll
                for (int j = 0; j < 11; j++) {
                    if (j * 16 < nmax) OG_WV(rv).v[j] = l + 16 * j < n ? x[a.base + a.step * j] : 0.f;
                }
                OG_WV(tkv) = tk;
                OG_WV(xoffv) = xoff;
                OG_WV(hmv) = tk.hm;
                OG_WAVE_END                                         // every element is in a register before any is written
                const int hm_any = OG_WAVE_GET(hmv, 0) | OG_WAVE_GET(hmv, 16) | OG_WAVE_GET(hmv, 32) | OG_WAVE_GET(hmv, 48);
                if (hm_any) og_wave_bflies<11, 16>(rv, hmv, hm_any, nmax);
                OG_WAVE_BEGIN(lane)
                const OgTask tk = OG_WV(tkv);
                const int l = lane & 15, n = tk.n;
                const bool xf = tk.hm != 0, wr = xf || tk.slog;
                float *x = &L.X[0][0] + OG_WV(xoffv);
                float *nr = tk.norm_row != 0xFF ? og_norm(L, tk.norm_row) + (tk.boff - norm_offset) : nullptr;
                const float sq = og_fsqrt((float)n), scale = og_xf_scale(tk);
#pragma unroll
                for (int j = 0; j < 11; j++) {
                    const int p = l + 16 * j;
                    if (j * 16 < nmax && p < n) {                   // n = 0 for lanes without a task
                        float v = OG_WV(rv).v[j];
                        if (xf) v = v * scale;
                        if (wr) x[p] = v;
                        if (nr) nr[p] = sq * v;
                    }
                }
                OG_WAVE_END
            }
            OG_WAVES_END
            OG_STAMP_TO(c.cycles, 17)
            if (c.stop == 2) return;
            continue;
        }
        OG_WAVES_BEGIN(w)
        const int t1 = (int)OG_U32(PL->task_start[lvl + 1]);
        for (int ti = (int)OG_U32(PL->task_start[lvl]) + w; ti < t1; ti += OG_WG / 64) {
            const int id = (int)OG_U32(PL->order[ti]), slot = id & 1;
            const OgTask tk = og_uniform(tasks[id]);
            if (tk.flags & OG_T_NEED_LB) og_task_xform<true>(L, tk, slot, L.w.b.s.lb[w], norm_offset);
            const int k0 = tk.ops & 1023, k1 = k0 + (tk.ops >> 10);
            for (int k = k0; k < k1; k++) og_fill_op(c, k, w, slot * 960 + tk.boff);
            if (tk.hm | tk.slog | (tk.norm_row != 0xFF)) og_task_xform<false>(L, tk, slot, nullptr, norm_offset);
        }
        OG_WAVES_END
        if (lvl == 0) OG_STAMP_TO(c.cycles, 17)
    }

    OG_STAMP_TO(c.cycles, 18)
    if (c.stop == 3) return;
    // ---- merges: stereo_merge (celt/decoder.rs:796-827) + the N == 2 case (:1826-1862).  A band of 64 bins or more takes a
    //      whole wave, the others a 16-lane group each, four to a wave (bands come in increasing width, so a wave's four are
    //      alike); both kinds are dealt round-robin to the waves.  A lane holds its (at most three) bin pairs in registers from
    //      the dot products to the rotation; the sums reduce inside the wave / group, no workgroup barrier.
    if (C == 2) {
        uint32_t jm = OG_U32(PL->joint);
        jm &= end > start ? (((end >= 21 ? (1u << 21) : (1u << end)) - 1u) & ~((1u << start) - 1u)) : 0u;
        const uint32_t widem = lm == 3 ? 0x1E0000u : lm == 2 ? 0x180000u : 0u;      // M * width >= 64 (celt/decoder.rs:281-283)
        OG_WAVES_BEGIN(w)
        OG_WAVE_VAR(float, av); OG_WAVE_VAR(float, bv);          // a lane's shares of <x, y> and <y, y>
        int turn = 0;                                   // whose turn the next piece of work is
        for (uint32_t m = jm & widem; m; m &= m - 1u, turn++) {
            if ((turn & (OG_WG / 64 - 1)) != w) continue;
            const int i = __builtin_ctz(m), n = M * L.k.width[i], off = M * L.k.ebands[i];
            const OgCeltBand bd = og_uniform(h->band[i]);
            float *x = L.X[0] + off, *y = L.X[1] + off;
            OG_WAVE_VAR(OgXr<3>, xv); OG_WAVE_VAR(OgXr<3>, yv);
            OG_WAVE_BEGIN(lane)
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (64 * k < n) {                       // (uniform)
                    const int p = lane + 64 * k;
                    const float xe = p < n ? x[p] : 0.f, ye = p < n ? y[p] : 0.f;
                    OG_WV(xv).v[k] = xe; OG_WV(yv).v[k] = ye;
                    a += xe * ye; b += ye * ye;
                }
            }
            OG_WV(av) = a; OG_WV(bv) = b;
            OG_WAVE_END
            float xy, yy;
            OG_WAVE_SUM(xy, av)
            OG_WAVE_SUM(yy, bv)
            const float mid = (1.f / 32768) * bd.imid, inv = (bd.flags & OG_B_INV) ? -1.f : 1.f;
            const float xp = mid * xy, e = mid * mid + yy, e0 = e - 2.f * xp, e1 = e + 2.f * xp;
            float ax = 1.f, bx = 0.f, ay = inv, by = 0.f;              // degenerate: y = x
            if (!(e0 < 6e-4f || e1 < 6e-4f)) {
                const float g0 = og_frsq(e0), g1 = og_frsq(e1);
                ax = g0 * mid; bx = -g0; ay = inv * g1 * mid; by = inv * g1;
            }
            OG_WAVE_BEGIN(lane)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int p = lane + 64 * k;
                if (64 * k < n && p < n) {
                    const float xe = OG_WV(xv).v[k], ye = OG_WV(yv).v[k];
                    x[p] = ax * xe + bx * ye;
                    y[p] = ay * xe + by * ye;
                }
            }
            OG_WAVE_END
        }
        for (uint32_t m = jm & ~widem; m; turn++) {
            int ib[4], nmaxr = 0;                       // the round's four bands (-1: none); the widest one's bins
            for (int g = 0; g < 4; g++) {
                ib[g] = m ? __builtin_ctz(m) : -1;
                if (m) nmaxr = M * L.k.width[ib[g]];
                m &= m - 1u;
            }
            if ((turn & (OG_WG / 64 - 1)) != w) continue;
            OG_WAVE_VAR(OgCeltBand, bdv);                            // the group's band: fetched once
            OG_WAVE_VAR(int, nv);                                    // its width, 0 = nothing to merge
            OG_WAVE_VAR(int, offv);
            OG_WAVE_VAR(OgXr<3>, xv); OG_WAVE_VAR(OgXr<3>, yv);
            OG_WAVE_BEGIN(lane)
            const int g = lane >> 4, l = lane & 15;
            const int i = g == 0 ? ib[0] : g == 1 ? ib[1] : g == 2 ? ib[2] : ib[3];
            float a = 0.f, b = 0.f;
            const OgCeltBand bd = h->band[i >= 0 ? i : start];
            const int n = i >= 0 ? M * L.k.width[i] : 0, off = i >= 0 ? M * L.k.ebands[i] : 0;
            const float *x = L.X[0] + off, *y = L.X[1] + off;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (16 * k < nmaxr) {                   // (uniform)
                    const int p = l + 16 * k;
                    const float xe = p < n ? x[p] : 0.f, ye = p < n ? y[p] : 0.f;
                    OG_WV(xv).v[k] = xe; OG_WV(yv).v[k] = ye;
                    if (n > 2) { a += xe * ye; b += ye * ye; }
                }
            }
            OG_WV(bdv) = bd; OG_WV(nv) = n; OG_WV(offv) = off;
            OG_WV(av) = a; OG_WV(bv) = b;
            OG_WAVE_END
            OG_WAVE_VAR(float, xyv); OG_WAVE_VAR(float, yyv);
            OG_ROW16_SUM(xyv, av)
            OG_ROW16_SUM(yyv, bv)
            OG_WAVE_BEGIN(lane)
            const int l = lane & 15, n = OG_WV(nv);
            const float xy = OG_WV(xyv), yy = OG_WV(yyv);
            if (n) {
                const OgCeltBand bd = OG_WV(bdv);
                float *x = L.X[0] + OG_WV(offv), *y = L.X[1] + OG_WV(offv);
                const float mid = (1.f / 32768) * bd.imid, side = (1.f / 32768) * bd.iside;
                const float inv = (bd.flags & OG_B_INV) ? -1.f : 1.f;
                if (n == 2) {
                    if (l == 0) {
                        const int sw = (bd.flags & OG_B_N2SWAP) != 0;
                        float *x2 = sw ? y : x, *y2 = sw ? x : y;
                        const float sign = (bd.flags & OG_B_N2NEG) ? -1.f : 1.f;
                        y2[0] = -sign * x2[1];
                        y2[1] = sign * x2[0];
                        x[0] *= mid; x[1] *= mid; y[0] *= side; y[1] *= side;
                        float t = x[0]; x[0] = t - y[0]; y[0] = inv * (t + y[0]);
                        t = x[1]; x[1] = t - y[1]; y[1] = inv * (t + y[1]);
                    }
                } else {
                    const float xp = mid * xy, sd = yy;
                    const float e = mid * mid + sd, e0 = e - 2.f * xp, e1 = e + 2.f * xp;
                    float ax = 1.f, bx = 0.f, ay = inv, by = 0.f;      // degenerate: y = x
                    if (!(e0 < 6e-4f || e1 < 6e-4f)) {
                        const float g0 = og_frsq(e0), g1 = og_frsq(e1);
                        ax = g0 * mid; bx = -g0; ay = inv * g1 * mid; by = inv * g1;
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const int p = l + 16 * k;
                        if (16 * k < nmaxr && p < n) {
                            const float xe = OG_WV(xv).v[k], ye = OG_WV(yv).v[k];
                            x[p] = ax * xe + bx * ye;
                            y[p] = ay * xe + by * ye;
                        }
                    }
                }
            }
            OG_WAVE_END
        }
        OG_WAVES_END
    }
    c.seed = PL->seed_out;
    OG_STAMP_TO(c.cycles, 19)
}

// RFC 6716 4.3.5 anti-collapse (bands.c); the reference only reserves its flag bit (celt/decoder.rs:1092).  In decode
// order -- band, channel, block -- every block no pulse or fold reached is filled with +-r from the frame's noise
// generator and its band renormalised.  Here: one lane per (band, channel) computes r, the collapsed blocks and, from the
// prefix of their widths, where the band's first draw sits in the generator's sequence (an LCG jump); then the (band,
// channel) pairs are dealt to the waves, which fill and renormalise their bands with no workgroup barrier in between.
struct OgAcScratch { float r[44]; uint32_t steps[44], seed[44]; };      // in the (idle) lowband buffers
OG_DEV void og_anti_collapse(OgBandCtx &c)
{
    const OgCeltHdr *h = c.h;
    OgCeltLds &L = *c.L;
    static_assert(sizeof(OgAcScratch) <= sizeof(L.w.b.s), "scratch fits the lowband buffers");
    OgAcScratch &A = *(OgAcScratch *)&L.w.b.s;
    const int lm = c.lm, M = 1 << lm, C = (h->flags & OG_F_STEREO) ? 2 : 1;
    const int start = h->start, end = h->end, npairs = (end - start) * C;
    const uint8_t (*collapse)[24] = ((const OgBandPlan *)L.w.b.plan_head)->collapse;
    OG_LANES_BEGIN(tid)
    if (tid < npairs) {
        const int i = start + (C == 2 ? tid >> 1 : tid), ch = C == 2 ? tid & 1 : 0;
        const int n0 = L.k.width[i];
        const float thresh = 0.5f * og_fexp2(-0.125f * (float)h->pulses_depth[i]);
        const float sqrt_1 = og_frsq((float)(n0 << lm));
        float ediff = h->energy[ch][i] - h->prev_min[ch][i];
        if (ediff < 0) ediff = 0;
        float r = 2.f * og_fexp2(-ediff);
        if (lm == 3) r *= 1.41421356f;
        if (r > thresh) r = thresh;
        A.r[tid] = r * sqrt_1;
        const uint32_t dead = ~(uint32_t)collapse[ch][i] & ((1u << M) - 1u);      // one bit per block
        A.steps[tid] = (uint32_t)__builtin_popcount(dead) * (uint32_t)n0;
    }
    OG_LANES_END
    const uint32_t seed0 = c.seed;
    OG_LANES_BEGIN(tid)
    if (tid < npairs) {
        uint32_t before = 0;
        for (int u = 0; u < tid; u++) before += A.steps[u];
        A.seed[tid] = og_lcg_jump<12>(before, seed0);       // at most 2 * 800 draws per frame
    }
    OG_LANES_END
    OG_WAVES_BEGIN(w)
    for (int t = w; t < npairs; t += OG_WG / 64) {
        if (OG_U32(A.steps[t]) == 0) continue;              // nothing collapsed in this band
        const int i = start + (C == 2 ? t >> 1 : t), ch = C == 2 ? t & 1 : 0;
        const int n0 = L.k.width[i], n = n0 << lm;
        const float r = A.r[t];
        const uint32_t sb = A.seed[t], dead = ~(uint32_t)collapse[ch][i] & ((1u << M) - 1u);
        float *x = L.X[ch] + (L.k.ebands[i] << lm);
        OG_WAVE_VAR(float, ev);
        OG_WAVE_BEGIN(lane)
        float e = 0.f;
        for (int p = lane; p < n; p += 64) {                // element p = coefficient j of block k
            const int k = p & (M - 1), j = p >> lm;
            float v = x[p];
            if (dead & (1u << k)) {
                const int rank = __builtin_popcount(dead & ((1u << k) - 1u));      // collapsed blocks before this one
                const uint32_t s = og_lcg_jump<8>((uint32_t)(OG_MUL24(rank, n0) + j + 1), sb);
                v = (s & 0x8000) ? r : -r;
                x[p] = v;
            }
            e += v * v;
        }
        OG_WV(ev) = e;
        OG_WAVE_END
        float esum;
        OG_WAVE_SUM(esum, ev)
        const float g = og_frsq(1e-15f + esum);
        OG_WAVE_BEGIN(lane)
        for (int p = lane; p < n; p += 64) x[p] *= g;
        OG_WAVE_END
    }
    OG_WAVES_END
}                                                       // (the generator's state after the frame is not needed: every frame's
                                                        //  header carries its own seed)

// ------------------------------------------------------------------------------------------------
// K3: IMDCT as an n-point complex FFT, n = 60 * 2^k = 15 * 2^(k+2) (RFC 6716 4.3.7), in two or three Stockham passes that
// ping-pong between two LDS buffers: a 15-point pass first (no twiddles in front of a first pass; inside, 3 x 5 by the
// prime-factor mapping, so none there either -- index permutations of registers only), then radix 4 and / or 8.  A pass
// holds its R points in registers as packed complex pairs (wg.h); one to three of the four wavefronts are busy in it, and
// which ones rotates with `rot` (the frame number) so that a CU's SIMDs share the load over its resident workgroups.
// (Fewer, fatter passes are what cuts instructions: this kernel is VALU-bound, and a pass costs its address arithmetic and
// twiddle fetches per work item whatever the radix.  Cross-lane butterflies (DPP) in place of an LDS pass cost MORE VALU
// instructions than the pass they replace -- an exchange is a VALU move per value, an LDS access is not -- so the passes stay.)
OG_DEV OgCf og_cmul(OgCf a, OgCf b) { OgCf r; r.r = a.r * b.r - a.i * b.i; r.i = a.r * b.i + a.i * b.r; return r; }

// floor(q / d) for the small non-negative integers used as work-item indices (q < 4096, d <= 480): exact
OG_DEV int og_div(int q, float inv_d) { return (int)(((float)q + 0.5f) * inv_d); }

OG_DEV void og_dft3(OgC2 &a, OgC2 &b, OgC2 &c)          // forward (e^{-2 pi i / 3}), in place
{
    const OgC2 s = b + c, d = og_c2_scale(b - c, 0.86602540378443864676f);
    const OgC2 m = og_c2_fma(s, -0.5f, a);
    a = a + s;
    b = og_c2_sub_i(m, d);
    c = og_c2_add_i(m, d);
}
OG_DEV void og_dft5(OgC2 &v0, OgC2 &v1, OgC2 &v2, OgC2 &v3, OgC2 &v4)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const OgC2 s14 = v1 + v4, d14 = v1 - v4, s23 = v2 + v3, d23 = v2 - v3;
    const OgC2 p1 = og_c2_fma(s23, c2, og_c2_fma(s14, c1, v0)), p2 = og_c2_fma(s23, c1, og_c2_fma(s14, c2, v0));
    const OgC2 q1 = og_c2_fma(d23, s2, og_c2_scale(d14, s1)), q2 = og_c2_fma(d23, -s1, og_c2_scale(d14, s2));
    v0 = v0 + s14 + s23;
    v1 = og_c2_sub_i(p1, q1);
    v4 = og_c2_add_i(p1, q1);
    v2 = og_c2_sub_i(p2, q2);
    v3 = og_c2_add_i(p2, q2);
}

// One Stockham pass of radix R over TT = 2^tsh transforms of size n.  The transforms are interleaved: element i of
// transform t sits at i * TT + t in both buffers, so a work item q splits into (transform, butterfly j) with a mask and a
// shift and its inputs are q, q + per * TT, ...  Ns = product of the radices already done; k = j mod Ns is the position
// inside the sub-transform (0 in a first pass, j in a last one); outputs go Ns apart from (j - k) R + k.
template <int R, int n, int Ns>      // compile-time sizes: every per-pass constant folds
OG_DEV void og_wg_fft_stage(const OgCf *src_, OgCf *dst_, int tsh, const OgCf *W_, int rot)
{
    static_assert(R == 4 || R == 8 || R == 15, "radices of 15 * 2^k");
    constexpr int per = n / R, wstep = 480 / (Ns * R);
    constexpr bool last = Ns * R == n;
    const OgC2 *src = (const OgC2 *)src_, *W = (const OgC2 *)W_;
    OgC2 *dst = (OgC2 *)dst_;
    const int total = per << tsh, rstep = per << tsh, ostep = Ns << tsh;          // total <= OG_WG: at most one item per lane
    OG_LANES_BEGIN(tid)
    const int q = (tid + rot) & (OG_WG - 1);
    if (q < total) {
        const int t = q & ((1 << tsh) - 1), j = q >> tsh;
        const int k = Ns == 1 ? 0 : last ? j : j - Ns * og_div(j, 1.f / Ns);
        OgC2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = src[q + r * rstep];
        if (Ns > 1) {
            const int kw = OG_MUL24(k, wstep);
#pragma unroll
            for (int r = 1; r < R; r++) v[r] = og_c2_mul(v[r], W[r * kw]);
        }
        OgC2 *o = dst + ((((j - k) * R + k) << tsh) + t);
        if (R == 4) {
            const OgC2 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3 % R], t3 = v[1] - v[3 % R];
            o[0] = t0 + t2;
            o[ostep] = og_c2_sub_i(t1, t3);
            o[2 * ostep] = t0 - t2;
            o[3 * ostep] = og_c2_add_i(t1, t3);
        } else if (R == 8) {
            const float h = 0.70710678118654752440f;
            const OgC2 a0 = v[0] + v[4 % R], a1 = v[0] - v[4 % R], a2 = v[2 % R] + v[6 % R], a3 = v[2 % R] - v[6 % R];
            const OgC2 a4 = v[1] + v[5 % R], a5 = v[1] - v[5 % R], a6 = v[3 % R] + v[7 % R], a7 = v[3 % R] - v[7 % R];
            const OgC2 b0 = a0 + a2, b2 = a0 - a2, b1 = og_c2_sub_i(a1, a3), b3 = og_c2_add_i(a1, a3);
            const OgC2 b4 = a4 + a6, b6 = a4 - a6, b5 = og_c2_sub_i(a5, a7), b7 = og_c2_add_i(a5, a7);
            const OgC2 u5 = og_c2_scale(og_c2_sub_i(b5, b5), h);          //  e^{-i pi/4}  b5
            const OgC2 u7 = og_c2_scale(og_c2_add_i(b7, b7), h);          // -e^{-3i pi/4} b7
            o[0] = b0 + b4;
            o[ostep] = b1 + u5;
            o[2 * ostep] = og_c2_sub_i(b2, b6);
            o[3 * ostep] = b3 - u7;
            o[4 * ostep] = b0 - b4;
            o[5 * ostep] = b1 - u5;
            o[6 * ostep] = og_c2_add_i(b2, b6);
            o[7 * ostep] = b3 + u7;
        } else {
            // 15 = 3 x 5, prime-factor mapping: input n = 5 n1 + 3 n2, output k = 10 k1 + 6 k2 (mod 15) -- five 3-point
            // transforms over n1, then three 5-point ones over n2, in place
#pragma unroll
            for (int n2 = 0; n2 < 5; n2++) og_dft3(v[(3 * n2) % R], v[(5 + 3 * n2) % R], v[(10 + 3 * n2) % R]);
#pragma unroll
            for (int k1 = 0; k1 < 3; k1++)
                og_dft5(v[(5 * k1) % R], v[(5 * k1 + 3) % R], v[(5 * k1 + 6) % R], v[(5 * k1 + 9) % R], v[(5 * k1 + 12) % R]);
#pragma unroll
            for (int k1 = 0; k1 < 3; k1++)
#pragma unroll
                for (int k2 = 0; k2 < 5; k2++) o[((10 * k1 + 6 * k2) % 15) * ostep] = v[(5 * k1 + 3 * k2) % R];
        }
    }
    OG_LANES_END
}

// RFC 6716 celt.c comb_filter, in place, for one channel by one wave (inside an OG_WAVES section): y points at the section's
// sample 0 inside the channel's time buffer (history below it).  y[i] depends on y[i - T - 2 .. i - T + 2], so T - 2 samples
// are independent at a time.  The section has two parts: the cross-fade from the old to the new parameter set over the first
// `overlap` samples (both periods are read: steps of min(T0, T1) - 2 samples through LDS), and the rest with the new set alone.
// For T1 <= 62 the rest runs in registers: a step's T1 - 2 new samples sit in lanes 4 .. T1 + 1 of one register behind the
// previous step's last four (lanes 0 .. 3), the five taps are that register shifted down by 0 .. 4 lanes (DPP), and LDS only
// sees the inputs coming in and the results going out -- nothing waits for an LDS round trip.  Longer periods take steps of up to
// 256 samples through LDS (at most a dozen steps).  Same operations in the same order on every path.
OG_DEV void og_wave_comb_filter(float *y, int T0, int T1, int n, float g0, float g1, int tap0, int tap1,
                                const float *window, const float (*taps)[4])
{
    if (g0 == 0.f && g1 == 0.f) return;
    if (T0 < 15) T0 = 15;
    if (T1 < 15) T1 = 15;
    const float g00 = g0 * taps[tap0][0], g01 = g0 * taps[tap0][1], g02 = g0 * taps[tap0][2];
    const float g10 = g1 * taps[tap1][0], g11 = g1 * taps[tap1][1], g12 = g1 * taps[tap1][2];
    const int overlap = (g0 == g1 && T0 == T1 && tap0 == tap1) ? 0 : 120;
    const int stop = g1 == 0.f ? overlap : n;          // with g1 == 0 only the cross-fade runs
    {
        // only a set with a gain makes a sample depend on earlier output: a silent set's period (15 when the post-filter was
        // off) does not shorten the steps.  A step's samples (at most 128: the whole cross-fade) are fetched together.
        int lim = 1 << 20;
        if (g0 != 0.f && T0 < lim) lim = T0;
        if (g1 != 0.f && T1 < lim) lim = T1;
        int step = lim - 2;
        if (step > 128) step = 128;
        for (int i0 = 0; i0 < overlap; i0 += step) {
            const int i1 = i0 + step < overlap ? i0 + step : overlap;
            OG_WAVE_BEGIN(lane)
            float r[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int iu = i0 + lane + 64 * u, i = iu < i1 ? iu : i0;       // (idle items recompute the step's first sample)
                const float f = window[i] * window[i];
                const float *q0 = y + i - T0, *q1 = y + i - T1;
                r[u] = y[i] + ((1.f - f) * g00) * q0[0]
                     + ((1.f - f) * g01) * (q0[1] + q0[-1])
                     + ((1.f - f) * g02) * (q0[2] + q0[-2])
                     + (f * g10) * q1[0]
                     + (f * g11) * (q1[1] + q1[-1])
                     + (f * g12) * (q1[2] + q1[-2]);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int iu = i0 + lane + 64 * u;
                if (iu < i1) y[iu] = r[u];
            }
            OG_WAVE_END
        }
    }
    if (stop <= overlap) return;
    if (T1 <= 62) {
        const int S = T1 - 2;                           // new samples per step
        OG_WAVE_VAR(float, R);                          // lane l: y[i0 - 4 + l], i0 = first sample of the step to come
        OG_WAVE_VAR(float, X);                          // lane l >= 4: the step's input y[i0 + l - 4]
        OG_WAVE_VAR(float, s1); OG_WAVE_VAR(float, s2); OG_WAVE_VAR(float, s3); OG_WAVE_VAR(float, s4); OG_WAVE_VAR(float, H);
        OG_WAVE_BEGIN(lane)
        const int i = overlap + lane - 4;
        OG_WV(R) = lane < S + 4 ? y[i - S] : 0.f;
        OG_WV(X) = (lane >= 4 && lane < S + 4 && i < stop) ? y[i] : 0.f;
        OG_WV(s1) = OG_WV(s2) = OG_WV(s3) = OG_WV(s4) = OG_WV(H) = 0.f;
        OG_WAVE_END
        for (int i0 = overlap; i0 < stop; i0 += S) {
            OG_WAVE_SHR1(s1, R)
            OG_WAVE_SHR1(s2, s1)
            OG_WAVE_SHR1(s3, s2)
            OG_WAVE_SHR1(s4, s3)
            OG_WAVE_PULL(H, R, S + lane)                // lanes 0 .. 3: this step's history = the last four of R
            OG_WAVE_REGS_BEGIN(lane)
            const int i = i0 + lane - 4;
            const bool mine = lane >= 4 && lane < S + 4;
            // y[i - T1 + k], k = 0, +-1, +-2, are R of lanes l - 2, l - 1 / l - 3, l / l - 4
            const float v = OG_WV(X) + g10 * OG_WV(s2) + g11 * (OG_WV(s1) + OG_WV(s3)) + g12 * (OG_WV(R) + OG_WV(s4));
            if (mine && i < stop) y[i] = v;
            OG_WV(R) = lane < 4 ? OG_WV(H) : v;
            OG_WV(X) = (mine && i + S < stop) ? y[i + S] : 0.f;
            OG_WAVE_REGS_END
        }
        OG_WAVE_FENCE
        return;
    }
    int step = T1 - 2;
    if (step > 256) step = 256;
    for (int i0 = overlap; i0 < stop; i0 += step) {     // at most a dozen steps; a step's (up to four) samples per lane are fetched together
        const int i1 = i0 + step < stop ? i0 + step : stop;
        OG_WAVE_BEGIN(lane)
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int iu = i0 + lane + 64 * u, i = iu < i1 ? iu : i0;
            const float *q1 = y + i - T1;
            r[u] = y[i] + g10 * q1[0] + g11 * (q1[1] + q1[-1]) + g12 * (q1[2] + q1[-2]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int iu = i0 + lane + 64 * u;
            if (iu < i1) y[iu] = r[u];
        }
        OG_WAVE_END
    }
}

// ------------------------------------------------------------------------------------------------
// First half of a frame (stateless: any number of frames of a stream can be in flight): band walk, anti-collapse,
// denormalisation and the IMDCT up to its post-twiddle.  coef_row holds K1's coefficients on entry and the frame's
// time-domain samples before the TDAC window on exit ([channel][960], N used per channel; row 1 is left alone for
// mono output).  Uniform arguments; every lane of the workgroup calls it.  `dbg` pointers are already offset to this
// frame (any may be null).
OG_DEV void og_celt_spectrum(const OgCeltHdr *hg, const OgBandPlan *plan, const OgCeltTables *T, OgCeltLds &L, const OgCeltDebug &dbg,
                             float *coef_row, int stop = 0 /* profiling aid: leave after stage `stop` (tools/stage_insts.sh) */,
                             int rot = 0 /* which wavefronts take the narrow FFT passes: any number, e.g. the frame's */)
{
    float *dbg_norm = dbg.norm, *dbg_freq = dbg.freq;
    uint8_t *dbg_collapse = dbg.collapse;
    OG_STAMP(0)

    // ---- phase 0: stage the side-info, the band plan and the constants in LDS; fetch K1's coefficients.  The frame's scratch
    //      row holds K1's output only where K1 wrote: the leaves of every (band, buffer) that has a PVQ leaf (its zero-pulse
    //      leaves included: opusgpu_sort_pvq_leaves lists them and K1 stores zeros) and the one-bin bands' signs (plan kernel).
    //      Everything else in the row is left over from earlier launches and is replaced by zero here, so nobody has to hand
    //      the rows back zero-filled.
    //      Two rounds of loads, each issued as a whole: (1) everything whose address is known from the kernel arguments -- header,
    //      plan head, first fill ops, constants, the row -- and, per wavefront, the few header / plan words the rest depends on,
    //      which then reach scalar registers by a lane read (a chain of dependent fetches here cost each workgroup several memory
    //      round trips before its first useful instruction); (2) the band of every bin and its sqrt(width), which depend on LM.
    struct OgRow2 { OgF4 v[2], sq[2]; uint32_t bb[2]; };
    struct OgStage { uint32_t hv, pv, ov, sel; float em; uint32_t eb, wd; };
    OG_LANE_VAR(OgRow2, rowv);
    OG_LANE_VAR(OgStage, stv);
    OG_LANE_VAR(uint32_t, selv);
    static_assert(sizeof(OgCeltHdr) / 4 <= OG_WG && OG_PLAN_HEAD_BYTES / 4 <= OG_WG && OG_LDS_OPS * 4 <= OG_WG, "one dword per lane");
    OG_LANES_BEGIN_NOSYNC(tid)
    {
        const uint32_t *src = (const uint32_t *)hg, *ps = (const uint32_t *)plan;       // plan: head, then the first fill ops
        OgStage st = {0, 0, 0, 0, 0.f, 0, 0};
        if (tid < (int)(sizeof(OgCeltHdr) / 4)) st.hv = src[tid];
        if (tid < OG_PLAN_HEAD_BYTES / 4) st.pv = ps[tid];
        if (tid < OG_LDS_OPS * 4) st.ov = ps[OG_PLAN_HEAD_BYTES / 4 + tid];
        if (tid < 24) { st.em = T->k.emeans[tid]; st.eb = T->k.ebands[tid]; st.wd = T->k.width[tid]; }
        const int l = tid & 63;                         // per wavefront: lanes 0..8 header head, 9..13 plain / joint / keep masks, 16..21 band edges
        uint32_t sel = 0;
        if (l < 9) sel = src[l];
        else if (l < 14) sel = ps[offsetof(OgBandPlan, plain) / 4 + (l - 9)];
        else if (l >= 16 && l < 22) sel = ((const uint32_t *)T->k.ebands)[l - 16];
        OG_LV(stv) = st;
        OG_LV(selv) = sel;
#pragma unroll
        for (int j = 0; j < 2; j++) {                   // 2 * 960 / 4 = 480 pieces of 16 bytes; piece p = bins 4p' .. 4p'+3 of buffer p / 240
            const int p = tid + j * OG_WG;
            OG_LV(rowv).v[j] = OgF4{0.f, 0.f, 0.f, 0.f};
            OG_LV(rowv).bb[j] = 0xFFFFFFFFu;
            if (p < 480) {
                OG_LV(rowv).v[j] = ((const OgF4 *)coef_row)[p];
                // round (2) issued early for LM = 3 (20 ms frames, the usual case): fetched again below if the frame's LM differs
                OG_LV(rowv).bb[j] = ((const uint32_t *)T->bin2band[3])[p >= 240 ? p - 240 : p];
                OG_LV(rowv).sq[j] = ((const OgF4 *)T->bin2sq[3])[p >= 240 ? p - 240 : p];
            }
        }
    }
    OG_LANES_END_NOSYNC
    static_assert(offsetof(OgBandPlan, keep) == offsetof(OgBandPlan, plain) + 12 && offsetof(OgBandPlan, joint) == offsetof(OgBandPlan, plain) + 8,
                  "plain[2], joint, keep[2] are consecutive words");
    static_assert(offsetof(OgCeltHdr, lm) == 18 && offsetof(OgCeltHdr, flags) == 19 && offsetof(OgCeltHdr, start) == 20 && offsetof(OgCeltHdr, end) == 21,
                  "header words 4 and 5");
    // which (buffer, band)s of the row K1 / the plan wrote: the plan's keep masks, cut to the coded bands (a silent frame's row
    // is untouched); uniform.  The calls with nothing to transform (most of a non-transient frame's) are finished here as well:
    // their lowband_out = sqrt(N0) x goes to the folding source rows, zero everywhere else (the plan's `plain` masks).
    uint32_t keepm[2], plainm[2];
    int norm_off0, lm0;
    {
        const uint32_t w4 = OG_LANE_GET0(selv, 4), w5 = OG_LANE_GET0(selv, 5);
        lm0 = (int)((w4 >> 16) & 3u);
        const uint32_t flags0 = w4 >> 24, b_lo = w5 & 255u, b_hi = (flags0 & OG_F_SILENCE) ? 0u : (w5 >> 8) & 255u;
        const uint32_t coded = b_hi > b_lo ? ((b_hi >= 21u ? (1u << 21) : (1u << b_hi)) - 1u) & ~((1u << b_lo) - 1u) : 0u;
        plainm[0] = OG_LANE_GET0(selv, 9) & coded;
        plainm[1] = OG_LANE_GET0(selv, 10) & coded;
        // (lane 11: joint)
        keepm[0] = OG_LANE_GET0(selv, 12) & coded;
        keepm[1] = OG_LANE_GET0(selv, 13) & coded;
        const uint32_t bl = b_lo < 21u ? b_lo : 20u;
        norm_off0 = (int)(((OG_LANE_GET0(selv, 16 + (bl >> 2)) >> (8 * (bl & 3u))) & 255u) << lm0);
    }
    OG_LANES_BEGIN(tid)
    {
        const OgStage st = OG_LV(stv);
        if (tid < (int)(sizeof(OgCeltHdr) / 4)) ((uint32_t *)&L.hdr)[tid] = st.hv;
        if (tid < OG_PLAN_HEAD_BYTES / 4) L.w.b.plan_head[tid] = st.pv;
        if (tid < OG_LDS_OPS * 4) ((uint32_t *)L.w.b.ops)[tid] = st.ov;
        if (tid < 24) { L.k.emeans[tid] = st.em; L.k.ebands[tid] = (uint8_t)st.eb; L.k.width[tid] = (uint8_t)st.wd; }
    }
    if (lm0 != 3) {                                     // (uniform)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int p = tid + j * OG_WG;
            if (p < 480) {
                OG_LV(rowv).bb[j] = ((const uint32_t *)T->bin2band[lm0])[p >= 240 ? p - 240 : p];     // the four bins' bands
                OG_LV(rowv).sq[j] = ((const OgF4 *)T->bin2sq[lm0])[p >= 240 ? p - 240 : p];
            }
        }
    }
    for (int p = tid + (960 - norm_off0); p < 800; p += OG_WG) L.R[p] = L.R[800 + p] = 0.f;   // norm bins no piece below covers (none when the frame starts at band 0)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p = tid + j * OG_WG;
        if (p < 480) {
            const OgF4 v = OG_LV(rowv).v[j];
            const OgF4 sq = OG_LV(rowv).sq[j];
            const int slot = p >= 240;
            const uint32_t bb = OG_LV(rowv).bb[j], m = keepm[slot], pm = plainm[slot];
            float e[4] = {v.x, v.y, v.z, v.w}, nv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {               // band 255 (above band 20) reads bit 31, which is clear
                const uint32_t b = (bb >> (8 * k)) & 255u;
                e[k] = ((m >> (b & 31u)) & 1u) ? e[k] : 0.f;
                nv[k] = ((pm >> (b & 31u)) & 1u) ? nv[k] * e[k] : 0.f;
            }
            ((OgF4 *)&L.X[0][0])[p] = OgF4{e[0], e[1], e[2], e[3]};
            const int ni = 4 * (p - 240 * slot) - norm_off0;            // the piece's place in the norm row
            float *nr = og_norm(L, slot);
            if ((norm_off0 & 3) == 0) {
                if (ni >= 0 && ni < 800) *(OgF4 *)(nr + ni) = OgF4{nv[0], nv[1], nv[2], nv[3]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (ni + k >= 0 && ni + k < 800) nr[ni + k] = nv[k];
            }
        }
    }
    OG_LANES_END
    const OgCeltHdr *h = &L.hdr;
    const int lm = h->lm, M = 1 << lm, N = 120 << lm;
    const int C = (h->flags & OG_F_STEREO) ? 2 : 1, CC = (h->flags & OG_F_OUT_STEREO) ? 2 : 1;
    const bool transient = (h->flags & OG_F_TRANSIENT) != 0, silence = (h->flags & OG_F_SILENCE) != 0;

    OG_STAMP(1)
    if (stop == 1) {
        if (dbg_norm) {
            OG_LANES_BEGIN(tid)
            for (int p = tid; p < 2 * 960; p += OG_WG) dbg_norm[p] = (&L.X[0][0])[p];
            OG_LANES_END
        }
        return;
    }
    if (!silence) {
        OG_STAMP(2)
        // ---- phase 2 (K2): bands in order
        OgBandCtx c;
        c.h = h; c.plan = plan; c.T = T; c.L = &L; c.seed = h->seed; c.lm = lm; c.cycles = dbg.cycles; c.stop = stop;
        og_celt_bands(c);
        if (dbg_norm || dbg_collapse) {
            OG_LANES_BEGIN(tid)
            if (dbg_norm) for (int p = tid; p < 2 * 960; p += OG_WG) dbg_norm[p] = (&L.X[0][0])[p];
            if (dbg_collapse && tid < 42) dbg_collapse[tid] = ((const OgBandPlan *)L.w.b.plan_head)->collapse[tid / 21][tid % 21];
            OG_LANES_END
        }
        OG_STAMP(3)
        if (stop >= 2 && stop <= 5) return;
        if (h->flags & OG_F_ANTICOLLAPSE) og_anti_collapse(c);
    }
    OG_STAMP(4)
    if (stop == 6) return;
    OG_BARRIER                                          // the uniform reads of the band-walk tables above are done in every wave
                                                        // before the twiddles below replace them

    // ---- phase 4: denormalise (RFC 6716 4.3.6) + channel up/down-mix, fused into the IMDCT's pre-twiddle.  The first region
    //      stages the FFT twiddles over the (now dead) band-walk tables and the per-(channel, band) gains, and fetches what a
    //      lane's work items need into registers: the IMDCT pre / post twiddle (the same for both: work item q of either pass
    //      has the same bin index) and the bands of the item's two bins
    const int B = transient ? M : 1, NB = N >> (transient ? lm : 0), n4 = NB >> 1;
    OgCf *ldsW = L.w.t.W;
    // the CC * B transforms of the frame are interleaved in the FFT buffers: work item q = bin i of transform t = ch * B + b
    // at q = i * TT + t, TT = CC * B = 2^tsh; it reads MDCT bins 2i and NB-1-2i of block b, which sit at j1, j2 in X[ch]
    const int bsh = transient ? lm : 0, tsh = bsh + (CC == 2), tmask = (1 << tsh) - 1;
    struct OgTw4 { OgC2 w[4]; uint32_t bands[4], xoff[4]; };     // twiddle; gain cells of the item's two bins (gband, flat) | << 8; the bins (X, flat) | << 16
    OG_LANE_VAR(OgTw4, twr);
    OG_LANES_BEGIN(tid)
    for (int t = tid; t < 48; t += OG_WG) {             // per-(channel, band) gains 2^(E + mean), zero outside [start, end)
        const int ch = t >= 24, b = t - 24 * ch;
        float g = 0.f;
        if (!silence && b >= h->start && b < h->end && b < 21) {
            const float lg = h->energy[ch < C ? ch : 0][b] + L.k.emeans[b];
            g = og_fexp2(lg < 32.f ? lg : 32.f);
        }
        L.gband[ch][b] = g;
    }
    {
        // every fetch first (independent: one memory latency for all of them; indices of idle work items are clamped, not
        // branched around), then the LDS stores
        const float *wsrc = (const float *)T->fft_w;
        const uint8_t *b2b = T->bin2band[lm0];
        const OgC2 *twp = (const OgC2 *)T->tw + (n4 - 60);
        const uint32_t items = (uint32_t)(CC * (N >> 1));
        float wv[4];
        uint32_t b1[4], b2[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t p = (uint32_t)tid + (uint32_t)j * OG_WG;
            wv[j] = wsrc[p < 960u ? p : 959u];
            const uint32_t q = p < items ? p : 0u;
            const uint32_t i = q >> tsh, b = (q & (uint32_t)tmask) & (uint32_t)(B - 1);
            const uint32_t j1 = ((2u * i) << bsh) + b, j2 = (uint32_t)(N - B) - ((2u * i) << bsh) + b;
            OG_LV(twr).w[j] = twp[i];
            b1[j] = b2b[j1];
            b2[j] = b2b[j2];
            const uint32_t cx = (C == 2 && CC == 2) ? (q & (uint32_t)tmask) >> bsh : 0u;       // coded channel the output channel reads
            OG_LV(twr).xoff[j] = (cx * 960u + j1) | (cx * 960u + j2) << 16;
            b1[j] |= cx << 8;                                                               // (kept beside the band until it arrives)
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t p = (uint32_t)tid + (uint32_t)j * OG_WG;
            if (p < 960u) ((float *)ldsW)[p] = wv[j];
            const uint32_t cg = (b1[j] >> 8) * 24u, a1 = b1[j] & 255u;
            OG_LV(twr).bands[j] = (cg + (a1 == 255u ? 23u : a1)) | (cg + (b2[j] == 255u ? 23u : b2[j])) << 8;     // gband[.][21..23] = 0
        }
    }
    OG_LANES_END
    OG_STAMP(9)
    if (dbg_freq) {                                     // stage dump: the denormalised spectrum as the separate pass would leave it
        OG_LANES_BEGIN(tid)
        for (int j = tid; j < 960; j += OG_WG) {
            const int bb = T->bin2band[lm0][j], b = bb == 255 ? 23 : bb;
            float v0 = L.X[0][j] * L.gband[0][b];
            const float v1 = C == 2 ? L.X[1][j] * L.gband[1][b] : v0;
            if (CC == 1 && C == 2) v0 = 0.5f * v0 + 0.5f * v1;
            dbg_freq[j] = v0;
            dbg_freq[960 + j] = v1;
        }
        OG_LANES_END
    }
    OG_STAMP(10)

    OG_STAMP(5)
    if (stop == 7) return;
    // ---- phase 5 (K3): IMDCT.  B blocks of NB bins, interleaved with stride B in X.  The FFT ping-pongs between R and X.
    OgCf *bufA = (OgCf *)L.R, *bufB = (OgCf *)&L.X[0][0];
    OG_LANES_BEGIN(tid)
#pragma unroll
    for (int j = 0; j < 4; j++) {                                            // denormalise + pre-twiddle
        const int q = tid + j * OG_WG;
        if (q < CC * (N >> 1)) {
            const uint32_t xo = OG_LV(twr).xoff[j], go = OG_LV(twr).bands[j];
            const float *Xf = &L.X[0][0], *Gf = &L.gband[0][0];
            float x1 = Xf[xo & 0xFFFFu] * Gf[go & 255u], x2 = Xf[xo >> 16] * Gf[go >> 8];
            if (C == 2 && CC == 1) {                                         // stereo coded, mono out
                x1 = 0.5f * x1 + 0.5f * (Xf[960u + (xo & 0xFFFFu)] * Gf[24u + (go & 255u)]);
                x2 = 0.5f * x2 + 0.5f * (Xf[960u + (xo >> 16)] * Gf[24u + (go >> 8)]);
            }
            ((OgC2 *)bufA)[q] = og_c2_mulc(OgC2{x1, x2}, OG_LV(twr).w[j]);     // (x1 w.r + x2 w.i, x2 w.r - x1 w.i)
        }
    }
    OG_LANES_END
    {
        // 60 * 2^k points = passes of radix 15, (4,) 4 or 8; `src` ends up holding the spectrum
        OgCf *src = bufA, *dst = bufB;
        const int r1 = (rot & 3) << 6, r2 = ((rot + 1) & 3) << 6, r3 = ((rot + 2) & 3) << 6;
        switch (n4) {
        case 60:
            og_wg_fft_stage<15, 60, 1>(bufA, bufB, tsh, ldsW, r1);
            og_wg_fft_stage<4, 60, 15>(bufB, bufA, tsh, ldsW, r2);
            break;
        case 120:
            og_wg_fft_stage<15, 120, 1>(bufA, bufB, tsh, ldsW, r1);
            og_wg_fft_stage<8, 120, 15>(bufB, bufA, tsh, ldsW, r2);
            break;
        case 240:
            og_wg_fft_stage<15, 240, 1>(bufA, bufB, tsh, ldsW, r1);
            og_wg_fft_stage<4, 240, 15>(bufB, bufA, tsh, ldsW, r2);
            og_wg_fft_stage<4, 240, 60>(bufA, bufB, tsh, ldsW, r3);
            src = bufB; dst = bufA;
            break;
        default:
            og_wg_fft_stage<15, 480, 1>(bufA, bufB, tsh, ldsW, r1);
            og_wg_fft_stage<4, 480, 15>(bufB, bufA, tsh, ldsW, r2);
            og_wg_fft_stage<8, 480, 60>(bufA, bufB, tsh, ldsW, r3);
            src = bufB; dst = bufA;
            break;
        }
        (void)dst;
        // `src` now holds the spectrum.  Post-twiddle, and straight out to the frame's scratch row in time order: each lane's
        // samples land 8 bytes apart (even positions ascending, odd ones descending), so a wavefront's two stores fill the
        // same cache lines and merge in L2; cheaper for this VALU-bound kernel than a second pass through LDS.
        OG_LANES_BEGIN(tid)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = tid + j * OG_WG;
            if (q < CC * (N >> 1)) {
                const OgC2 v = og_c2_mulc(((const OgC2 *)src)[q], OG_LV(twr).w[j]);          // (Z.r w.r + Z.i w.i, Z.i w.r - Z.r w.i)
                const uint32_t t = (uint32_t)q & (uint32_t)tmask, k = (uint32_t)q >> tsh, ch = t >> bsh, b = t & (uint32_t)(B - 1);
                const uint32_t base = ch * 960u + (uint32_t)OG_MUL24(NB, (int)b);         // 32-bit byte offsets from the row's (uniform) address
                char *row = (char *)coef_row;
                *(float *)(row + ((base + 2u * k) << 2)) = v.y;
                *(float *)(row + ((base + (uint32_t)NB - 1u - 2u * k) << 2)) = -v.x;
            }
        }
        OG_LANES_END
    }
    OG_STAMP(11)
    OG_STAMP(13)
}

// ------------------------------------------------------------------------------------------------
// Second half of the frames, in stream order: TDAC window + overlap-add (RFC 6716 mdct.c), comb post-filter (4.3.7.1),
// de-emphasis, PCM.  A workgroup of OG_WGT lanes walks a RUN of consecutive frames of one stream, one wavefront per channel
// for the recursive parts, and keeps the stream's synthesis state in LDS for the whole run.  Per channel the time buffer is
//     [ history OG_HIST | frame N | IMDCT tail 60 ]        (samples after the comb filter, before de-emphasis)
// the comb filter reads its history straight below the frame; after it the newest OG_HIST + 60 samples slide down by N to
// become the next frame's history and TDAC input, and the de-emphasised output is staged in the slots that frees -- where the
// next frame's row will land -- on its way to the interleaved PCM store.  HBM sees, per frame, the sample row
// in (fetched into registers one frame ahead, while the previous frame filters) and the PCM out; per run, the history the
// comb filters reach back to before the run, and at the end the state write-back (newest <= 1024 samples, IMDCT tail,
// de-emphasis memory, position).
struct alignas(16) OgTailLds {
    float out[2][OG_OUTLEN];          // per channel: [OG_HIST history | frame | tail]
    float window[120];
    float taps[3][4];
    float carry[2][64];               // de-emphasis carries
    float dm[2];                      // the stream's de-emphasis memory, carried from frame to frame of the run
    float pad_[2];
    uint32_t hh[84];                  // the heads (9 dwords each) of nine consecutive headers of the run
};

// (OgHdrHead, celt_plan.h: the head of OgCeltHdr, everything up to the band arrays, is what the tail needs of a header)

// comb post-filter state of a frame (RFC 6716 4.3.7.1): sets [0] old, [1] current, [2] new; how far back it reads
OG_DEV bool og_pf_on(const OgHdrHead &h) { return h.pf_gain_q[0] != 0 || h.pf_gain_q[1] != 0 || ((h.lm & 3) != 0 && h.pf_gain_q[2] != 0); }
OG_DEV int og_pf_reach(const OgHdrHead &h)
{
    if (!og_pf_on(h)) return 0;
    int reach = 0;
    for (int k = 0; k < 3; k++) { int t = h.pf_period[k] < 15 ? 15 : h.pf_period[k]; if (t > reach) reach = t; }
    reach += 2;
    return reach > OG_HIST ? OG_HIST : reach;
}

struct OgTailRow { OgF4 v[4]; };      // a lane's share of a frame's sample row (CC * N / 4 <= 480 pieces over 128 lanes)

template <class Lv>                   // rv: the lane's OgTailRow (OG_LV form)
OG_DEV void og_tail_fetch_row(Lv &rv, const float *coef_row, int tid, int CC, int N)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = tid + j * OG_WGT;
        rv.v[j] = OgF4{0.f, 0.f, 0.f, 0.f};
        if (q < CC * (N >> 2)) {
            const int ch = q >= (N >> 2), i = q - ch * (N >> 2);
            rv.v[j] = ((const OgF4 *)coef_row)[ch * 240 + i];
        }
    }
}

// hg[0 .. nfr): consecutive frames of ONE stream, in time order; their sample rows are og_work_coef(work, f0 + t).
// cycles: profiling stamps, [64] per frame from frame f0 on (may be null).
OG_DEV void og_celt_tail_run(const OgCeltHdr *hg, uint32_t nfr, OgCeltState *states, float *pcm_base, size_t pcm_stride,
                             OgTailLds &L, uint64_t *cycles, void *work, uint32_t f0)
{
    OgHdrHead hd = og_uniform(*(const OgHdrHead *)hg);
    OgCeltState *st = states + hd.stream;
    const bool reset0 = (hd.flags & OG_F_RESET) != 0;
    const uint32_t pos_start = reset0 ? 0u : OG_U32(st->pos);
    // what the run's comb filters read from before its first sample (frames after a reset inside the run read nothing older)
    const OgHdrHead hn1 = nfr > 1 ? og_uniform(*(const OgHdrHead *)(hg + 1)) : hd;
    int R = 0;
    {
        int before = 0;
        for (uint32_t t = 0; t < nfr; t++) {
            const OgHdrHead ht = t == 0 ? hd : t == 1 ? hn1 : og_uniform(*(const OgHdrHead *)(hg + t));
            if (t && (ht.flags & OG_F_RESET)) break;
            const int need = og_pf_reach(ht) - before;
            if (need > R) R = need;
            before += 120 << (ht.lm & 3);
            if (before >= OG_HIST) break;
        }
    }
    const int hgroups = reset0 ? OG_HIST / 4 : (R + 3) >> 2;     // aligned groups of four samples; a reset clears the whole history
    OG_LANE_VAR(OgTailRow, rv);
    {
        const int CC0 = (hd.flags & OG_F_OUT_STEREO) ? 2 : 1, N0 = 120 << (hd.lm & 3);
        const float *row0 = og_work_coef(work, f0);
        OG_TLANES_BEGIN(tid)
        // every load first (independent: their latencies overlap), then the LDS stores
        OgF4 hv[4];
        float ov = 0.f, dm = 0.f;
        og_tail_fetch_row(OG_LV(rv), row0, tid, CC0, N0);
#pragma unroll
        for (int j = 0; j < 4; j++) {                   // CC * hgroups <= 512 pieces
            const int q = tid + j * OG_WGT;
            hv[j] = OgF4{0.f, 0.f, 0.f, 0.f};
            if (q < CC0 * hgroups && !reset0) {
                const int ch = q >= hgroups, g = q - ch * hgroups;
                hv[j] = *(const OgF4 *)&st->hist[ch][(pos_start - 4u * (uint32_t)(g + 1)) & 1023u];     // pos is a multiple of 8
            }
        }
        uint32_t hv9 = 0;
        if (tid < 81) { const uint32_t fr = (uint32_t)tid / 9u; hv9 = ((const uint32_t *)(hg + (fr < nfr ? fr : nfr - 1)))[tid % 9]; }
        if (tid < CC0 * 60 && !reset0) ov = st->overlap[tid >= 60][tid >= 60 ? tid - 60 : tid];       // the previous frame's IMDCT tail
        if (tid < CC0 && !reset0) dm = st->deemph[tid];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = tid + j * OG_WGT;
            if (q < CC0 * hgroups) {
                const int ch = q >= hgroups, g = q - ch * hgroups;
                *(OgF4 *)&L.out[ch][OG_HIST - 4 * (g + 1)] = hv[j];
            }
        }
        if (tid < CC0 * 60) L.out[tid >= 60][OG_HIST + (tid >= 60 ? tid - 60 : tid)] = ov;
        if (tid < 2) L.dm[tid] = dm;
        if (tid < 81) L.hh[tid] = hv9;
        OG_TLANES_END
    }

    uint32_t pos = pos_start, total = 0;
    bool reset_seen = reset0;
    int CCrun = (hd.flags & OG_F_OUT_STEREO) ? 2 : 1;
    const float coef = 0.85000610f;
    const int Lc = 15;
    float aL = 1.f;
    for (int j = 0; j < Lc; j++) aL *= coef;            // coef^Lc
    for (uint32_t t = 0; t < nfr; t++) {
        uint64_t *cyc = cycles ? cycles + (size_t)t * 64 : nullptr;
        // the heads of frames t .. t + 8 wait in LDS (staged nine at a time: a header is a cache line of its own in HBM, and
        // fetched where it is needed it cost the run a memory round trip per frame)
        if (t && (t & 7u) == 0) {
            OG_TLANES_BEGIN(tid)
            uint32_t hv = 0;
            if (tid < 81) { const uint32_t fr = t + (uint32_t)tid / 9u; hv = ((const uint32_t *)(hg + (fr < nfr ? fr : nfr - 1)))[tid % 9]; }
            if (tid < 81) L.hh[tid] = hv;
            OG_TLANES_END
        }
        const OgHdrHead hn = og_uniform(*(const OgHdrHead *)&L.hh[9 * ((t & 7u) + 1)]);
        const int lm = hd.lm & 3, M = 1 << lm, N = 120 << lm;
        const uint32_t flags = hd.flags;
        const int CC = (flags & OG_F_OUT_STEREO) ? 2 : 1;
        const bool transient = (flags & OG_F_TRANSIENT) != 0;
        const int B = transient ? M : 1, NB = N >> (transient ? lm : 0);
        if (CC > CCrun) CCrun = CC;
        if (t && (flags & OG_F_RESET)) {                // a reset inside the run: forget everything older
            pos = 0;
            reset_seen = true;
            OG_TLANES_BEGIN(tid)
            for (int q = tid; q < 2 * (OG_HIST + 60) / 4; q += OG_WGT) {
                const int ch = q >= (OG_HIST + 60) / 4, g = q - ch * ((OG_HIST + 60) / 4);
                *(OgF4 *)&L.out[ch][4 * g] = OgF4{0.f, 0.f, 0.f, 0.f};
            }
            if (tid < 2) L.dm[tid] = 0.f;
            OG_TLANES_END
        }
        OG_STAMP_TO(cyc, 12)
        // ---- the frame's samples (og_celt_spectrum's output, before the window) behind the previous frame's IMDCT tail
        OG_TLANES_BEGIN(tid)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = tid + j * OG_WGT;
            if (q < CC * (N >> 2)) {
                const int ch = q >= (N >> 2), i = q - ch * (N >> 2);
                *(OgF4 *)&L.out[ch][OG_HIST + 60 + 4 * i] = OG_LV(rv).v[j];
            }
        }
        OG_TLANES_END
        // ---- the next frame's row starts its way in now; TDAC mirror (RFC 6716 mdct.c) of this one
        {
            const bool more = t + 1 < nfr;
            const int CCn = (hn.flags & OG_F_OUT_STEREO) ? 2 : 1, Nn = 120 << (hn.lm & 3);
            const float *rown = og_work_coef(work, f0 + t + (more ? 1 : 0));
            OG_TLANES_BEGIN(tid)
            if (more) og_tail_fetch_row(OG_LV(rv), rown, tid, CCn, Nn);
            for (int q = tid; q < CC * B * 60; q += OG_WGT) {
                const int ch = q >= B * 60, r = q - ch * (B * 60), b = og_div(r, 1.f / 60.f), i = r - b * 60;
                float *o = &L.out[ch][OG_HIST + NB * b];
                const float x1 = o[119 - i], x2 = o[i];
                const float w1 = L.window[i], w2 = L.window[119 - i];
                o[i] = w2 * x2 - w1 * x1;
                o[119 - i] = w1 * x2 + w2 * x1;
            }
            OG_TLANES_END
        }
        OG_STAMP_TO(cyc, 7)
        // ---- per channel, by one wave, no workgroup barrier in between: comb filter in place; the slide; de-emphasis t[j] = y[j] +
        //      0.85 t[j-1] as a blocked scan (N/15 lanes x 15 samples: an odd chunk length keeps the lanes' LDS accesses on
        //      distinct banks) into the staging slots (the samples before de-emphasis stay: they are the next frames' history)
        const float pg0 = 0.09375f * hd.pf_gain_q[0], pg1 = 0.09375f * hd.pf_gain_q[1], pg2 = 0.09375f * hd.pf_gain_q[2];
        const bool pf_on = og_pf_on(hd);
        const int nl = N / Lc;                          // 8 << lm lanes
        OG_TWAVES_BEGIN(w)
        if (w < CC) {
            const int ch = w;
            float *y0 = &L.out[ch][OG_HIST];
            if (pf_on) {
                og_wave_comb_filter(y0, hd.pf_period[0], hd.pf_period[1], 120, pg0, pg1, hd.pf_tapset[0], hd.pf_tapset[1], L.window, L.taps);
                if (lm != 0)
                    og_wave_comb_filter(y0 + 120, hd.pf_period[1], hd.pf_period[2], N - 120, pg1, pg2, hd.pf_tapset[1], hd.pf_tapset[2],
                                        L.window, L.taps);
            }
            if (w == 0) OG_STAMP_TO(cyc, 20)
            // the slide: the newest OG_HIST + 60 samples move down by N (every piece is read before any is written)
            {
                const int P = (OG_HIST + 60) / 4;       // 271 pieces of four samples, at most five per lane
                struct OgSlide { OgF4 v[5]; };
                OG_WAVE_VAR(OgSlide, mv);
                OG_WAVE_BEGIN(lane)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int g = lane + 64 * j;
                    OG_WV(mv).v[j] = g < P ? *(const OgF4 *)&L.out[ch][N + 4 * g] : OgF4{0.f, 0.f, 0.f, 0.f};
                }
                OG_WAVE_END
                OG_WAVE_BEGIN(lane)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int g = lane + 64 * j;
                    if (g < P) *(OgF4 *)&L.out[ch][4 * g] = OG_WV(mv).v[j];
                }
                OG_WAVE_END
            }
            // de-emphasis of the frame (now at [OG_HIST - N, OG_HIST)) into the staging slots behind the IMDCT tail, where the
            // next frame's row will land
            const float *ys = &L.out[ch][OG_HIST - N];
            float *z0 = &L.out[ch][OG_HIST + 60];
            struct OgChunk { float v[15]; };
            OG_WAVE_VAR(OgChunk, yv);                   // the lane's 15 samples, read once for both passes
            OG_WAVE_BEGIN(lane)
            const int l = lane;
            if (l < nl) {
                const float *y = ys + l * Lc;
                float m = 0.f, v = 0.f;
#pragma unroll
                for (int j = 0; j < 15; j++) { OG_WV(yv).v[j] = y[j]; v = y[j] + m; m = coef * v; }
                L.carry[ch][l] = m;                     // chunk's own contribution to the running memory
            }
            OG_WAVE_END
            // carry-in of chunk l = sum over earlier chunks q of carry[q] * coef^(Lc*(l-1-q)) (+ the stream's memory); terms
            // more than `back` chunks back are below 1e-9 of their chunk's value and are dropped
            const int back = 128 / Lc + 1;
            OG_WAVE_BEGIN(lane)
            const int l = lane;
            if (l < nl) {
                const int cnt = l > back ? back : l;    // earlier chunks that count: l - cnt .. l - 1
                float m = l > back ? 0.f : L.dm[ch];    // the stream's de-emphasis memory (0 after a reset)
                float cq[back];                         // their carries, fetched together (the Horner steps below then wait for none)
#pragma unroll
                for (int j = 0; j < back; j++) cq[j] = L.carry[ch][l - 1 - j >= 0 ? l - 1 - j : 0];
#pragma unroll
                for (int j = back - 1; j >= 0; j--)
                    if (j < cnt) m = m * aL + cq[j];    // oldest first
                float *z = z0 + l * Lc;
#pragma unroll
                for (int j = 0; j < 15; j++) { const float v = OG_WV(yv).v[j] + m; m = coef * v; z[j] = v; }
                if (l == nl - 1) L.dm[ch] = m;          // read above by every lane before this wave-ordered store
            }
            OG_WAVE_END
        }
        OG_TWAVES_END
        OG_STAMP_TO(cyc, 21)
        float *pcm = pcm_base + (size_t)hd.stream * pcm_stride + (size_t)hd.pcm_off * CC;
        const bool accum = (flags & OG_F_ACCUM) != 0;
        const int xfade = (hd.spread >> OG_X_SHIFT) & 3;   // redundant frames (opusgpu_sideinfo.h OG_X_*): rare, scalar stores
        const bool vec = (((uintptr_t)pcm) & 15) == 0 && xfade == OG_X_NONE;      // 16-byte stores when the caller's row allows it
        const float *t0 = &L.out[0][OG_HIST + 60], *t1 = &L.out[1][OG_HIST + 60];
        OG_TLANES_BEGIN(tid)
        const float k = 1.f / 32768.f;
        if (vec) {
            for (int q = tid; q < N * CC / 4; q += OG_WGT) {
                OgF4 v;
                if (CC == 2) {                          // two stereo samples
                    const int j = 2 * q;
                    v = OgF4{t0[j] * k, t1[j] * k, t0[j + 1] * k, t1[j + 1] * k};
                } else {
                    const OgF4 u = ((const OgF4 *)t0)[q];
                    v = OgF4{u.x * k, u.y * k, u.z * k, u.w * k};
                }
                if (accum) { const OgF4 o = ((const OgF4 *)pcm)[q]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                ((OgF4 *)pcm)[q] = v;
            }
        } else {
            for (int q = tid; q < N * CC; q += OG_WGT) {
                const int j = CC == 2 ? q >> 1 : q, ch = CC == 2 ? q & 1 : 0;
                const float v = (ch ? t1 : t0)[j] * k;
                if (xfade == OG_X_NONE) {
                    pcm[q] = accum ? pcm[q] + v : v;
                } else {
                    // out = a v + b pcm: the smooth_fade of RFC 6716 4.5.1.4 with the squared CELT window over samples 120 .. 239
                    const float w = j >= 120 && j < 240 ? L.window[j - 120] * L.window[j - 120] : 0.f;
                    float a, b;
                    if (xfade == OG_X_RED_HEAD) { a = j < 120 ? 1.f : 1.f - w; b = j < 120 ? 0.f : w; }
                    else if (xfade == OG_X_AFTER_RED_HEAD) { a = j < 120 ? 0.f : j < 240 ? w : 1.f; b = 1.f; }
                    else { a = w; b = 1.f - w; }                            // OG_X_RED_TAIL (a = 0, b = 1 for j < 120)
                    if (!(xfade == OG_X_RED_TAIL && j < 120) && j < (xfade == OG_X_AFTER_RED_HEAD ? N : 240)) pcm[q] = a * v + b * pcm[q];
                }
            }
        }
        OG_TLANES_END                                   // the staging slots are read before the next frame's row lands in them
        pos += (uint32_t)N;
        total += (uint32_t)N;
        hd = hn;
        OG_STAMP_TO(cyc, 8)
    }

    // ---- state write-back, once per run: the newest samples (all 1024 after a reset), the IMDCT tail awaiting the next
    //      frame, the de-emphasis memory and the position
    const int nh = (reset_seen || total > (uint32_t)OG_HIST) ? OG_HIST : (int)total, hq = nh >> 2;
    OG_TLANES_BEGIN(tid)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = tid + j * OG_WGT;
        if (q < CCrun * hq) {
            const int ch = q >= hq, g = q - ch * hq;
            *(OgF4 *)&st->hist[ch][(pos - 4u * (uint32_t)(g + 1)) & 1023u] = *(const OgF4 *)&L.out[ch][OG_HIST - 4 * (g + 1)];
        }
    }
    if (tid < CCrun * 15) {
        const int ch = tid >= 15, l = tid - 15 * ch;
        *(OgF4 *)&st->overlap[ch][4 * l] = *(const OgF4 *)&L.out[ch][OG_HIST + 4 * l];
    }
    if (tid < CCrun) st->deemph[tid] = L.dm[tid];
    if (tid == 0) st->pos = pos;
    OG_TLANES_END                                       // the next run of the workgroup restages the buffer
}

// A group of the tail kernel: frames hg[0 .. nfr) in time order, cut into runs of one stream each.
OG_DEV void og_celt_tail_group(const OgCeltHdr *hg, uint32_t nfr, OgCeltState *states, float *pcm_base, size_t pcm_stride,
                               OgTailLds &L, uint64_t *cycles, void *work, uint32_t f0)
{
    uint32_t t = 0;
    while (t < nfr) {
        const uint32_t s = OG_U32(hg[t].stream);
        uint32_t e = t + 1;
        while (e < nfr && OG_U32(hg[e].stream) == s) e++;
        og_celt_tail_run(hg + t, e - t, states, pcm_base, pcm_stride, L, cycles ? cycles + (size_t)t * 64 : nullptr, work, f0 + t);
        t = e;
    }
}
