// K1: PVQ leaf expansion (lu-zero/opus src/celt/decoder.rs: cwrsi :541-660, decode_pulses :662-673, unquantize :766-786,
// exp_rotation :676-721, extract_collapse_mask :723-736).  One lane expands one leaf; the vector lives at x[0], x[xs],
// x[2*xs] ... so a wavefront can keep 64 leaves lane-interleaved in LDS (xs = 64: bank = lane, conflict-free).
#pragma once
#include "wg.h"
#include "celt_frame_types.h"
#include "../../../include/opusgpu.h"

struct alignas(16) OgF4s { float x, y, z, w; };
struct alignas(8) OgF2s { float x, y; };
OG_DEV uint32_t og_lcg(uint32_t s) { return 1664525u * s + 1013904223u; }   // celt/decoder.rs:1595-1599

// ------------------------------------------------------------------------------------------------
// K1: one lane expands one PVQ leaf.  Serial in N by nature (celt/decoder.rs:541-660); the
// parallelism is across the ~10^5 leaves of a batch.
#define OG_U2_STRIDE 177              // U2[lo * 177 + hi] = U(lo, hi), lo = min(n, k) <= 14, hi = max(n, k) <= 176
OG_DEV uint32_t og_u2(const uint32_t *U2, int n, int k)
{
    const int lo = n < k ? n : k, hi = n < k ? k : n;
    return U2[OG_MUL24(lo, OG_U2_STRIDE) + hi];         // (small factors: the full-rate 24-bit multiply)
}

// cwrsi (celt/decoder.rs:541-660) as one branch-uniform loop, so that the 64 leaves of a wavefront -- each in a
// different place of its own vector -- execute the same instructions.  The reference's two cases ("lots of pulses" /
// "lots of dimensions") are two ways of walking the same search: with (n, k) dimensions / pulses left, the sign is
// i >= U(n, k+1), and the number of pulses left after this coefficient is the largest k' <= k with U(n, k') <= i.
// Every iteration probes two table entries whose addresses do not depend on each other:
//   phase 0 (new coefficient): U(n, k) and U(n, k+1) -> sign, and a zero coefficient is finished in this iteration;
//   phase 1 (search):          U(n, kk) and U(n, kk-1) -> two candidates per iteration.
// Writes the pulse vector as floats to x[0], x[xs], ...; returns ||y||^2; *cm_out = collapse mask over `blocks` blocks.
OG_DEV uint32_t og_cwrsi(const uint32_t *U2, int n, int k, uint32_t i, float *x, int xs, int logb, int *cm_out, int16_t *dbg)
{
    uint32_t yy = 0;
    int cm = 0, pos = 0, s = 0, k0 = k, kk = k, phase = 0;
    const int nb = n >> logb, blocks = 1 << logb;
    int bl = nb, bi = 0;
#define OG_PUT(v)                                                                                                   \
    {                                                                                                               \
        x[pos * xs] = (float)(v);                                                                                   \
        if (dbg) dbg[pos] = (int16_t)(v);                                                                           \
        yy += (uint32_t)OG_MUL24((v), (v));                                 /* |v| <= 176 */                        \
        if (v) cm |= 1 << bi;                                                                                       \
        pos++;                                                                                                      \
        if (--bl == 0) { bl = nb; bi++; }                                                                           \
    }
    while (n > 2 && k > 0) {
        // straight-line: both phases' outcomes are computed and selected, so that the wavefront's lanes -- each in its own
        // phase -- run one instruction stream instead of three masked ones (the branches cost more than the arithmetic)
        const int kb = phase ? (kk > 0 ? kk - 1 : 0) : kk + 1;
        const uint32_t pA = og_u2(U2, n, kk), pB = og_u2(U2, n, kb);
        // phase 0 (new coefficient)
        const int s0 = -(int)(i >= pB);
        const uint32_t i0 = i - (pB & (uint32_t)s0);
        const bool zero0 = !s0 && pA <= i0;                                     // this coefficient is zero
        // phase 1 (search)
        const bool atA = pA <= i, atB = pB <= i, hit1 = atA || atB;
        const int found1 = atA ? kk : kk - 1;
        const uint32_t i1 = i - (atA ? pA : pB);
        // the step taken
        const bool p1 = phase != 0;
        const bool emit = p1 ? hit1 : zero0;
        s = p1 ? s : s0;
        k0 = p1 ? k0 : k;
        const int val = (p1 && hit1) ? ((k0 - found1 + s) ^ s) : 0;
        i = p1 ? (hit1 ? i1 : i) : (zero0 ? i0 - pA : i0);
        k = (p1 && hit1) ? found1 : k;
        kk = p1 ? (hit1 ? found1 : kk - 2) : (zero0 ? kk : k - 1);             // (phase 0 with pA > i: at least one pulse)
        phase = p1 ? (hit1 ? 0 : 1) : (zero0 ? 0 : 1);
        if (emit) { OG_PUT(val) n--; }
    }
    for (; n > 2; n--) OG_PUT(0)                                               // every pulse placed: the rest is zeros
    {
        uint32_t p = 2 * (uint32_t)k + 1;                                      // n == 2
        s = -(int)(i >= p);
        i -= p & (uint32_t)s;
        k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (uint32_t)k - 1;
        int val = (k0 - k + s) ^ s;
        OG_PUT(val)
        s = -(int)i;                                                            // n == 1
        val = (k + s) ^ s;
        OG_PUT(val)
    }
#undef OG_PUT
    *cm_out = blocks <= 1 ? 1 : cm;
    return yy;
}

// The same walk for the leaves that are mostly zeros (N > 24 bins carry 3-4 pulses on average at the bench's rate, and a lane
// walks up to 176 positions one iteration each: the longest chain of the whole launch): a new coefficient that is zero looks at
// the next position in the same iteration -- the probes U(n-1, k), U(n-1, k+1) do not depend on the first pair -- and two zeros
// are finished per iteration.  Everything else is og_cwrsi.
OG_DEV uint32_t og_cwrsi_sparse(const uint32_t *U2, int n, int k, uint32_t i, float *x, int xs, int logb, int *cm_out, int16_t *dbg)
{
    uint32_t yy = 0;
    int cm = 0, pos = 0, s = 0, k0 = k, kk = k, phase = 0;
    const int nb = n >> logb, blocks = 1 << logb;
    int bl = nb, bi = 0;
#define OG_PUT(v)                                                                                                   \
    {                                                                                                               \
        x[pos * xs] = (float)(v);                                                                                   \
        if (dbg) dbg[pos] = (int16_t)(v);                                                                           \
        yy += (uint32_t)OG_MUL24((v), (v));                                 /* |v| <= 176 */                        \
        if (v) cm |= 1 << bi;                                                                                       \
        pos++;                                                                                                      \
        if (--bl == 0) { bl = nb; bi++; }                                                                           \
    }
    while (n > 2 && k > 0) {
        const int kb = phase ? (kk > 0 ? kk - 1 : 0) : kk + 1;
        const uint32_t pA = og_u2(U2, n, kk), pB = og_u2(U2, n, kb);
        const uint32_t pC = og_u2(U2, n - 1, k), pD = og_u2(U2, n - 1, k + 1);  // the next position's pair (used by a new coefficient only: kk == k there)
        // phase 0 (new coefficient)
        const int s0 = -(int)(i >= pB);
        const uint32_t i0 = i - (pB & (uint32_t)s0);
        const bool zero0 = !s0 && pA <= i0;                                     // this coefficient is zero
        const uint32_t iz = i0 - pA;                                            // ... and the index the next one starts from
        const bool p1 = phase != 0;
        const bool zero1 = !p1 && zero0 && n > 3 && iz < pD && pC <= iz;        // the next one is zero as well
        // phase 1 (search)
        const bool atA = pA <= i, atB = pB <= i, hit1 = atA || atB;
        const int found1 = atA ? kk : kk - 1;
        const uint32_t i1 = i - (atA ? pA : pB);
        // the step taken
        const bool emit = p1 ? hit1 : zero0;
        s = p1 ? s : s0;
        k0 = p1 ? k0 : k;
        const int val = (p1 && hit1) ? ((k0 - found1 + s) ^ s) : 0;
        i = p1 ? (hit1 ? i1 : i) : (zero0 ? (zero1 ? iz - pC : iz) : i0);
        k = (p1 && hit1) ? found1 : k;
        kk = p1 ? (hit1 ? found1 : kk - 2) : (zero0 ? kk : k - 1);
        phase = p1 ? (hit1 ? 0 : 1) : (zero0 ? 0 : 1);
        if (emit) { OG_PUT(val) n--; }
        if (zero1) { OG_PUT(0) n--; }
    }
    for (; n > 2; n--) OG_PUT(0)                                               // every pulse placed: the rest is zeros
    {
        uint32_t p = 2 * (uint32_t)k + 1;                                      // n == 2
        s = -(int)(i >= p);
        i -= p & (uint32_t)s;
        k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (uint32_t)k - 1;
        int val = (k0 - k + s) ^ s;
        OG_PUT(val)
        s = -(int)i;                                                            // n == 1
        val = (k + s) ^ s;
        OG_PUT(val)
    }
#undef OG_PUT
    *cm_out = blocks <= 1 ? 1 : cm;
    return yy;
}

// The same walk with P table probes per iteration instead of two, for the leaves whose time goes into the search: N <= 8 bins
// carry 19 pulses on average at the bench's rate, and with two probes a coefficient of m pulses costs 1 + ceil(m / 2)
// iterations.  With b = k + 1 (new coefficient) or b = kk (search), the probes are U(n, b), U(n, b-1) .. U(n, b-P+1), clamped at
// U(n, 0) = 0: a new coefficient takes its sign from the first and is finished in the same iteration if it has fewer than P - 1
// pulses; a search iteration settles P candidates.  P = 2 is og_cwrsi above, instruction for instruction.
template <int P>
OG_DEV uint32_t og_cwrsi_wide(const uint32_t *U2, int n, int k, uint32_t i, float *x, int xs, int logb, int *cm_out, int16_t *dbg)
{
    uint32_t yy = 0;
    int cm = 0, pos = 0, s = 0, k0 = k, kk = k, phase = 0;
    const int nb = n >> logb, blocks = 1 << logb;
    int bl = nb, bi = 0;
#define OG_PUT(v)                                                                                                   \
    {                                                                                                               \
        x[pos * xs] = (float)(v);                                                                                   \
        if (dbg) dbg[pos] = (int16_t)(v);                                                                           \
        yy += (uint32_t)OG_MUL24((v), (v));                                 /* |v| <= 176 */                        \
        if (v) cm |= 1 << bi;                                                                                       \
        pos++;                                                                                                      \
        if (--bl == 0) { bl = nb; bi++; }                                                                           \
    }
    while (n > 2 && k > 0) {
        const bool p1 = phase != 0;
        const int b = p1 ? kk : k + 1;
        int kj[P];
        uint32_t pr[P];
#pragma unroll
        for (int j = 0; j < P; j++) { kj[j] = b - j > 0 ? b - j : 0; pr[j] = og_u2(U2, n, kj[j]); }
        const int s0 = -(int)(i >= pr[0]);                                      // a new coefficient's sign
        const uint32_t iref = p1 ? i : i - (pr[0] & (uint32_t)s0);
        // the first probe at or below the index (for a new coefficient the first probe is the sign's, not a candidate)
        bool hit = false;
        int found = 0;
        uint32_t sub = 0;
#pragma unroll
        for (int j = P - 1; j >= 0; j--) {
            const bool h = pr[j] <= iref && (j > 0 || p1) && (j != 1 || p1 || !s0);
            hit = hit || h;
            found = h ? kj[j] : found;
            sub = h ? pr[j] : sub;
        }
        s = p1 ? s : s0;
        k0 = p1 ? k0 : k;
        const int val = (k0 - found + s) ^ s;                                   // (used on a hit only)
        i = iref - (hit ? sub : 0u);
        k = hit ? found : k;
        kk = b - P;
        phase = hit ? 0 : 1;
        if (hit) { OG_PUT(val) n--; }
    }
    for (; n > 2; n--) OG_PUT(0)                                               // every pulse placed: the rest is zeros
    {
        uint32_t p = 2 * (uint32_t)k + 1;                                      // n == 2
        s = -(int)(i >= p);
        i -= p & (uint32_t)s;
        k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (uint32_t)k - 1;
        int val = (k0 - k + s) ^ s;
        OG_PUT(val)
        s = -(int)i;                                                            // n == 1
        val = (k + s) ^ s;
        OG_PUT(val)
    }
#undef OG_PUT
    *cm_out = blocks <= 1 ? 1 : cm;
    return yy;
}

// RFC 6716 vq.c exp_rotation1 (celt/decoder.rs:676-692).  The reference loops walk x[i], x[i+stride] in place; the
// `stride` residue classes are independent first-order recurrences, so each runs with its carry in a register
// (same order within a class; each step's two outputs are one multiply and one fused multiply-add, so the carried value
// sees one operation per step).  Four steps are loaded, rotated and stored at a time so that the LDS latency is paid per
// block, not per coefficient (eight at a time cost the kernel a wavefront per SIMD in registers: celt_launch.hip).
OG_DEV void og_rot_fwd(float *x, int xs, int i, int stride, int cnt, float c, float s)      // steps at i, i+stride, ...
{
    float t = x[i * xs];
    while (cnt >= 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = x[(i + (j + 1) * stride) * xs];
#pragma unroll
        for (int j = 0; j < 4; j++) { const float x2 = v[j]; v[j] = fmaf(c, t, -(s * x2)); t = fmaf(s, t, c * x2); }
#pragma unroll
        for (int j = 0; j < 4; j++) x[(i + j * stride) * xs] = v[j];
        i += 4 * stride;
        cnt -= 4;
    }
    for (; cnt > 0; cnt--, i += stride) {
        const float x2 = x[(i + stride) * xs];
        x[i * xs] = fmaf(c, t, -(s * x2));
        t = fmaf(s, t, c * x2);
    }
    x[i * xs] = t;
}

OG_DEV void og_rot_bwd(float *x, int xs, int i, int stride, int cnt, float c, float s)      // steps at i, i-stride, ...
{
    float u = x[(i + stride) * xs];
    while (cnt >= 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = x[(i - j * stride) * xs];
#pragma unroll
        for (int j = 0; j < 4; j++) { const float x1 = v[j]; v[j] = fmaf(c, u, s * x1); u = fmaf(-s, u, c * x1); }
#pragma unroll
        for (int j = 0; j < 4; j++) x[(i - j * stride + stride) * xs] = v[j];
        i -= 4 * stride;
        cnt -= 4;
    }
    for (; cnt > 0; cnt--, i -= stride) {
        const float x1 = x[i * xs];
        x[(i + stride) * xs] = fmaf(c, u, s * x1);
        u = fmaf(-s, u, c * x1);
    }
    x[(i + stride) * xs] = u;
}

OG_DEV void og_exp_rotation1(float *x, int xs, int len, int stride, float c, float s)
{
    // class r of the forward pass (i = 0 .. len-stride-1) has q + (r < rem) steps, len - stride = q stride + rem; the backward
    // pass (i = len-2 stride-1 .. 0) likewise with len - 2 stride: one division per pass, none per class
    const int nf = len - stride;
    if (nf > 0) {
        const int q = stride == 1 ? nf : og_idiv_small(nf, stride), rem = nf - q * stride;
        for (int r = 0; r < stride && r < nf; r++) og_rot_fwd(x, xs, r, stride, q + (r < rem), c, s);
    }
    const int nbk = len - 2 * stride;
    if (nbk > 0) {
        const int q = stride == 1 ? nbk : og_idiv_small(nbk, stride), rem = nbk - q * stride;
        for (int r = 0; r < stride && r < nbk; r++) {
            const int cnt = q + (r < rem);
            og_rot_bwd(x, xs, r + (cnt - 1) * stride, stride, cnt, c, s);     // from the largest i < len - 2 stride with i = r (mod stride)
        }
    }
}

// cos(pi/2 t), 0 <= t <= 1: the Taylor series to x^14 (truncation < 1e-8 at x = pi/2), Horner in x^2
OG_DEV float og_cos_halfpi(float t)
{
    const float x = 1.57079632679489661923f * t, z = x * x;
    float p = -1.f / 87178291200.f;
    p = fmaf(p, z, 1.f / 479001600.f);
    p = fmaf(p, z, -1.f / 3628800.f);
    p = fmaf(p, z, 1.f / 40320.f);
    p = fmaf(p, z, -1.f / 720.f);
    p = fmaf(p, z, 1.f / 24.f);
    p = fmaf(p, z, -0.5f);
    return fmaf(p, z, 1.f);
}

OG_DEV void og_exp_rotation(float *x, int xs, int len, int logb, int k, int spread)   // celt/decoder.rs:694-721, stride = 2^logb blocks
{
    if (2 * k >= len || spread == 0) return;
    const int stride = 1 << logb;
    const float gain = (float)len * og_frcp((float)(len + (20 - 5 * spread) * k));
    const float theta = 0.5f * gain * gain;
    const float c = og_cos_halfpi(theta), s = og_cos_halfpi(1.f - theta);
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    len >>= logb;
    for (int i = 0; i < stride; i++) {
        if (stride2) og_exp_rotation1(x + i * len * xs, xs, len, stride2, s, c);
        og_exp_rotation1(x + i * len * xs, xs, len, 1, c, s);
    }
}

// One PVQ leaf end to end (celt/decoder.rs:662-673 decode_pulses + :766-786 unquantize): pulses -> unit-norm * gain ->
// spreading rotation, in x[0], x[xs], ... ; returns the collapse mask (already shifted to the leaf's place in its band).
#if defined(__HIP__)
#define OG_PVQ_STAMP(k) { if (stamps) stamps[k] = (uint64_t)__builtin_readcyclecounter(); }
#else
#define OG_PVQ_STAMP(k) {}
#endif
template <int P = 2, bool ROT = true>  // P: table probes per cwrsi iteration; 0: the form for sparse leaves (og_cwrsi_sparse).  ROT: rotate here
OG_DEV uint32_t og_pvq_leaf(const uint32_t *U2, const OgCeltLeaf &lf, int spread, float *x, int xs, int16_t *dbg,
                            uint64_t *stamps /* profiling aid, may be null */)
{
    OG_PVQ_STAMP(0)
    const int logb = lf.meta & OG_L_LOGB_MASK;
    int cm;
    uint32_t yy;
    if constexpr (P == 2) yy = og_cwrsi(U2, lf.n, lf.k, lf.index, x, xs, logb, &cm, dbg);
    else if constexpr (P == 0) yy = og_cwrsi_sparse(U2, lf.n, lf.k, lf.index, x, xs, logb, &cm, dbg);
    else yy = og_cwrsi_wide<P>(U2, lf.n, lf.k, lf.index, x, xs, logb, &cm, dbg);
    OG_PVQ_STAMP(1)
    const float g = lf.gain * og_frsq((float)yy);
    for (int i = 0; i < lf.n; i++) x[i * xs] *= g;
    OG_PVQ_STAMP(2)
    if constexpr (ROT) og_exp_rotation(x, xs, lf.n, logb, lf.k, spread);
    return (uint32_t)cm << lf.cmshift;
}

// K1 body for one lane.  xl = this lane's vector storage (element stride xs), at least lf.n elements.
// ref.frame carries the frame's `spread` in its top bits (opusgpu_sideinfo.h); lf = leaves[ref.leaf], fetched by the caller.
template <int P = 2>
OG_DEV void og_pvq_lane(const OgLeafRef ref, const OgCeltLeaf lf, void *work,
                        const uint32_t *U2, float *xl, int xs, int16_t *dbg_pulses, uint64_t *stamps)
{
    const uint32_t frame = ref.frame & OG_REF_FRAME_MASK;
    const int ch = (lf.meta & OG_L_CH_Y) ? 1 : 0;
    const bool zeros = lf.k == 0;                       // a zero-pulse leaf next to PVQ leaves in its (band, buffer): K1 owns the
                                                        // zeros there (live fill leaves are rewritten by the frame kernel); no collapse mask
    uint32_t cm = 0;
    if (!zeros) {
        const int spread = (int)(ref.frame >> OG_REF_SPREAD_SHIFT);
        int16_t *dbg = dbg_pulses ? dbg_pulses + (size_t)frame * 1920 + ch * 960 + lf.pos : nullptr;
        cm = og_pvq_leaf<P>(U2, lf, spread, xl, xs, dbg, stamps);
    }
    // the leaf's n values out, in the widest pieces its place in the row allows: one value to an even place, two to a
    // multiple of four, then fours, then the rest (a wavefront's leaves lie in 64 different rows: a store instruction costs
    // the memory pipeline the same whether a lane writes 4 or 16 bytes)
    float *dst = og_work_coef(work, frame) + ch * 960 + lf.pos;
    const int n = lf.n;
    int i = 0;
#define OG_XV(j) (zeros ? 0.f : xl[(j) * xs])
    if ((lf.pos & 1) && i < n) { dst[i] = OG_XV(i); i++; }
    if (((lf.pos + i) & 2) && i + 2 <= n) { *(OgF2s *)(dst + i) = OgF2s{OG_XV(i), OG_XV(i + 1)}; i += 2; }
    for (; i + 4 <= n; i += 4) *(OgF4s *)(dst + i) = OgF4s{OG_XV(i), OG_XV(i + 1), OG_XV(i + 2), OG_XV(i + 3)};
    if (i + 2 <= n) { *(OgF2s *)(dst + i) = OgF2s{OG_XV(i), OG_XV(i + 1)}; i += 2; }
    if (i < n) dst[i] = OG_XV(i);
#undef OG_XV
    if (zeros) return;
    OG_ATOMIC_OR(og_work_cm(work, frame) + ch * 24 + lf.band, cm);
    OG_PVQ_STAMP(3)
}
