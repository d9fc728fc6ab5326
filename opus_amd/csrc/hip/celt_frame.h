// CELT frame decode, device side: one workgroup turns one frame's side-info (OgCeltHdr + leaves) into
// PCM.  Fuses what SURVEY.md section 2.1 lists as K1 (PVQ index -> pulse vector, normalise, spreading
// rotation, collapse mask), K2 (folding / noise fill, Haar + Hadamard reordering, stereo merge) and K3
// (anti-collapse, denormalise, IMDCT, TDAC window, comb post-filter, de-emphasis): coefficients never
// leave LDS between the stages.
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

struct alignas(16) OgF4 { float x, y, z, w; };

struct OgTask {                       // one quant_band call = (band, buffer); 16 bytes in LDS
    uint16_t boff, n;                 // first bin in the channel buffer, width
    uint16_t n0p, cm;                 // OgXform.n0p; final collapse mask (PVQ-only tasks)
    uint8_t buf, flags, hm, slog;     // OgXform.hm / .slog
    uint8_t m, norm_row, pad0, pad1;  // popcount(hm); norm row lowband_out goes to (0xFF = none)
};
enum { OG_T_ACTIVE = 1, OG_T_CHAIN = 2, OG_T_LONG = 8 };

struct alignas(16) OgCeltLds {
    float X[2][960];                  // coefficients (bands) -> MDCT bins -> FFT ping buffer -> de-emphasised PCM
    union {
        float norm[2][800];           // folding source (celt/decoder.rs `norm_mid`/`norm_side`)
        OgCf A[2][480];               // FFT pong buffer
    } u;
    union {
        float out[2][OG_OUTLEN];      // [history | frame | tail] time-domain buffer
        struct {
            float scratch[176];       // transformed lowband copy
            float tmp[176];           // Hadamard reorder bounce buffer
        } b;
    } v;
    OgCeltConst k;                    // LDS copy of the small constant tables
    uint32_t band_cm[2][24];
    OgTask task[44];
    uint32_t stepmask[4];             // which lock-step rounds / sequential bands the frame needs
    uint8_t bin2band[960];
    float mrg[21][8][2];              // stereo merge partial sums
    alignas(16) float mg[24][4];      // per-band merge matrix
    float gband[2][24];               // denormalisation gains
          // OR of the PVQ leaves' collapse masks per (buffer, band), filled by K1
    OgCeltHdr hdr;                    // the frame's side-info, staged once (uniform control code reads it from LDS)
    OgCeltLeaf leaves[OG_LDS_LEAVES];
    uint8_t collapse[2][24];
    float red[4];
    float carry[2][64];
};

#include "celt_pvq.h"

// ------------------------------------------------------------------------------------------------
// Reductions evaluated in uniform code: every lane gets the same value.  On the GPU each wavefront reduces the
// (<= 176-element) vector on its own (lane-strided partials + xor butterfly), so no LDS scratch and no barrier.
#if defined(__HIP__)
#define OG_BARRIER __syncthreads();
OG_DEV float og_wave_sum(float v)
{
    for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
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
OG_DEV void og_wg_haar1(float *x, int n0, int stride)   // celt/decoder.rs:453-466
{
    const float r = 0.70710678118654752440f;
    const int sh = __builtin_ctz((unsigned)stride), total = (n0 >> 1) * stride;
    OG_LANES_BEGIN(tid)
    for (int p = tid; p < total; p += OG_WG) {
        int i = p & (stride - 1), j = p >> sh;
        float a = x[stride * 2 * j + i], b = x[stride * (2 * j + 1) + i];
        x[stride * 2 * j + i] = (a + b) * r;
        x[stride * (2 * j + 1) + i] = (a - b) * r;
    }
    OG_LANES_END
}

// celt/decoder.rs:472-538.  inverse = 0: deinterleave (frequency order -> block order), 1: interleave.
OG_DEV void og_wg_hadamard_reorder(float *x, float *tmp, int n0, int stride, int hadamard, int inverse, const uint8_t *ordery)
{
    const int sh = __builtin_ctz((unsigned)stride), n = n0 * stride;
    const uint8_t *ord = ordery + stride - 2;
    OG_LANES_BEGIN(tid)
    for (int p = tid; p < n; p += OG_WG) {
        int i = p & (stride - 1), j = p >> sh;
        int row = hadamard ? ord[i] : i;
        if (inverse) tmp[j * stride + i] = x[row * n0 + j];
        else tmp[row * n0 + j] = x[j * stride + i];
    }
    OG_LANES_END
    OG_LANES_BEGIN(tid)
    for (int p = tid; p < n; p += OG_WG) x[p] = tmp[p];
    OG_LANES_END
}

// x *= gain / ||x||  (celt/decoder.rs:788-794)
OG_DEV void og_wg_renormalize(float *x, int n, float gain)
{
    const float g = gain / sqrtf(1e-15f + og_uniform_sumsq(x, n));
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

struct OgBandCtx {
    const OgCeltHdr *h;               // staged in LDS
    const OgCeltLeaf *lv;             // the frame's leaves in HBM (only those beyond OG_LDS_LEAVES are read from there)
    const OgCeltTables *T;
    OgCeltLds *L;
    uint32_t seed;                    // uniform across the workgroup
    int lm, band;
    uint64_t *cycles;                 // profiling stamps (may be null)
};

// The Haar steps of one quant_band call -- tf recombination (celt/decoder.rs:1708-1716, strides 1, 2, 4) and tf
// time-division (:1722-1740, strides B, 2B, ...) -- are butterflies on distinct bits of the coefficient index, so
// together they are a Walsh-Hadamard transform over the bit mask `hm`, scaled by 2^(-m/2).  The Hadamard
// (de)interleave (:472-538) that follows / precedes them is a permutation.  Fused, every output coefficient is one
// gather of 2^m inputs: a whole pre- or post-processing chain costs one barrier instead of one per step.
struct OgXform {
    int hm;                           // index bits the butterflies act on
    int slog, n0p, had;               // reorder: stride 2^slog (0 = none), row length, natural / ordery rows
    float scale;                      // 2^(-popcount(hm)/2)
};

OG_DEV void og_make_xform(int n, int B, int tf_change, OgXform *t, int *recombine_out, int *time_divide_out, int *B0_out)
{
    const int long_blocks = B == 1;
    int n_b = n / B, recombine = tf_change > 0 ? tf_change : 0, time_divide = 0;
    B >>= recombine;
    n_b <<= recombine;
    const int b1 = 31 - __builtin_clz((unsigned)B);
    while ((n_b & 1) == 0 && tf_change < 0) { B <<= 1; n_b >>= 1; time_divide++; tf_change++; }
    t->hm = ((1 << recombine) - 1) | (((1 << time_divide) - 1) << b1);
    t->slog = B > 1 ? (31 - __builtin_clz((unsigned)B)) + recombine : 0;
    t->n0p = n_b >> recombine;
    t->had = long_blocks;
    const int m = recombine + time_divide;
    t->scale = (m & 1 ? 0.70710678118654752440f : 1.f) / (float)(1 << (m >> 1));
    *recombine_out = recombine; *time_divide_out = time_divide; *B0_out = B;
}

// where interleave_hadamard takes the value at position d from (= where deinterleave_hadamard puts the value of d)
OG_DEV int og_reorder(const OgXform &t, int d, const uint8_t *ordery)
{
    if (!t.slog) return d;
    const int S = 1 << t.slog, i = d & (S - 1), j = d >> t.slog;
    return (t.had ? ordery[S - 2 + i] : i) * t.n0p + j;
}

// Walsh-Hadamard gather: scale * sum over s (= p outside hm) of (-1)^popcount(p & s & hm) * src[perm(s)]
OG_DEV float og_wht_gather(const OgXform &t, const float *src, int p, const uint8_t *ordery, bool permute_src)
{
    float acc = 0.f;
    int q = 0;
    do {
        const int s = (p & ~t.hm) | q;
        const float v = src[permute_src ? og_reorder(t, s, ordery) : s];
        acc += (__builtin_popcount((unsigned)(p & q)) & 1) ? -v : v;
        q = (q - t.hm) & t.hm;
    } while (q);
    return acc * t.scale;
}

OG_DEV OgCeltLeaf og_leaf(const OgBandCtx &c, int idx) { return idx < OG_LDS_LEAVES ? c.L->leaves[idx] : c.lv[idx]; }

// One quant_band call (celt/decoder.rs:1665-1768 pre, :1601-1663 leaves, :1994-2032 post) on channel
// buffer x, for the leaves of band `bd` whose OG_L_CH_Y flag equals chy.  PVQ leaves were expanded by K1.
OG_DEV int og_quant_band(OgBandCtx &c, const OgCeltBand &bd, int chy, float *x, int n, int B, const float *lowband,
                         float *lowband_out, int fill)
{
    OgCeltLds &L = *c.L;
    const int g0 = bd.leaf0 + (chy ? bd.nx : 0), nfill = chy ? bd.fy : bd.fx;     // this buffer's leaf group
    if (n == 1) {                                       // celt/decoder.rs:1466-1495
        const float v = og_leaf(c, g0).index ? -1.f : 1.f;
        OG_LANES_BEGIN(tid)
        if (tid == 0) {
            x[0] = v;
            if (lowband_out) lowband_out[0] = v;
        }
        OG_LANES_END
        return 1;
    }
    const int n0 = n;
    OgXform xf;
    int recombine, time_divide, B0;
    og_make_xform(n, B, bd.tf_change, &xf, &recombine, &time_divide, &B0);
    for (int k = 0; k < recombine; k++) fill = L.k.bit_interleave[fill & 0xF] | L.k.bit_interleave[fill >> 4] << 2;
    for (int k = 0, Bk = B >> recombine; k < time_divide; k++, Bk <<= 1) fill |= fill << Bk;
    const bool need_lowband = lowband && (bd.flags & (chy ? OG_B_FOLD_Y : OG_B_FOLD_X));
    float *lb = L.v.b.scratch, *tmp = L.v.b.tmp;
    if (need_lowband) {                                 // Haar steps + deinterleave_hadamard on a copy of the folding source
        OG_LANES_BEGIN(tid)
        for (int p = tid; p < n; p += OG_WG) lb[og_reorder(xf, p, L.k.hadamard)] = og_wht_gather(xf, lowband, p, L.k.hadamard, false);
        OG_LANES_END
    }

    int cm = (int)L.band_cm[chy ? 1 : 0][c.band];       // PVQ leaves: expanded (and masks collected) by K1
    for (int l = 0; l < nfill; l++) {                   // fill leaves in decode order (the LCG seed is sequential)
        const OgCeltLeaf lf = og_leaf(c, g0 + l);
        const int Bl = 1 << (lf.meta & OG_L_LOGB_MASK), sh = lf.cmshift;
        // the leaf's share of the quant_band fill: the recursion only shifts it (side halves) or clears it (theta)
        const int fl = (lf.meta & OG_L_FILL_DEAD) ? 0 : (fill >> sh) & ((1 << Bl) - 1);
        if (!fl) continue;                              // x stays zero (celt/decoder.rs:1640-1643)
        float *xl = x + lf.off;
        const int nn = lf.n;
        const uint32_t s0 = c.seed;
        const uint32_t *la = c.T->lcg_a, *lc = c.T->lcg_c;
        if (!(lf.meta & OG_L_HAS_LOWBAND)) {            // noise (celt/decoder.rs:1644-1651)
            OG_LANES_BEGIN(tid)
            for (int p = tid; p < nn; p += OG_WG) {
                uint32_t s = la[p + 1] * s0 + lc[p + 1];
                xl[p] = (float)((int32_t)s >> 20);
            }
            OG_LANES_END
            cm |= ((1 << Bl) - 1) << sh;
        } else {                                        // fold the lower band in (celt/decoder.rs:1652-1659)
            const float *lbl = lb + lf.off;
            OG_LANES_BEGIN(tid)
            for (int p = tid; p < nn; p += OG_WG) {
                uint32_t s = la[p + 1] * s0 + lc[p + 1];
                xl[p] = lbl[p] + ((s & 0x8000) ? 1.f / 256 : -1.f / 256);
            }
            OG_LANES_END
            cm |= fl << sh;
        }
        c.seed = la[nn] * s0 + lc[nn];
        og_wg_renormalize(xl, nn, lf.gain);
    }

    // interleave_hadamard + the Haar steps that undo time-division and recombination (celt/decoder.rs:1994-2019),
    // fused; lowband_out = sqrt(N0) * x (:2021-2028) rides on the copy-back
    for (int k = 0, Bk = B0; k < time_divide; k++) { Bk >>= 1; cm |= cm >> Bk; }
    for (int k = 0; k < recombine; k++) cm = L.k.bit_deinterleave[cm];
    const int Bend = (B0 >> time_divide) << recombine;
    if (xf.hm || xf.slog) {
        OG_LANES_BEGIN(tid)
        for (int p = tid; p < n; p += OG_WG) tmp[p] = og_wht_gather(xf, x, p, L.k.hadamard, true);
        OG_LANES_END
        const float sq = sqrtf((float)n0);
        OG_LANES_BEGIN(tid)
        for (int p = tid; p < n; p += OG_WG) {
            const float v = tmp[p];
            x[p] = v;
            if (lowband_out) lowband_out[p] = sq * v;
        }
        OG_LANES_END
    } else if (lowband_out) {
        const float sq = sqrtf((float)n0);
        OG_LANES_BEGIN(tid)
        for (int p = tid; p < n0; p += OG_WG) lowband_out[p] = sq * x[p];
        OG_LANES_END
    }
    return cm & ((1 << Bend) - 1);
}

// celt/decoder.rs:2037-2220 with the range-decoder half removed (the host ran it).
//
// The reference walks the bands one after the other, but only two things are truly sequential: folding reads the
// finished lower bands (`norm`), and the noise seed runs through the fill leaves in decode order.  Everything a
// quant_band call does to a band whose leaves are all PVQ-coded -- Hadamard interleave, the Haar steps that undo
// tf time-division / recombination, lowband_out -- depends on that band alone.  So:
//   pass A  all such (band, buffer) tasks advance in lock-step, one barrier per step for the whole frame;
//   pass B  the few tasks with live fill leaves run in band order through og_quant_band (as the reference does);
//   pass C  the mid/side -> left/right merges of all joint bands, in lock-step (norm keeps the mid, so nothing
//           downstream of a band needs its merged form).
OG_DEV void og_celt_bands(OgBandCtx &c)
{
    const OgCeltHdr *h = c.h;
    OgCeltLds &L = *c.L;
    const int lm = c.lm, M = 1 << lm, start = h->start, end = h->end;
    const int C = (h->flags & OG_F_STEREO) ? 2 : 1;
    const int B = (h->flags & OG_F_TRANSIENT) ? M : 1;
    const int norm_offset = M * L.k.ebands[start];
    float *tmp_all = &L.v.out[1][0];                    // [2][960] bounce buffer for the lock-step Hadamard reorder

    // ---- set-up: one lane per (band, buffer) task; another 22 lanes build the bin -> band map
    OG_LANES_BEGIN(tid)
    for (int t = tid; t < 42; t += OG_WG) {
        const int i = t >> 1, slot = t & 1;
        OgTask tk;
        tk.boff = tk.n = tk.n0p = tk.cm = 0;
        tk.buf = (uint8_t)slot; tk.flags = 0; tk.hm = tk.slog = tk.m = 0; tk.norm_row = 0xFF; tk.pad0 = tk.pad1 = 0;
        if (i >= start && i < end) {
            const OgCeltBand bd = h->band[i];
            const int n = M * L.k.width[i], boff = M * L.k.ebands[i];
            const bool dual = (bd.flags & OG_B_DUAL) != 0;
            const int nfill = slot ? bd.fy : bd.fx;
            bool active, chain = false;
            int norm_row = -1;
            if (C == 1) { active = slot == 0; norm_row = 0; chain = nfill > 0; }
            else if (dual) { active = true; norm_row = slot; chain = nfill > 0; }
            else if (n == 1) { active = true; norm_row = slot == 0 ? 0 : -1; }
            else if (n == 2) { active = slot == ((bd.flags & OG_B_N2SWAP) ? 1 : 0); norm_row = 0; chain = nfill > 0; }
            else { active = true; norm_row = slot == 0 ? 0 : -1; chain = slot == 0 && nfill > 0 && !(bd.flags & OG_B_MIDFILL0); }
            if (bd.flags & OG_B_LAST) norm_row = -1;
            if (active) {
                tk.boff = (uint16_t)boff; tk.n = (uint16_t)n;
                tk.flags = OG_T_ACTIVE | (chain ? OG_T_CHAIN : 0) | (B == 1 ? OG_T_LONG : 0);
                tk.norm_row = (uint8_t)norm_row;
                if (n == 1) {                           // decode_band_1 (celt/decoder.rs:1466-1495)
                    const OgCeltLeaf lf = og_leaf(c, bd.leaf0 + (slot ? bd.nx : 0));
                    L.X[slot][boff] = lf.index ? -1.f : 1.f;
                    tk.cm = 1;
                } else {
                    OgXform xf;
                    int recombine, time_divide, B0;
                    og_make_xform(n, B, bd.tf_change, &xf, &recombine, &time_divide, &B0);
                    tk.hm = (uint8_t)xf.hm; tk.slog = (uint8_t)xf.slog; tk.n0p = (uint16_t)xf.n0p;
                    tk.m = (uint8_t)(recombine + time_divide);
                    if (!chain) {
                        int cm = (int)L.band_cm[slot][i];       // fill leaves of such a task are dead: the PVQ masks are all
                        for (int k = 0, Bk = B0; k < time_divide; k++) { Bk >>= 1; cm |= cm >> Bk; }
                        for (int k = 0; k < recombine; k++) cm = L.k.bit_deinterleave[cm];
                        const int Bend = (B0 >> time_divide) << recombine;
                        tk.cm = (uint16_t)(cm & ((1 << Bend) - 1));
                        if (xf.hm || xf.slog) OG_ATOMIC_OR(&L.stepmask[0], 1u);
                    }
                }
                if (chain || (bd.flags & OG_B_MERGE_NORM)) OG_ATOMIC_OR(&L.stepmask[3], 1u << i);
            } else if (bd.flags & OG_B_MERGE_NORM) {
                OG_ATOMIC_OR(&L.stepmask[3], 1u << i);
            }
        }
        L.task[t] = tk;
    }
    if (tid >= 64 && tid < 64 + 22) {
        const int b = tid - 64;
        const int lo = b < 21 ? M * L.k.ebands[b] : M * 100, hi = b < 21 ? M * L.k.ebands[b + 1] : 960;
        for (int j = lo; j < hi; j++) L.bin2band[j] = (uint8_t)(b < 21 ? b : 255);
    }
    OG_LANES_END

#define OG_FOR_TASK_ELEMS(e, tk, p)                                                                                  \
    for (int e = tid; e < 1920; e += OG_WG) {                                                                        \
        const int buf = e >= 960, bin = e - 960 * buf;                                                               \
        const int band_ = L.bin2band[bin];                                                                           \
        if (band_ == 255) continue;                                                                                  \
        const OgTask tk = L.task[band_ * 2 + buf];                                                                   \
        if ((tk.flags & (OG_T_ACTIVE | OG_T_CHAIN)) != OG_T_ACTIVE) continue;                                        \
        const int p = bin - tk.boff;

    OG_STAMP_TO(c.cycles, 16)
    // ---- pass A: fused post-processing (see OgXform) of every PVQ-only task, then copy-back + lowband_out
    const uint32_t m_xf = L.stepmask[0], m_seq = L.stepmask[3];
    if (m_xf) {
        OG_LANES_BEGIN(tid)
        OG_FOR_TASK_ELEMS(e, tk, p)
            if (!(tk.hm | tk.slog)) continue;
            OgXform xf;
            xf.hm = tk.hm; xf.slog = tk.slog; xf.n0p = tk.n0p; xf.had = (tk.flags & OG_T_LONG) != 0;
            xf.scale = (tk.m & 1 ? 0.70710678118654752440f : 1.f) / (float)(1 << (tk.m >> 1));
            tmp_all[e] = og_wht_gather(xf, &L.X[buf][tk.boff], p, L.k.hadamard, true);
        }
        OG_LANES_END
    }
    OG_LANES_BEGIN(tid)                                 // lowband_out = sqrt(N0) * x (celt/decoder.rs:2021-2028)
    OG_FOR_TASK_ELEMS(e, tk, p)
        (void)p;
        float v = L.X[buf][bin];
        if (tk.hm | tk.slog) { v = tmp_all[e]; L.X[buf][bin] = v; }
        if (tk.norm_row != 0xFF) L.u.norm[tk.norm_row][bin - norm_offset] = sqrtf((float)tk.n) * v;
    }
    for (int i = tid; i < 21; i += OG_WG) {             // collapse masks of the bands pass B will not visit
        if (i < start || i >= end || ((m_seq >> i) & 1)) continue;
        const int cm0 = L.task[2 * i].cm, cm1 = L.task[2 * i + 1].cm;
        const bool dual = (h->band[i].flags & OG_B_DUAL) != 0;
        L.collapse[0][i] = (uint8_t)(dual || C == 1 ? cm0 : (cm0 | cm1));
        L.collapse[C - 1][i] = (uint8_t)(dual ? cm1 : C == 1 ? cm0 : (cm0 | cm1));
    }
    OG_LANES_END

    OG_STAMP_TO(c.cycles, 17)
    // ---- pass B: bands with live fill leaves (or the dual -> intensity switch), in order
    for (int i = start; i < end; i++) {
        if (!((m_seq >> i) & 1)) continue;
        const OgCeltBand bd = h->band[i];
        c.band = i;
        const int boff = M * L.k.ebands[i], n = M * L.k.width[i];
        const bool last = (bd.flags & OG_B_LAST) != 0;
        const OgTask t0 = L.task[2 * i], t1 = L.task[2 * i + 1];
        int x_cm, y_cm;
        if (bd.lowband != 0xFF) {
            x_cm = y_cm = 0;
            for (int f = bd.fold_lo; f < bd.fold_hi; f++) { x_cm |= L.collapse[0][f]; y_cm |= L.collapse[C - 1][f]; }
        } else {
            x_cm = y_cm = (1 << B) - 1;
        }
        if (bd.flags & OG_B_MERGE_NORM) {               // dual -> intensity switch (celt/decoder.rs:2133-2138)
            const int cnt = boff - norm_offset;
            OG_LANES_BEGIN(tid)
            for (int p = tid; p < cnt; p += OG_WG) L.u.norm[0][p] = 0.5f * (L.u.norm[0][p] + L.u.norm[1][p]);
            OG_LANES_END
        }
        const float *lb0 = bd.lowband != 0xFF ? L.u.norm[0] + bd.lowband * M : nullptr;
        const float *lb1 = bd.lowband != 0xFF ? L.u.norm[1] + bd.lowband * M : nullptr;
        float *out0 = last ? nullptr : L.u.norm[0] + boff - norm_offset;
        float *out1 = last ? nullptr : L.u.norm[1] + boff - norm_offset;
        int cm0 = t0.cm, cm1 = t1.cm;
        const bool dual = (bd.flags & OG_B_DUAL) != 0;
        const int fill0 = dual ? x_cm : (x_cm | y_cm), fill1 = dual ? y_cm : (x_cm | y_cm);
        if (t0.flags & OG_T_CHAIN) cm0 = og_quant_band(c, bd, 0, L.X[0] + boff, n, B, lb0, out0, fill0);
        if (t1.flags & OG_T_CHAIN) cm1 = og_quant_band(c, bd, 1, L.X[1] + boff, n, B, dual ? lb1 : lb0, dual ? out1 : out0, fill1);
        L.collapse[0][i] = (uint8_t)(dual || C == 1 ? cm0 : (cm0 | cm1));
        L.collapse[C - 1][i] = (uint8_t)(dual ? cm1 : C == 1 ? cm0 : (cm0 | cm1));
    }
    OG_BARRIER

    OG_STAMP_TO(c.cycles, 18)
    // ---- pass C: stereo merges (celt/decoder.rs:796-827 + the N == 2 case :1826-1862), all joint bands at once
    if (C == 2) {
        OG_LANES_BEGIN(tid)
        if (tid < 21 * 8) {
            const int i = tid >> 3, sub = tid & 7;
            const int n = M * L.k.width[i];
            float xy = 0.f, yy = 0.f;
            if (i >= start && i < end && (h->band[i].flags & OG_B_JOINT) && n > 2) {
                const float *x = L.X[0] + M * L.k.ebands[i], *y = L.X[1] + M * L.k.ebands[i];
                for (int p = sub; p < n; p += 8) { xy += x[p] * y[p]; yy += y[p] * y[p]; }
            }
            L.mrg[i][sub][0] = xy;
            L.mrg[i][sub][1] = yy;
        }
        OG_LANES_END
        OG_LANES_BEGIN(tid)
        if (tid < 21) {                                 // per band: x' = ax x + bx y, y' = ay x + by y
            const int i = tid;
            const OgCeltBand bd = h->band[i];
            const int n = M * L.k.width[i];
            float ax = 1.f, bx = 0.f, ay = 0.f, by = 1.f;
            if (i >= start && i < end && (bd.flags & OG_B_JOINT)) {
                const float mid = (1.f / 32768) * bd.imid, side = (1.f / 32768) * bd.iside;
                const float inv = (bd.flags & OG_B_INV) ? -1.f : 1.f;
                if (n == 2) {
                    float *x = L.X[0] + M * L.k.ebands[i], *y = L.X[1] + M * L.k.ebands[i];
                    const int sw = (bd.flags & OG_B_N2SWAP) != 0;
                    float *x2 = sw ? y : x, *y2 = sw ? x : y;
                    const float sign = (bd.flags & OG_B_N2NEG) ? -1.f : 1.f;
                    y2[0] = -sign * x2[1];
                    y2[1] = sign * x2[0];
                    x[0] *= mid; x[1] *= mid; y[0] *= side; y[1] *= side;
                    float t = x[0]; x[0] = t - y[0]; y[0] = inv * (t + y[0]);
                    t = x[1]; x[1] = t - y[1]; y[1] = inv * (t + y[1]);
                } else {
                    float xp = 0.f, sd = 0.f;
                    for (int q = 0; q < 8; q++) { xp += L.mrg[i][q][0]; sd += L.mrg[i][q][1]; }
                    xp *= mid;
                    const float e = mid * mid + sd, e0 = e - 2.f * xp, e1 = e + 2.f * xp;
                    if (e0 < 6e-4f || e1 < 6e-4f) {     // degenerate: y = x
                        ay = inv; by = 0.f;
                    } else {
                        const float g0 = 1.f / sqrtf(e0), g1 = 1.f / sqrtf(e1);
                        ax = g0 * mid; bx = -g0; ay = inv * g1 * mid; by = inv * g1;
                    }
                }
            }
            L.mg[i][0] = ax; L.mg[i][1] = bx; L.mg[i][2] = ay; L.mg[i][3] = by;
        }
        OG_LANES_END
        OG_LANES_BEGIN(tid)
        for (int bin = tid; bin < 960; bin += OG_WG) {
            const int i = L.bin2band[bin];
            if (i == 255) continue;
            const OgF4 m4 = *(const OgF4 *)L.mg[i];
            const float x = L.X[0][bin], y = L.X[1][bin];
            L.X[0][bin] = m4.x * x + m4.y * y;
            L.X[1][bin] = m4.z * x + m4.w * y;
        }
        OG_LANES_END
    }
#undef OG_FOR_TASK_ELEMS
    OG_STAMP_TO(c.cycles, 19)
}

// RFC 6716 4.3.5 anti-collapse (bands.c); the reference only reserves its flag bit (celt/decoder.rs:1092)
OG_DEV void og_anti_collapse(OgBandCtx &c)
{
    const OgCeltHdr *h = c.h;
    OgCeltLds &L = *c.L;
    const int lm = c.lm, C = (h->flags & OG_F_STEREO) ? 2 : 1;
    const uint32_t *la = c.T->lcg_a, *lc = c.T->lcg_c;
    for (int i = h->start; i < h->end; i++) {
        const int n0 = L.k.width[i];
        const int depth = h->pulses_depth[i];
        const float thresh = 0.5f * exp2f(-0.125f * (float)depth);
        const float sqrt_1 = 1.f / sqrtf((float)(n0 << lm));
        for (int ch = 0; ch < C; ch++) {
            float ediff = h->energy[ch][i] - h->prev_min[ch][i];
            if (ediff < 0) ediff = 0;
            float r = 2.f * exp2f(-ediff);
            if (lm == 3) r *= 1.41421356f;
            if (r > thresh) r = thresh;
            r *= sqrt_1;
            float *x = L.X[ch] + (L.k.ebands[i] << lm);
            const int mask = L.collapse[ch][i];
            bool renorm = false;
            for (int k = 0; k < (1 << lm); k++) {
                if (mask & (1 << k)) continue;
                const uint32_t s0 = c.seed;
                OG_LANES_BEGIN(tid)
                for (int j = tid; j < n0; j += OG_WG) {
                    uint32_t s = la[j + 1] * s0 + lc[j + 1];
                    x[(j << lm) + k] = (s & 0x8000) ? r : -r;
                }
                OG_LANES_END
                c.seed = la[n0] * s0 + lc[n0];
                renorm = true;
            }
            if (renorm) og_wg_renormalize(x, n0 << lm, 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: IMDCT as an N/4-point complex FFT (N/4 = 60 * 2^lm = 15 * 2^(lm+2): radix 4/2/3/5 Stockham
// stages ping-ponging between two LDS buffers), RFC 6716 4.3.7.
OG_DEV OgCf og_cmul(OgCf a, OgCf b) { OgCf r; r.r = a.r * b.r - a.i * b.i; r.i = a.r * b.i + a.i * b.r; return r; }

// floor(q / d) for the small non-negative integers used as work-item indices (q < 4096, d <= 480): exact
OG_DEV int og_div(int q, float inv_d) { return (int)(((float)q + 0.5f) * inv_d); }

// One Stockham stage of radix R over nch*blocks transforms of size n; transform b of channel ch starts at
// ch*480 + b*n.  Ns = product of the radices already done: a power of two for R = 2, 3, 4 (stage order 4,4,2,3,5)
// and n/5 for the final R = 5 stage, so the position inside the sub-transform never needs a division.
template <int R>
OG_DEV void og_wg_fft_stage(const OgCf *src, OgCf *dst, int n, int nch, int blocks, int Ns, const OgCf *W)
{
    const int per = n / R, total = nch * blocks * per, wstep = 480 / (Ns * R);
    const float inv_per = 1.f / (float)per;
    const int bsh = __builtin_ctz((unsigned)blocks);
    OG_LANES_BEGIN(tid)
    for (int q = tid; q < total; q += OG_WG) {
        const int tt = og_div(q, inv_per), j = q - tt * per;
        const int ch = tt >> bsh, b = tt & (blocks - 1);
        const int base = ch * 480 + b * n;
        const int k = R == 5 ? j : (j & (Ns - 1));
        const int j0 = (j - k) * R + k;
        OgCf v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            v[r] = src[base + j + r * per];
            if (r) v[r] = og_cmul(v[r], W[k * r * wstep]);
        }
        OgCf *o = dst + base + j0;
        if (R == 2) {
            o[0].r = v[0].r + v[1].r; o[0].i = v[0].i + v[1].i;
            o[Ns].r = v[0].r - v[1].r; o[Ns].i = v[0].i - v[1].i;
        } else if (R == 4) {
            OgCf t0 = {v[0].r + v[2].r, v[0].i + v[2].i}, t1 = {v[0].r - v[2].r, v[0].i - v[2].i};
            OgCf t2 = {v[1].r + v[3 % R].r, v[1].i + v[3 % R].i}, t3 = {v[1].i - v[3 % R].i, v[3 % R].r - v[1].r};   // -i (v1 - v3)
            o[0].r = t0.r + t2.r; o[0].i = t0.i + t2.i;
            o[Ns].r = t1.r + t3.r; o[Ns].i = t1.i + t3.i;
            o[2 * Ns].r = t0.r - t2.r; o[2 * Ns].i = t0.i - t2.i;
            o[3 * Ns].r = t1.r - t3.r; o[3 * Ns].i = t1.i - t3.i;
        } else if (R == 3) {
            const float s3 = 0.86602540378443864676f;
            OgCf s = {v[1].r + v[2 % R].r, v[1].i + v[2 % R].i}, d = {v[1].r - v[2 % R].r, v[1].i - v[2 % R].i};
            OgCf m = {v[0].r - 0.5f * s.r, v[0].i - 0.5f * s.i};
            OgCf e = {s3 * d.i, -s3 * d.r};                                                          // -i s3 d
            o[0].r = v[0].r + s.r; o[0].i = v[0].i + s.i;
            o[Ns].r = m.r + e.r; o[Ns].i = m.i + e.i;
            o[2 * Ns].r = m.r - e.r; o[2 * Ns].i = m.i - e.i;
        } else {
            const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
            const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
            OgCf s14 = {v[1].r + v[4 % R].r, v[1].i + v[4 % R].i}, d14 = {v[1].r - v[4 % R].r, v[1].i - v[4 % R].i};
            OgCf s23 = {v[2 % R].r + v[3 % R].r, v[2 % R].i + v[3 % R].i}, d23 = {v[2 % R].r - v[3 % R].r, v[2 % R].i - v[3 % R].i};
            OgCf p1 = {v[0].r + c1 * s14.r + c2 * s23.r, v[0].i + c1 * s14.i + c2 * s23.i};
            OgCf p2 = {v[0].r + c2 * s14.r + c1 * s23.r, v[0].i + c2 * s14.i + c1 * s23.i};
            OgCf q1 = {s1 * d14.r + s2 * d23.r, s1 * d14.i + s2 * d23.i};
            OgCf q2 = {s2 * d14.r - s1 * d23.r, s2 * d14.i - s1 * d23.i};
            o[0].r = v[0].r + s14.r + s23.r; o[0].i = v[0].i + s14.i + s23.i;
            o[Ns].r = p1.r + q1.i; o[Ns].i = p1.i - q1.r;                                           // p1 - i q1
            o[4 * Ns].r = p1.r - q1.i; o[4 * Ns].i = p1.i + q1.r;
            o[2 * Ns].r = p2.r + q2.i; o[2 * Ns].i = p2.i - q2.r;
            o[3 * Ns].r = p2.r - q2.i; o[3 * Ns].i = p2.i + q2.r;
        }
    }
    OG_LANES_END
}

// RFC 6716 celt.c comb_filter, in place, for both channels; y points at sample 0 of channel 0 inside
// L.v.out, channel stride OG_OUTLEN.  The recursion reaches back >= 13 samples, so `step` samples are
// independent at a time.
OG_DEV void og_wg_comb_filter(float *y0, int nch, int T0, int T1, int n, float g0, float g1, int tap0, int tap1,
                              const float *window, const float (*taps)[4])
{
    if (g0 == 0.f && g1 == 0.f) return;
    if (T0 < 15) T0 = 15;
    if (T1 < 15) T1 = 15;
    const float g00 = g0 * taps[tap0][0], g01 = g0 * taps[tap0][1], g02 = g0 * taps[tap0][2];
    const float g10 = g1 * taps[tap1][0], g11 = g1 * taps[tap1][1], g12 = g1 * taps[tap1][2];
    const int overlap = (g0 == g1 && T0 == T1 && tap0 == tap1) ? 0 : 120;
    const int stop = g1 == 0.f ? overlap : n;          // with g1 == 0 only the cross-fade runs
    const int lanes = OG_WG / 2;                       // lanes per channel
    int step = (T0 < T1 ? T0 : T1) - 2;
    if (step > lanes) step = lanes;
    for (int i0 = 0; i0 < stop; i0 += step) {
        const int i1 = i0 + step < stop ? i0 + step : stop;
        OG_LANES_BEGIN(tid)
        const int ch = tid / lanes, i = i0 + (tid - ch * lanes);
        if (ch < nch && i < i1) {
            float *y = y0 + ch * OG_OUTLEN;
            if (i < overlap) {
                const float f = window[i] * window[i];
                y[i] = y[i] + ((1.f - f) * g00) * y[i - T0]
                     + ((1.f - f) * g01) * (y[i - T0 + 1] + y[i - T0 - 1])
                     + ((1.f - f) * g02) * (y[i - T0 + 2] + y[i - T0 - 2])
                     + (f * g10) * y[i - T1]
                     + (f * g11) * (y[i - T1 + 1] + y[i - T1 - 1])
                     + (f * g12) * (y[i - T1 + 2] + y[i - T1 - 2]);
            } else {
                y[i] = y[i] + g10 * y[i - T1] + g11 * (y[i - T1 + 1] + y[i - T1 - 1]) + g12 * (y[i - T1 + 2] + y[i - T1 - 2]);
            }
        }
        OG_LANES_END
    }
}

// ------------------------------------------------------------------------------------------------
// The whole frame.  Uniform arguments; every lane of the workgroup calls it.
// `dbg` pointers are already offset to this frame (any may be null).
OG_DEV void og_celt_frame(const OgCeltHdr *hg, const OgCeltLeaf *leaves_base, const OgCeltTables *T, OgCeltState *states,
                          float *pcm_base, size_t pcm_stride, OgCeltLds &L, const OgCeltDebug &dbg, float *coef_row,
                          uint32_t *cm_row)
{
    int16_t *dbg_pulses = dbg.pulses;
    float *dbg_norm = dbg.norm, *dbg_freq = dbg.freq;
    uint8_t *dbg_collapse = dbg.collapse;
    OG_STAMP(0)
    const OgCeltLeaf *lv = leaves_base + hg->leaf_begin;
    const int nleaves = hg->nleaves < OG_MAXLEAF ? hg->nleaves : OG_MAXLEAF;

    // ---- phase 0: stage the side-info and the PVQ table in LDS, clear the coefficient buffers
    OG_LANES_BEGIN(tid)
    {
        const uint32_t *src = (const uint32_t *)hg;
        uint32_t *dst = (uint32_t *)&L.hdr;
        for (int p = tid; p < (int)(sizeof(OgCeltHdr) / 4); p += OG_WG) dst[p] = src[p];
        const uint32_t *ls = (const uint32_t *)lv;      // leaves are 4 words
        uint32_t *ld = (uint32_t *)L.leaves;
        const int nst = 4 * (nleaves < OG_LDS_LEAVES ? nleaves : OG_LDS_LEAVES);
        for (int p = tid; p < nst; p += OG_WG) ld[p] = ls[p];
    }
    for (int p = tid; p < (int)(sizeof(OgCeltConst) / 4); p += OG_WG) ((uint32_t *)&L.k)[p] = ((const uint32_t *)&T->k)[p];
    // K1's output for this frame: shaped PVQ coefficients (zero elsewhere) and the collapse masks of the PVQ leaves.
    // Both scratch rows are handed back zero-filled for the next launch.
    for (int p = tid; p < 2 * 960 / 4; p += OG_WG) {    // 16 bytes per lane
        ((OgF4 *)&L.X[0][0])[p] = ((const OgF4 *)coef_row)[p];
        ((OgF4 *)coef_row)[p] = OgF4{0.f, 0.f, 0.f, 0.f};
    }
    for (int p = tid; p < 2 * 800; p += OG_WG) (&L.u.norm[0][0])[p] = 0.f;
    if (tid < 48) { (&L.collapse[0][0])[tid] = 0; (&L.band_cm[0][0])[tid] = cm_row[tid]; cm_row[tid] = 0; }
    if (tid < 4) L.stepmask[tid] = 0;
    OG_LANES_END
    const OgCeltHdr *h = &L.hdr;
    const int lm = h->lm, M = 1 << lm, N = 120 << lm;
    const int C = (h->flags & OG_F_STEREO) ? 2 : 1, CC = (h->flags & OG_F_OUT_STEREO) ? 2 : 1;
    const bool transient = (h->flags & OG_F_TRANSIENT) != 0, silence = (h->flags & OG_F_SILENCE) != 0;
    OgCeltState *st = states + h->stream;
    const bool reset = (h->flags & OG_F_RESET) != 0;

    OG_STAMP(1)
    if (!silence) {
        OG_STAMP(2)
        // ---- phase 2 (K2): bands in order
        OgBandCtx c;
        c.h = h; c.lv = lv; c.T = T; c.L = &L; c.seed = h->seed; c.lm = lm; c.cycles = dbg.cycles;
        og_celt_bands(c);
        if (dbg_norm || dbg_collapse) {
            OG_LANES_BEGIN(tid)
            if (dbg_norm) for (int p = tid; p < 2 * 960; p += OG_WG) dbg_norm[p] = (&L.X[0][0])[p];
            if (dbg_collapse && tid < 42) dbg_collapse[tid] = L.collapse[tid / 21][tid % 21];
            OG_LANES_END
        }
        OG_STAMP(3)
        if (h->flags & OG_F_ANTICOLLAPSE) og_anti_collapse(c);
    }
    OG_STAMP(4)

    // ---- phase 4: denormalise (RFC 6716 4.3.6) + channel up/down-mix into MDCT bins; the same two regions stage the
    //      FFT twiddles and this block size's IMDCT twiddles into the (currently idle) history part of the time buffer
    const int B = transient ? M : 1, NB = N / B, n4 = NB >> 1;
    OgCf *ldsW = (OgCf *)&L.v.out[0][0], *ldsTw = (OgCf *)&L.v.out[1][0];
    OG_LANES_BEGIN(tid)
    for (int t = tid; t < 42; t += OG_WG) {             // per-(channel, band) gains 2^(E + mean), zero outside [start, end)
        const int ch = t >= 21, b = t - 21 * ch;
        float g = 0.f;
        if (!silence && b >= h->start && b < h->end) {
            const float lg = h->energy[ch < C ? ch : 0][b] + L.k.emeans[b];
            g = exp2f(lg < 32.f ? lg : 32.f);
        }
        L.gband[ch][b] = g;
    }
    if (silence)                                        // the band walk (and its bin -> band map) was skipped
        for (int j = tid; j < 960; j += OG_WG) L.bin2band[j] = 255;
    for (int p = tid; p < 960; p += OG_WG) ((float *)ldsW)[p] = ((const float *)T->fft_w)[p];
    for (int p = tid; p < 2 * n4; p += OG_WG) ((float *)ldsTw)[p] = ((const float *)(T->tw + (n4 - 60)))[p];
    OG_LANES_END
    OG_LANES_BEGIN(tid)
    for (int j = tid; j < 960; j += OG_WG) {
        const int b = L.bin2band[j];
        float v0 = 0.f, v1 = 0.f;
        if (b != 255) {
            v0 = L.X[0][j] * L.gband[0][b];
            v1 = C == 2 ? L.X[1][j] * L.gband[1][b] : v0;
            if (CC == 1 && C == 2) v0 = 0.5f * v0 + 0.5f * v1;
        }
        L.X[0][j] = v0;
        L.X[1][j] = v1;
    }
    OG_LANES_END
    if (dbg_freq) {
        OG_LANES_BEGIN(tid)
        for (int p = tid; p < 2 * 960; p += OG_WG) dbg_freq[p] = (&L.X[0][0])[p];
        OG_LANES_END
    }

    OG_STAMP(5)
    // ---- phase 5 (K3): IMDCT.  B blocks of NB bins, interleaved with stride B in X.
    const OgCf *tw = ldsTw;
    const float inv_n4 = 1.f / (float)n4;
    OgCf *bufA = &L.u.A[0][0], *bufB = (OgCf *)&L.X[0][0];
    OG_LANES_BEGIN(tid)
    for (int q = tid; q < CC * (N >> 1); q += OG_WG) {                       // pre-twiddle
        const int ch = q >= (N >> 1), r = q - ch * (N >> 1), b = og_div(r, inv_n4), i = r - b * n4;
        const float x1 = L.X[ch][(2 * i) * B + b], x2 = L.X[ch][(NB - 1 - 2 * i) * B + b];
        const OgCf w = tw[i];
        OgCf z;
        z.r = x1 * w.r + x2 * w.i;
        z.i = x2 * w.r - x1 * w.i;
        bufA[ch * 480 + b * n4 + i] = z;
    }
    OG_LANES_END
    {
        int radix[5], ns = 0;                           // 60 * 2^lm = 4 * 4^a * 2^b * 3 * 5
        int rem = n4 / 15;
        while (rem % 4 == 0) { radix[ns++] = 4; rem /= 4; }
        if (rem == 2) radix[ns++] = 2;
        radix[ns++] = 3;
        radix[ns++] = 5;
        int Ns = 1;
        OgCf *src = bufA, *dst = bufB;
        for (int s = 0; s < ns; s++) {
            switch (radix[s]) {
            case 4: og_wg_fft_stage<4>(src, dst, n4, CC, B, Ns, ldsW); break;
            case 2: og_wg_fft_stage<2>(src, dst, n4, CC, B, Ns, ldsW); break;
            case 3: og_wg_fft_stage<3>(src, dst, n4, CC, B, Ns, ldsW); break;
            default: og_wg_fft_stage<5>(src, dst, n4, CC, B, Ns, ldsW); break;
            }
            Ns *= radix[s];
            OgCf *t = src; src = dst; dst = t;
        }
        // `src` now holds the spectrum; load state + post-twiddle into the time buffer
        OG_LANES_BEGIN(tid)
        for (int q = tid; q < CC * 60; q += OG_WG) {                         // previous frame's tail
            const int ch = q / 60, i = q - ch * 60;
            L.v.out[ch][OG_HIST + i] = reset ? 0.f : st->overlap[ch][i];
        }
        for (int q = tid; q < CC * (N >> 1); q += OG_WG) {
            const int ch = q >= (N >> 1), r = q - ch * (N >> 1), b = og_div(r, inv_n4), k = r - b * n4;
            const OgCf Z = src[ch * 480 + b * n4 + k], w = tw[k];
            const float vr = Z.r * w.r + Z.i * w.i, vi = Z.i * w.r - Z.r * w.i;
            float *D = &L.v.out[ch][OG_HIST + NB * b + 60];
            D[2 * k] = vi;
            D[NB - 1 - 2 * k] = -vr;
        }
        OG_LANES_END
    }
    OG_LANES_BEGIN(tid)
    for (int q = tid; q < CC * B * 60; q += OG_WG) {                         // TDAC mirror (RFC 6716 mdct.c)
        const int ch = q >= B * 60, r = q - ch * (B * 60), b = og_div(r, 1.f / 60.f), i = r - b * 60;
        float *o = &L.v.out[ch][OG_HIST + NB * b];
        const float x1 = o[119 - i], x2 = o[i];
        const float w1 = T->window[i], w2 = T->window[119 - i];
        o[i] = w2 * x2 - w1 * x1;
        o[119 - i] = w1 * x2 + w2 * x1;
    }
    OG_LANES_END

    OG_STAMP(6)
    // ---- phase 6: comb post-filter (RFC 6716 4.3.7.1); parameter sets [0] old, [1] current, [2] new
    const float pg0 = 0.09375f * h->pf_gain_q[0], pg1 = 0.09375f * h->pf_gain_q[1], pg2 = 0.09375f * h->pf_gain_q[2];
    const uint32_t pos0 = reset ? 0u : st->pos;
    if (pg0 != 0.f || pg1 != 0.f || (lm != 0 && pg2 != 0.f)) {
        int reach = 0;
        for (int k = 0; k < 3; k++) { int t = h->pf_period[k] < 15 ? 15 : h->pf_period[k]; if (t > reach) reach = t; }
        reach += 2;
        if (reach > OG_HIST) reach = OG_HIST;
        OG_LANES_BEGIN(tid)
        for (int q = tid; q < CC * reach; q += OG_WG) {
            const int ch = q / reach, d = q - ch * reach + 1;
            L.v.out[ch][OG_HIST - d] = reset ? 0.f : st->hist[ch][(pos0 - (uint32_t)d) & 1023u];
        }
        OG_LANES_END
        float *y0 = &L.v.out[0][OG_HIST];
        og_wg_comb_filter(y0, CC, h->pf_period[0], h->pf_period[1], 120, pg0, pg1, h->pf_tapset[0], h->pf_tapset[1], T->window, L.k.pf_taps);
        if (lm != 0)
            og_wg_comb_filter(y0 + 120, CC, h->pf_period[1], h->pf_period[2], N - 120, pg1, pg2, h->pf_tapset[1],
                              h->pf_tapset[2], T->window, L.k.pf_taps);
    }

    OG_STAMP(7)
    // ---- phase 7: de-emphasis t[j] = y[j] + 0.85 t[j-1] as a blocked scan (60 lanes x N/60 samples per channel),
    //      then coalesced PCM + state write-back
    const float coef = 0.85000610f;
    const int Lc = N / 60;
    float *tbuf = &L.X[0][0];                           // [2][960], free since the FFT finished... (bufB may hold
                                                        // the spectrum when the stage count is odd; it was consumed above)
    OG_LANES_BEGIN(tid)
    if (tid < CC * 60) {
        const int ch = tid >= 60, l = tid - ch * 60;
        const float *y = &L.v.out[ch][OG_HIST + l * Lc];
        float m = 0.f, t = 0.f;
        for (int j = 0; j < Lc; j++) { t = y[j] + m; m = coef * t; }
        L.carry[ch][l] = m;                             // chunk's own contribution to the running memory
    }
    OG_LANES_END
    float aL = 1.f;
    for (int j = 0; j < Lc; j++) aL *= coef;            // coef^Lc
    // carry-in of chunk l = sum over earlier chunks q of carry[q] * coef^(Lc*(l-1-q)) (+ the stream's memory); terms
    // more than `reach` chunks back are below 1e-9 of their chunk's value and are dropped
    const int reach = 128 / Lc + 1;
    OG_LANES_BEGIN(tid)
    if (tid < CC * 60) {
        const int ch = tid >= 60, l = tid - ch * 60;
        const int q0 = l > reach ? l - reach : 0;
        float m = (reset || q0 > 0) ? 0.f : st->deemph[ch];
        for (int q = q0; q < l; q++) m = m * aL + L.carry[ch][q];
        const float *y = &L.v.out[ch][OG_HIST + l * Lc];
        float *t = tbuf + ch * 960 + l * Lc;
        for (int j = 0; j < Lc; j++) { const float v = y[j] + m; m = coef * v; t[j] = v; }
        if (l == 59) st->deemph[ch] = m;
    }
    OG_LANES_END
    float *pcm = pcm_base + (size_t)h->stream * pcm_stride + (size_t)h->pcm_off * CC;
    const bool accum = (h->flags & OG_F_ACCUM) != 0;
    const bool vec = (((uintptr_t)pcm) & 15) == 0;      // 16-byte stores when the caller's row allows it
    OG_LANES_BEGIN(tid)
    if (vec) {
        const float k = 1.f / 32768.f;
        for (int q = tid; q < N * CC / 4; q += OG_WG) {
            OgF4 v;
            if (CC == 2) {                              // two stereo samples
                const int j = 2 * q;
                v = OgF4{tbuf[j] * k, tbuf[960 + j] * k, tbuf[j + 1] * k, tbuf[960 + j + 1] * k};
            } else {
                const OgF4 u = ((const OgF4 *)tbuf)[q];
                v = OgF4{u.x * k, u.y * k, u.z * k, u.w * k};
            }
            if (accum) { const OgF4 o = ((const OgF4 *)pcm)[q]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            ((OgF4 *)pcm)[q] = v;
        }
    } else {
        for (int q = tid; q < N * CC; q += OG_WG) {
            const int j = CC == 2 ? q >> 1 : q, ch = CC == 2 ? q & 1 : 0;
            const float v = tbuf[ch * 960 + j] * (1.f / 32768.f);
            pcm[q] = accum ? pcm[q] + v : v;
        }
    }
    // history ring: pos0 and N are multiples of 8, so groups of four samples never straddle the wrap
    for (int q = tid; q < CC * (N >> 2); q += OG_WG) {
        const int ch = q >= (N >> 2), i = 4 * (q - ch * (N >> 2));
        *(OgF4 *)&st->hist[ch][(pos0 + (uint32_t)i) & 1023u] = *(const OgF4 *)&L.v.out[ch][OG_HIST + i];
    }
    for (int q = tid; q < CC * 15; q += OG_WG) {
        const int ch = q >= 15, i = 4 * (q - ch * 15);
        *(OgF4 *)&st->overlap[ch][i] = *(const OgF4 *)&L.v.out[ch][OG_HIST + N + i];
    }
    if (reset)                                          // forget the ring's older samples
        for (int q = tid; q < CC * ((1024 - N) >> 2); q += OG_WG) {
            const int ch = q >= ((1024 - N) >> 2), i = 4 * (q - ch * ((1024 - N) >> 2));
            *(OgF4 *)&st->hist[ch][N + i] = OgF4{0.f, 0.f, 0.f, 0.f};
        }
    if (tid == 0) st->pos = pos0 + (uint32_t)N;
    OG_LANES_END
    OG_STAMP(8)
}
