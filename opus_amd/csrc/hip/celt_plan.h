// CELT band plan: the integer half of decode_bands (celt/decoder.rs:2037-2220), one lane per frame.
//
// Two things make the reference's band walk sequential: folding reads the finished lower bands, and the noise seed runs
// through the zero-pulse ("fill") leaves in decode order.  Which fill leaves are live, their seeds, and every collapse
// mask are integer recurrences over a few dozen records -- slow as uniform code inside a 256-lane workgroup (each step
// waits a full LDS round trip), trivial as 32768 independent lanes.  So after K1 (which produced the PVQ leaves'
// collapse masks) one lane per frame derives:
//   * a task record per quant_band call (band, buffer): geometry of its fused Haar/Hadamard transform, where its
//     lowband_out goes, where it folds from, and its dependency level (0 = no fill leaves; otherwise 1 + the level of
//     any fill band it folds from, and past the dual -> intensity switch of `norm`);
//   * the list of live fill leaves ("fill ops") with their starting seeds, in decode order (a call's are consecutive);
//   * the collapse masks of all bands and the seed the frame ends its bands with.
// The frame kernel then runs level by level with no serial integer work left.
#pragma once
#include "wg.h"
#include "../../../include/opusgpu.h"
#include "celt_frame_types.h"

struct alignas(16) OgTask {           // one quant_band call = (band, buffer), task id = 2 * band + buffer; 16 bytes (one store)
    uint16_t boff, n;                 // first bin in the channel buffer, width
    uint16_t n0p, lowpos;             // OgXform.n0p; offset of the folding source in its norm row
    uint8_t flags, hm, slog, rctd;    // OG_T_*; OgXform.hm / .slog; recombine | time_divide << 4
    uint8_t norm_row, level;          // norm row lowband_out goes to (0xFF = none); dependency level
    uint16_t ops;                     // its live fill leaves (consecutive in OgBandPlan.op): first index | count << 10
};
enum { OG_T_ACTIVE = 1, OG_T_CHAIN = 2, OG_T_NEED_LB = 4, OG_T_LONG = 8, OG_T_LB_ROW1 = 16 };

struct OgFillOp {                     // one live fill leaf (celt/decoder.rs:1633-1661); 16 bytes
    uint32_t seed;                    // noise seed before the leaf
    float gain;                       // the leaf's renormalisation target
    uint16_t pos;                     // buffer * 960 + first bin
    uint8_t n, flags;                 // width; OG_OP_FOLD: copy of the lower band + dither, else noise
    uint8_t level, task, pad_[2];     // level and id of the quant_band call it belongs to
};
enum { OG_OP_FOLD = 1 };
struct OgLeafWords { uint32_t w[4]; };                  // an OgCeltLeaf: index | gain | off, n << 8, k << 16, meta << 24 | cmshift, band << 8, pos << 16
static_assert(sizeof(OgCeltLeaf) == 16 && offsetof(OgCeltLeaf, gain) == 4 && offsetof(OgCeltLeaf, off) == 8 && offsetof(OgCeltLeaf, n) == 9 &&
              offsetof(OgCeltLeaf, meta) == 11 && offsetof(OgCeltLeaf, cmshift) == 12, "OgCeltLeaf as four words");
struct OgFillOpWords { uint32_t w[4]; };                // seed | gain | pos, n << 16, flags << 24 | level, task << 8
static_assert(sizeof(OgFillOp) == 16 && offsetof(OgFillOp, pos) == 8 && offsetof(OgFillOp, n) == 10 && offsetof(OgFillOp, flags) == 11 &&
              offsetof(OgFillOp, level) == 12 && offsetof(OgFillOp, task) == 13, "OgFillOp as four words");

#define OG_MAXLEVEL 23                // 21 bands + the intensity switch, and level 0
struct alignas(16) OgBandPlan {
    OgTask task[42];
    uint8_t collapse[2][24];
    uint32_t seed_out;
    uint16_t nops, merge_cnt;         // live fill leaves; norm bins below the dual -> intensity switch
    uint8_t nlevels, merge_level;     // highest level in use; level the switch precedes (0xFF = none)
    uint8_t pad_[2];
    uint32_t plain[2];                // per buffer, bit = band: a call with no fill leaves and no transform whose lowband_out goes to
                                      // the norm row of its buffer's number: the frame kernel does it while staging (not in `order`)
    uint32_t joint;                   // bit = band: coded jointly (mid / side), i.e. the frame kernel has a stereo merge to do
    uint32_t keep[2];                 // per buffer, bit = band: K1 (or the plan, for one-bin bands) wrote this band of the frame's
                                      // scratch row; the frame kernel reads every other bin as zero (bits 21..31 are zero)
    uint8_t order[44];                // active task ids sorted by level (band order inside a level)
    uint8_t task_start[OG_MAXLEVEL + 3];   // tasks of level l: order[task_start[l]] .. order[task_start[l + 1]]
    uint8_t pad2_[3];
    OgFillOp op[OG_MAXLEAF];          // decode order
};
#define OG_PLAN_HEAD_BYTES ((int)offsetof(OgBandPlan, op))

// The Haar steps of one quant_band call -- tf recombination (celt/decoder.rs:1708-1716, strides 1, 2, 4) and tf
// time-division (:1722-1740, strides B, 2B, ...) -- are butterflies on distinct bits of the coefficient index, so
// together they are a Walsh-Hadamard transform over the bit mask `hm` (a run of m <= 3 bits), scaled by 2^(-m/2).  The
// Hadamard (de)interleave (:472-538) that follows / precedes them is a permutation.  Fused, every output coefficient
// is one gather of 2^m inputs.
struct OgXform {
    int hm;                           // index bits the butterflies act on
    int slog, n0p, had;               // reorder: stride 2^slog (0 = none), row length, natural / ordery rows
    float scale;                      // 2^(-popcount(hm)/2)
};

OG_DEV void og_make_xform(int n, int B, int tf_change, OgXform *t, int *recombine_out, int *time_divide_out, int *B0_out)
{
    const int long_blocks = B == 1;
    int n_b = n >> (31 - __builtin_clz((unsigned)B)), recombine = tf_change > 0 ? tf_change : 0, time_divide = 0;
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

// The plan of a frame is made in three steps, so that only what is a recurrence over the bands runs as one:
//   A  per band (21 independent work items): the band's two task records (kept by the work item), the masks `keep` / `joint` /
//      `plain`, the one-bin bands' signs, the collapse masks of the calls without live fill leaves; K1's masks are consumed
//      (and re-zeroed) here;
//   B  per frame (serial over the bands that have fill leaves or the dual -> intensity switch): levels, which fill leaves
//      are live and their seeds, the collapse masks that depend on lower bands; then the tasks' order by level (og_plan_sort);
//   C  per band again: step B's amendments applied to the two task records, records out; the low work items also write
//      the order and the header fields.
// OgPlanShared holds what passes between the steps (LDS in the kernel, one per frame of the workgroup: 628 bytes, so that
// eight workgroups of 16 frames fit a CU).  The steps read the frame's header through its head (OgHdrHead) and its band
// records, which is what the kernel keeps of it in LDS.
struct OgHdrHead {                    // the head of OgCeltHdr (everything up to the band arrays), fetched in one go
    uint32_t stream, pcm_off, leaf_begin, seed;
    uint16_t nleaves;
    uint8_t lm, flags, start, end, spread, intensity;     // spread: also the cross-fade kind and launch round (opusgpu_sideinfo.h)
    uint16_t pf_period[3];
    uint8_t pf_gain_q[3], pf_tapset[3];
};
static_assert(sizeof(OgHdrHead) == 36 && offsetof(OgCeltHdr, pf_tapset) == offsetof(OgHdrHead, pf_tapset) &&
              offsetof(OgCeltHdr, flags) == offsetof(OgHdrHead, flags), "OgHdrHead mirrors the start of OgCeltHdr");

enum { OG_PB_CHAIN0 = 1, OG_PB_CHAIN1 = 2, OG_PB_SERIAL = 4 /* step B has work here */, OG_PB_CODED = 8 /* inside [start, end), not silent */ };
struct OgPlanBand {                   // step A -> step B
    uint8_t flags;                    // OG_PB_*
    uint8_t cm[2];                    // collapse masks of the calls step B leaves alone
    uint8_t rctd[2], B0[2];           // per call: recombine | time_divide << 4; blocks after recombination (og_make_xform)
    uint8_t pad_;
    uint16_t raw[2];                  // K1's collapse mask of the call's PVQ leaves
};
// step B's amendment of a task with live fill leaves: level | OG_PA_NEED_LB | OG_PA_LB_ROW1 | ops << 16 (0: none)
enum { OG_PA_LEVEL = 31, OG_PA_NEED_LB = 32, OG_PA_LB_ROW1 = 64 };
struct OgPlanShared {
    OgPlanBand rec[21];
    uint32_t amend[42];
    uint32_t bits[5];                 // keep[0], keep[1], joint, plain[0], plain[1]: OR-ed together by the bands' work items
    uint32_t seed_out;
    uint16_t nops, merge_cnt;
    uint8_t maxlvl, merge_level;
    uint8_t tl[42];                   // level of the active tasks that go into `order` (0xFF: not in it)
    uint8_t col[2][21], blevel[21];   // step B's running arrays (col: the bands' collapse masks, as the fill of higher bands reads them)
    uint8_t pad_[1];
    uint8_t order[44];                // as in OgBandPlan (dword-aligned here as there: step C copies them as dwords)
    uint8_t task_start[OG_MAXLEVEL + 3], zero_[2];      // + the plan's two pad bytes behind it
};
static_assert(offsetof(OgPlanShared, order) % 4 == 0 && offsetof(OgPlanShared, task_start) % 4 == 0 && offsetof(OgBandPlan, order) % 4 == 0 &&
              offsetof(OgBandPlan, task_start) % 4 == 0 && offsetof(OgBandPlan, task_start) + OG_MAXLEVEL + 3 + 3 <= offsetof(OgBandPlan, op),
              "order and task_start are copied as dwords");
static_assert(sizeof(OgPlanBand) == 12 && sizeof(OgPlanShared) == 628, "OgPlanShared: an odd number of dwords (the frames' lanes of step B hit distinct banks)");
// (Step B fetches its fill leaves from the leaf array, one memory round trip each.  Staging them in LDS in step A was
// measured, round 2: no faster for a small launch, 25 % slower for a large one -- the LDS it takes costs a workgroup per CU.)

// the fill leaves step B walks for call `slot` of a band: at most nfill, and none past the frame's leaves
OG_DEV int og_plan_nfill(const OgCeltBand &bd, int slot, int nleaves)
{
    const int g0 = bd.leaf0 + (slot ? bd.nx : 0), nfill = slot ? bd.fy : bd.fx, left = nleaves - g0;
    return left <= 0 ? 0 : nfill < left ? nfill : left;
}

// cm_row: K1's collapse masks [buffer][24] (global: read and re-zeroed here); coef_row: the frame's K1 output (the signs of the
// one-bin bands are added to it); tk[2]: the band's task records
OG_DEV void og_plan_band_a(const OgHdrHead &h, const OgCeltBand *bands, const OgCeltLeaf *lv, const OgCeltConst &K, uint32_t *cm_row,
                           float *coef_row, int i, OgPlanShared &S, OgTask tk[2])
{
    const int lm = h.lm, M = 1 << lm, start = h.start, end = h.end;
    const int C = (h.flags & OG_F_STEREO) ? 2 : 1;
    const int B = (h.flags & OG_F_TRANSIENT) ? M : 1;
    const int nleaves = h.nleaves < OG_MAXLEAF ? h.nleaves : OG_MAXLEAF;
    const bool silence = (h.flags & OG_F_SILENCE) != 0;
    OgPlanBand r;
    r.flags = 0; r.cm[0] = r.cm[1] = 0; r.rctd[0] = r.rctd[1] = 0; r.B0[0] = r.B0[1] = 1; r.pad_ = 0;
    r.raw[0] = (uint16_t)cm_row[i]; r.raw[1] = (uint16_t)cm_row[24 + i];
    cm_row[i] = 0; cm_row[24 + i] = 0;
    int tl[2] = {0xFF, 0xFF};
    for (int slot = 0; slot < 2; slot++) {
        OgTask &t = tk[slot];
        t.boff = t.n = t.n0p = t.ops = 0; t.lowpos = 0;
        t.flags = t.hm = t.slog = t.rctd = t.level = 0; t.norm_row = 0xFF;
    }
    if (!(silence || i < start || i >= end)) {
        const OgCeltBand bd = bands[i];
        const int n = M * K.width[i], boff = M * K.ebands[i];
        r.flags = OG_PB_CODED;
        // leaves of this band that write a buffer with pulses (or a sign): all of them but the fill-path ones
        if (C == 2 && (bd.flags & OG_B_JOINT)) OG_ATOMIC_OR(&S.bits[2], 1u << i);
        if ((int)bd.nx - (int)bd.fx > 0) OG_ATOMIC_OR(&S.bits[0], 1u << i);
        if ((int)bd.ny - (int)bd.fy > 0) OG_ATOMIC_OR(&S.bits[1], 1u << i);
        const bool dual = (bd.flags & OG_B_DUAL) != 0;
        for (int slot = 0; slot < 2; slot++) {
            OgTask &t = tk[slot];
            const int nfill = slot ? bd.fy : bd.fx;
            bool active, chain = false;
            int norm_row = -1;
            if (C == 1) { active = slot == 0; norm_row = 0; chain = nfill > 0; }
            else if (dual) { active = true; norm_row = slot; chain = nfill > 0; }
            else if (n == 1) { active = true; norm_row = slot == 0 ? 0 : -1; }
            else if (n == 2) { active = slot == ((bd.flags & OG_B_N2SWAP) ? 1 : 0); norm_row = 0; chain = nfill > 0; }
            else { active = true; norm_row = slot == 0 ? 0 : -1; chain = slot == 0 && nfill > 0 && !(bd.flags & OG_B_MIDFILL0); }
            if (bd.flags & OG_B_LAST) norm_row = -1;
            if (!active) continue;
            t.boff = (uint16_t)boff; t.n = (uint16_t)n;
            t.flags = (uint8_t)(OG_T_ACTIVE | (chain ? OG_T_CHAIN : 0) | (B == 1 ? OG_T_LONG : 0));
            t.norm_row = (uint8_t)norm_row;
            tl[slot] = 0;
            if (n == 1) {                               // decode_band_1 (celt/decoder.rs:1466-1495)
                const int li = bd.leaf0 + (slot ? bd.nx : 0);
                coef_row[slot * 960 + boff] = (li < nleaves && lv[li].index) ? -1.f : 1.f;
                r.cm[slot] = 1;
                continue;
            }
            OgXform xf;
            int rc, td, B0;
            og_make_xform(n, B, bd.tf_change, &xf, &rc, &td, &B0);
            t.hm = (uint8_t)xf.hm; t.slog = (uint8_t)xf.slog; t.n0p = (uint16_t)xf.n0p;
            t.rctd = (uint8_t)(rc | td << 4);
            r.rctd[slot] = t.rctd; r.B0[slot] = (uint8_t)B0;
            if (chain) {
                r.flags |= (uint8_t)(OG_PB_CHAIN0 << slot);
            } else {                                    // its fill leaves (if any) are dead: the PVQ masks are all
                int m = r.raw[slot];
                for (int k = 0, Bk = B0; k < td; k++) { Bk >>= 1; m |= m >> Bk; }
                for (int k = 0; k < rc; k++) m = K.bit_deinterleave[m & 255];
                r.cm[slot] = (uint8_t)(m & ((1 << ((B0 >> td) << rc)) - 1));
            }
        }
        if ((r.flags & (OG_PB_CHAIN0 | OG_PB_CHAIN1)) || (bd.flags & OG_B_MERGE_NORM)) r.flags |= OG_PB_SERIAL;
        for (int slot = 0; slot < 2; slot++) {          // the calls that stay at level 0
            const OgTask &t = tk[slot];
            if (tl[slot] == 0 && !(t.flags & OG_T_CHAIN) && !(t.hm | t.slog) && (t.norm_row == 0xFF || t.norm_row == slot)) {
                // nothing to transform: at most lowband_out = sqrt(N0) x, which the frame kernel writes while staging
                if (t.norm_row == slot) OG_ATOMIC_OR(&S.bits[3 + slot], 1u << i);
                tl[slot] = 0xFF;
            }
        }
    }
    S.rec[i] = r;
    S.tl[2 * i] = (uint8_t)tl[0];
    S.tl[2 * i + 1] = (uint8_t)tl[1];
}

// K, la, lc: the small constant tables (LDS copies in the kernel).  Fill ops go straight to the plan P.  S.amend, S.col,
// S.blevel, S.order, S.task_start and S.zero_ are zero on entry.
OG_DEV void og_plan_chain(const OgHdrHead &h, const OgCeltBand *bands, const OgCeltLeaf *lv, const OgCeltConst &K, const uint32_t *la,
                          const uint32_t *lc, OgBandPlan *P, OgPlanShared &S)
{
    const int lm = h.lm, M = 1 << lm, start = h.start;
    const int C = (h.flags & OG_F_STEREO) ? 2 : 1;
    const int B = (h.flags & OG_F_TRANSIENT) ? M : 1;
    const int nleaves = h.nleaves < OG_MAXLEAF ? h.nleaves : OG_MAXLEAF;
    const int norm_offset = M * K.ebands[start < 21 ? start : 20];
    uint32_t seed = h.seed;
    int floor_lvl = 1, maxlvl = 0, merge_level = 0xFF, merge_cnt = 0, nops = 0;
    for (int i = 0; i < 21; i++) {
        const OgPlanBand r = S.rec[i];
        int cm[2] = {r.cm[0], r.cm[1]};
        bool dual = false;
        if (r.flags & OG_PB_SERIAL) {
            const OgCeltBand bd = bands[i];
            const int boff = M * K.ebands[i];
            const bool any_chain = (r.flags & (OG_PB_CHAIN0 | OG_PB_CHAIN1)) != 0;
            dual = (bd.flags & OG_B_DUAL) != 0;
            int x_cm, y_cm, lvl = floor_lvl;
            if (bd.lowband != 0xFF) {
                x_cm = y_cm = 0;
                for (int f = bd.fold_lo; f < bd.fold_hi && f < 21; f++) {
                    x_cm |= S.col[0][f];
                    y_cm |= S.col[1][f];
                    if (S.blevel[f] >= lvl) lvl = S.blevel[f] + 1;
                }
            } else {
                x_cm = y_cm = (1 << B) - 1;
            }
            if (bd.flags & OG_B_MERGE_NORM) {           // dual -> intensity switch (celt/decoder.rs:2133-2138): every lower
                if (lvl <= maxlvl) lvl = maxlvl + 1;    // band is finished first, every later fill band comes after
                floor_lvl = merge_level = lvl;
                merge_cnt = boff - norm_offset;
            }
            if (lvl > OG_MAXLEVEL) lvl = OG_MAXLEVEL;   // cannot happen (one level per band at most, plus the switch)
            for (int slot = 0; slot < 2; slot++) {
                if (!(r.flags & (OG_PB_CHAIN0 << slot))) continue;
                const int rc = r.rctd[slot] & 15, td = r.rctd[slot] >> 4, B0 = r.B0[slot];
                // the integer half of a quant_band call with fill leaves: fill transforms (celt/decoder.rs:1700-1740),
                // which leaves fold / make noise (:1633-1661), mask transforms (:1994-2019)
                int fill = dual ? (slot ? y_cm : x_cm) : (x_cm | y_cm);
                for (int k = 0; k < rc; k++) fill = K.bit_interleave[fill & 0xF] | K.bit_interleave[(fill >> 4) & 0xF] << 2;
                for (int k = 0, Bk = B >> rc; k < td; k++, Bk <<= 1) fill |= fill << Bk;
                uint32_t amend = (uint32_t)lvl;
                S.tl[2 * i + slot] = (uint8_t)lvl;
                if (bd.lowband != 0xFF && (bd.flags & (slot ? OG_B_FOLD_Y : OG_B_FOLD_X)))
                    amend |= OG_PA_NEED_LB | ((dual && slot) ? OG_PA_LB_ROW1 : 0);
                const int g0 = bd.leaf0 + (slot ? bd.nx : 0), nfill = og_plan_nfill(bd, slot, nleaves);
                int m = r.raw[slot];                    // PVQ leaves: expanded (and masks collected) by K1
                int live = 0;
                const int first_op = nops;
                // (the next leaf is on its way while this one is worked on; a record is fetched as its four words in one piece --
                // field by field it took a load per field -- and the fields read here are cut out of the words)
                OgLeafWords nx = {{0, 0, 0, 0}};
                if (nfill > 0) nx = *(const OgLeafWords *)&lv[g0];
                for (int l = 0; l < nfill; l++) {
                    const OgLeafWords lw = nx;
                    if (l + 1 < nfill) nx = *(const OgLeafWords *)&lv[g0 + l + 1];
                    struct { float gain; int off, n, meta, cmshift; } lf;
                    memcpy(&lf.gain, &lw.w[1], 4);
                    lf.off = (int)(lw.w[2] & 255u); lf.n = (int)((lw.w[2] >> 8) & 255u); lf.meta = (int)(lw.w[2] >> 24); lf.cmshift = (int)(lw.w[3] & 255u);
                    const int Bl = 1 << (lf.meta & OG_L_LOGB_MASK), sh = lf.cmshift;
                    // the leaf's share of the fill: the recursion only shifts it (side halves) or clears it (theta)
                    const int fl = (lf.meta & OG_L_FILL_DEAD) ? 0 : (fill >> sh) & ((1 << Bl) - 1);
                    if (!fl) continue;                  // x stays zero (celt/decoder.rs:1640-1643)
                    const bool fold = (lf.meta & OG_L_HAS_LOWBAND) != 0;
                    const int nn = lf.n < 184 ? lf.n : 183;
                    if (nops < OG_MAXLEAF) {
                        // the record as its four words, one store (field by field it went out in four pieces)
                        OgFillOpWords ow;
                        float g = lf.gain;
                        ow.w[0] = seed;
                        memcpy(&ow.w[1], &g, 4);
                        ow.w[2] = (uint32_t)((slot * 960 + boff + lf.off) & 0xFFFF) | (uint32_t)nn << 16 | (fold ? (uint32_t)OG_OP_FOLD << 24 : 0u);
                        ow.w[3] = (uint32_t)lvl | (uint32_t)(2 * i + slot) << 8;
                        *(OgFillOpWords *)&P->op[nops++] = ow;
                        live++;
                    }
                    m |= (fold ? fl : (1 << Bl) - 1) << sh;
                    seed = la[nn] * seed + lc[nn];
                }
                for (int k = 0, Bk = B0; k < td; k++) { Bk >>= 1; m |= m >> Bk; }
                for (int k = 0; k < rc; k++) m = K.bit_deinterleave[m & 255];
                cm[slot] = m & ((1 << ((B0 >> td) << rc)) - 1);
                S.amend[2 * i + slot] = amend | (uint32_t)(live << 10 | (live ? first_op : 0)) << 16;
            }
            if (any_chain) S.blevel[i] = (uint8_t)lvl;
            if (lvl > maxlvl) maxlvl = lvl;
        } else if (r.flags & OG_PB_CODED) {
            dual = (bands[i].flags & OG_B_DUAL) != 0;
        }
        const int both = cm[0] | cm[1];
        S.col[0][i] = (uint8_t)(dual || C == 1 ? cm[0] : both);
        S.col[1][i] = (uint8_t)(dual ? cm[1] : C == 1 ? cm[0] : both);      // (mono: the fill of a band reads col[0] for both)
    }
    S.seed_out = seed;
    S.nops = (uint16_t)nops; S.merge_cnt = (uint16_t)merge_cnt;
    S.maxlvl = (uint8_t)maxlvl; S.merge_level = (uint8_t)merge_level;
}

// The active tasks ordered by level (band order inside a level): S.order, S.task_start from S.tl and S.maxlvl.  This is one
// lane's form (CPU builds): a pass over the 42 levels per level in use.  The kernel does it with a lane per task and a
// wavefront vote per level (celt_launch.hip, og_plan_kernel) -- in the frame's chain it was 9 of the chain's 40 us.
// (levels above maxlvl hold nothing: the frame kernel reads task_start[0 .. nlevels + 1])
OG_DEV void og_plan_sort(OgPlanShared &S)
{
    const int maxlvl = S.maxlvl;
    int n = 0;
    for (int l = 0; l <= maxlvl; l++) {
        S.task_start[l] = (uint8_t)n;
        for (int t = 0; t < 42; t++)
            if (S.tl[t] == l) S.order[n++] = (uint8_t)t;
    }
    S.task_start[maxlvl + 1] = S.task_start[maxlvl + 2] = (uint8_t)n;
}

// the work item of band i again: step B's amendments applied to its two task records, records out; the low work items also
// write the order and the header fields
OG_DEV void og_plan_emit(const OgHdrHead &h, const OgCeltBand *bands, int i, OgTask tk[2], OgBandPlan *P, const OgPlanShared &S)
{
    for (int slot = 0; slot < 2; slot++) {
        const uint32_t am = S.amend[2 * i + slot];
        if (am) {
            OgTask &t = tk[slot];
            t.level = (uint8_t)(am & OG_PA_LEVEL);
            if (am & OG_PA_NEED_LB) {
                t.flags |= OG_T_NEED_LB | ((am & OG_PA_LB_ROW1) ? OG_T_LB_ROW1 : 0);
                t.lowpos = (uint16_t)(bands[i].lowband << h.lm);
            }
            t.ops = (uint16_t)(am >> 16);
        }
        P->task[2 * i + slot] = tk[slot];
    }
    const int t = i;
    if (t < 11) ((uint32_t *)&P->order[0])[t] = ((const uint32_t *)&S.order[0])[t];                 // order[44]
    if (t < (OG_MAXLEVEL + 3 + 3) / 4) ((uint32_t *)&P->task_start[0])[t] = ((const uint32_t *)&S.task_start[0])[t];   // task_start[26] + pad2_
    if (t < 12) {                                       // collapse[2][24], four bytes per work item (bands 21 .. 23: zero)
        const int b = t >= 6, i0 = 4 * (t - 6 * b);
        uint32_t w = 0;
        for (int k = 0; k < 4; k++)
            if (i0 + k < 21) w |= (uint32_t)S.col[b][i0 + k] << (8 * k);
        ((uint32_t *)&P->collapse[0][0])[t] = w;
    }
    if (t == 0) {
        P->seed_out = S.seed_out;
        P->nops = S.nops; P->merge_cnt = S.merge_cnt;
        P->nlevels = S.maxlvl; P->merge_level = S.merge_level;
        P->pad_[0] = P->pad_[1] = 0;
        P->keep[0] = S.bits[0]; P->keep[1] = S.bits[1]; P->joint = S.bits[2]; P->plain[0] = S.bits[3]; P->plain[1] = S.bits[4];
        P->pad2_[2] = 0;
    }
}

// the three steps for one frame, one after the other (CPU builds; the kernel spreads A and C over lanes).  cm_row is consumed.
OG_DEV void og_band_plan(const OgCeltHdr *hdr, const OgCeltLeaf *lv, const OgCeltConst &K, const uint32_t *la, const uint32_t *lc,
                         uint32_t *cm_row, float *coef_row, OgBandPlan *P, OgPlanShared &S)
{
    const OgHdrHead h = *(const OgHdrHead *)hdr;
    OgTask tk[21][2];
    for (size_t k = 0; k < sizeof(OgPlanShared) / 4; k++) ((uint32_t *)&S)[k] = 0;      // (the kernel clears it with all its lanes)
    for (int i = 0; i < 21; i++) og_plan_band_a(h, hdr->band, lv, K, cm_row, coef_row, i, S, tk[i]);
    og_plan_chain(h, hdr->band, lv, K, la, lc, P, S);
    og_plan_sort(S);
    for (int i = 0; i < 21; i++) og_plan_emit(h, hdr->band, i, tk[i], P, S);
}

// The same plan made start to end by ONE lane (the form large launches use: og_plan_lane_kernel, celt_launch.hip): the three
// steps above fused, tasks written straight to the plan, the order by a counting sort.  Per-lane serial code on global memory.  coef_row: the frame's K1 output (the signs of the
// one-bin bands are added to it); cm_row: K1's collapse masks [buffer][24].
// K, la, lc: the small constant tables (LDS copies in the kernel).  sc: OG_PLAN_SCRATCH cells private to the lane: the
// small per-band / per-level arrays are indexed with run-time indices, which in registers costs a chain of selects per
// access (it was most of this code's instructions); in LDS it is one access.
struct OgLaneScratch {                // u16 cells, cell i at p[i * stride] (lane-interleaved in the kernel, stride 1 on the CPU)
    uint16_t *p;
    int stride;
    OG_MEMBER uint16_t &operator[](int i) const { return p[i * stride]; }
    OG_MEMBER OgLaneScratch at(int o) const { OgLaneScratch r = {p + o * stride, stride}; return r; }
};
#define OG_PLAN_SCRATCH 184
OG_DEV void og_band_plan_lane(const OgCeltHdr *h, const OgCeltLeaf *lv, const OgCeltConst &K, const uint32_t *la, const uint32_t *lc,
                         const uint32_t *cm_row, float *coef_row, OgBandPlan *P, OgLaneScratch sc)
{
    const OgLaneScratch tl = sc.at(0), blevel = sc.at(44), col0 = sc.at(65), col1 = sc.at(86), tcur = sc.at(157);
    const int lm = h->lm, M = 1 << lm, start = h->start, end = h->end;
    const int C = (h->flags & OG_F_STEREO) ? 2 : 1;
    const int B = (h->flags & OG_F_TRANSIENT) ? M : 1;
    const int nleaves = h->nleaves < OG_MAXLEAF ? h->nleaves : OG_MAXLEAF;
    const int norm_offset = M * K.ebands[start < 21 ? start : 20];
    for (int i = 0; i < 21; i++) blevel[i] = col0[i] = col1[i] = 0;
    uint32_t seed = h->seed, keep0 = 0, keep1 = 0, joint = 0, plain0 = 0, plain1 = 0;
    int floor_lvl = 1, maxlvl = 0, merge_level = 0xFF, merge_cnt = 0, nops = 0;
    const bool silence = (h->flags & OG_F_SILENCE) != 0;

    for (int i = 0; i < 21; i++) {
        OgTask tk[2];
        int cm[2] = {0, 0};
        for (int slot = 0; slot < 2; slot++) {
            OgTask &t = tk[slot];
            t.boff = t.n = t.n0p = t.ops = 0; t.lowpos = 0;
            t.flags = t.hm = t.slog = t.rctd = t.level = 0; t.norm_row = 0xFF;
        }
        if (silence || i < start || i >= end) {
            P->task[2 * i] = tk[0]; P->task[2 * i + 1] = tk[1];
            tl[2 * i] = tl[2 * i + 1] = 0xFF;
            continue;
        }
        const OgCeltBand bd = h->band[i];
        const int n = M * K.width[i], boff = M * K.ebands[i];
        // leaves of this band that write a buffer with pulses (or a sign): all of them but the fill-path ones
        if (C == 2 && (bd.flags & OG_B_JOINT)) joint |= 1u << i;
        if ((int)bd.nx - (int)bd.fx > 0) keep0 |= 1u << i;
        if ((int)bd.ny - (int)bd.fy > 0) keep1 |= 1u << i;
        const bool dual = (bd.flags & OG_B_DUAL) != 0;
        bool any_chain = false;
        int rc[2] = {0, 0}, td[2] = {0, 0}, B0[2] = {1, 1};
        for (int slot = 0; slot < 2; slot++) {
            OgTask &t = tk[slot];
            const int nfill = slot ? bd.fy : bd.fx;
            bool active, chain = false;
            int norm_row = -1;
            if (C == 1) { active = slot == 0; norm_row = 0; chain = nfill > 0; }
            else if (dual) { active = true; norm_row = slot; chain = nfill > 0; }
            else if (n == 1) { active = true; norm_row = slot == 0 ? 0 : -1; }
            else if (n == 2) { active = slot == ((bd.flags & OG_B_N2SWAP) ? 1 : 0); norm_row = 0; chain = nfill > 0; }
            else { active = true; norm_row = slot == 0 ? 0 : -1; chain = slot == 0 && nfill > 0 && !(bd.flags & OG_B_MIDFILL0); }
            if (bd.flags & OG_B_LAST) norm_row = -1;
            if (!active) continue;
            t.boff = (uint16_t)boff; t.n = (uint16_t)n;
            t.flags = (uint8_t)(OG_T_ACTIVE | (chain ? OG_T_CHAIN : 0) | (B == 1 ? OG_T_LONG : 0));
            t.norm_row = (uint8_t)norm_row;
            if (n == 1) {                               // decode_band_1 (celt/decoder.rs:1466-1495)
                const int li = bd.leaf0 + (slot ? bd.nx : 0);
                coef_row[slot * 960 + boff] = (li < nleaves && lv[li].index) ? -1.f : 1.f;
                cm[slot] = 1;
                continue;
            }
            OgXform xf;
            og_make_xform(n, B, bd.tf_change, &xf, &rc[slot], &td[slot], &B0[slot]);
            t.hm = (uint8_t)xf.hm; t.slog = (uint8_t)xf.slog; t.n0p = (uint16_t)xf.n0p;
            t.rctd = (uint8_t)(rc[slot] | td[slot] << 4);
            any_chain |= chain;
            if (!chain) {                               // its fill leaves (if any) are dead: the PVQ masks are all
                int m = (int)cm_row[slot * 24 + i];
                for (int k = 0, Bk = B0[slot]; k < td[slot]; k++) { Bk >>= 1; m |= m >> Bk; }
                for (int k = 0; k < rc[slot]; k++) m = K.bit_deinterleave[m & 255];
                cm[slot] = m & ((1 << ((B0[slot] >> td[slot]) << rc[slot])) - 1);
            }
        }
        int lvl = 0;
        if (any_chain || (bd.flags & OG_B_MERGE_NORM)) {
            int x_cm, y_cm;
            lvl = floor_lvl;
            if (bd.lowband != 0xFF) {
                x_cm = y_cm = 0;
                for (int f = bd.fold_lo; f < bd.fold_hi && f < 21; f++) {
                    x_cm |= col0[f];
                    y_cm |= C == 2 ? col1[f] : col0[f];
                    if (blevel[f] >= lvl) lvl = blevel[f] + 1;
                }
            } else {
                x_cm = y_cm = (1 << B) - 1;
            }
            if (bd.flags & OG_B_MERGE_NORM) {           // dual -> intensity switch (celt/decoder.rs:2133-2138): every lower
                if (lvl <= maxlvl) lvl = maxlvl + 1;    // band is finished first, every later fill band comes after
                floor_lvl = merge_level = lvl;
                merge_cnt = boff - norm_offset;
            }
            if (lvl > OG_MAXLEVEL) lvl = OG_MAXLEVEL;   // cannot happen (one level per band at most, plus the switch)
            for (int slot = 0; slot < 2; slot++) {
                OgTask &t = tk[slot];
                if (!(t.flags & OG_T_CHAIN)) continue;
                // the integer half of a quant_band call with fill leaves: fill transforms (celt/decoder.rs:1700-1740),
                // which leaves fold / make noise (:1633-1661), mask transforms (:1994-2019)
                int fill = dual ? (slot ? y_cm : x_cm) : (x_cm | y_cm);
                for (int k = 0; k < rc[slot]; k++) fill = K.bit_interleave[fill & 0xF] | K.bit_interleave[(fill >> 4) & 0xF] << 2;
                for (int k = 0, Bk = B >> rc[slot]; k < td[slot]; k++, Bk <<= 1) fill |= fill << Bk;
                t.level = (uint8_t)lvl;
                if (bd.lowband != 0xFF && (bd.flags & (slot ? OG_B_FOLD_Y : OG_B_FOLD_X))) {
                    t.flags |= OG_T_NEED_LB | ((dual && slot) ? OG_T_LB_ROW1 : 0);
                    t.lowpos = (uint16_t)(bd.lowband * M);
                }
                const int g0 = bd.leaf0 + (slot ? bd.nx : 0), nfill = slot ? bd.fy : bd.fx;
                int m = (int)cm_row[slot * 24 + i];     // PVQ leaves: expanded (and masks collected) by K1
                int live = 0;
                const int first_op = nops;
                for (int l = 0; l < nfill && g0 + l < nleaves; l++) {
                    const OgCeltLeaf lf = lv[g0 + l];
                    const int Bl = 1 << (lf.meta & OG_L_LOGB_MASK), sh = lf.cmshift;
                    // the leaf's share of the fill: the recursion only shifts it (side halves) or clears it (theta)
                    const int fl = (lf.meta & OG_L_FILL_DEAD) ? 0 : (fill >> sh) & ((1 << Bl) - 1);
                    if (!fl) continue;                  // x stays zero (celt/decoder.rs:1640-1643)
                    const bool fold = (lf.meta & OG_L_HAS_LOWBAND) != 0;
                    const int nn = lf.n < 184 ? lf.n : 183;
                    if (nops < OG_MAXLEAF) {
                        OgFillOp op;
                        op.seed = seed; op.gain = lf.gain;
                        op.pos = (uint16_t)(slot * 960 + boff + lf.off);
                        op.n = (uint8_t)nn; op.flags = fold ? OG_OP_FOLD : 0;
                        op.level = (uint8_t)lvl; op.task = (uint8_t)(2 * i + slot); op.pad_[0] = op.pad_[1] = 0;
                        P->op[nops++] = op;
                        live++;
                    }
                    m |= (fold ? fl : (1 << Bl) - 1) << sh;
                    seed = la[nn] * seed + lc[nn];
                }
                for (int k = 0, Bk = B0[slot]; k < td[slot]; k++) { Bk >>= 1; m |= m >> Bk; }
                for (int k = 0; k < rc[slot]; k++) m = K.bit_deinterleave[m & 255];
                cm[slot] = m & ((1 << ((B0[slot] >> td[slot]) << rc[slot])) - 1);
                t.ops = (uint16_t)(live << 10 | (live ? first_op : 0));
            }
            if (any_chain) blevel[i] = (uint8_t)lvl;
            if (lvl > maxlvl) maxlvl = lvl;
        }
        col0[i] = (uint8_t)(dual || C == 1 ? cm[0] : (cm[0] | cm[1]));
        col1[i] = (uint8_t)(dual ? cm[1] : C == 1 ? cm[0] : (cm[0] | cm[1]));
        P->task[2 * i] = tk[0];
        P->task[2 * i + 1] = tk[1];
        for (int slot = 0; slot < 2; slot++) {          // level of the active tasks, kept where the sort below reads it cheaply
            const OgTask &t = tk[slot];
            int lv = (t.flags & OG_T_ACTIVE) ? t.level : 0xFF;
            if (lv == 0 && !(t.hm | t.slog) && (t.norm_row == 0xFF || t.norm_row == slot)) {
                // nothing to transform: at most lowband_out = sqrt(N0) x, which the frame kernel writes while staging
                if (t.norm_row == slot) { if (slot) plain1 |= 1u << i; else plain0 |= 1u << i; }
                lv = 0xFF;
            }
            tl[2 * i + slot] = (uint16_t)lv;
        }
    }
    for (int i = 0; i < 24; i++) {
        P->collapse[0][i] = i < 21 ? col0[i] : 0;
        P->collapse[1][i] = i < 21 ? (C == 2 ? col1[i] : col0[i]) : 0;
    }
    P->seed_out = seed;
    P->nops = (uint16_t)nops; P->merge_cnt = (uint16_t)merge_cnt;
    P->nlevels = (uint8_t)maxlvl; P->merge_level = (uint8_t)merge_level;
    P->pad_[0] = P->pad_[1] = 0;
    P->keep[0] = keep0; P->keep[1] = keep1; P->joint = joint; P->plain[0] = plain0; P->plain[1] = plain1;
    // counting sort of the active tasks by level (stable: band order inside a level)
    // (levels above maxlvl hold nothing: the frame kernel reads task_start[0 .. nlevels + 1])
    for (int l = 0; l <= maxlvl + 2; l++) tcur[l] = 0;
    for (int t = 0; t < 42; t++)
        if (tl[t] != 0xFF) tcur[tl[t] + 1]++;
    for (int l = 0; l <= maxlvl + 1; l++) tcur[l + 1] = (uint16_t)(tcur[l + 1] + tcur[l]);
    for (int l = 0; l <= maxlvl + 2; l++) P->task_start[l] = (uint8_t)tcur[l];
    for (int t = 0; t < 42; t++)
        if (tl[t] != 0xFF) P->order[tcur[tl[t]]++] = (uint8_t)t;
    P->pad2_[0] = P->pad2_[1] = P->pad2_[2] = 0;
}

// Scratch shared by the CELT kernels (layout of `work`, include/opusgpu.h), one block per frame so that the layout does
// not move with the launch size: coef[2][960] f32 (K1 -> frame kernel; kept zero between launches), band_cm[48] u32
// ([buffer][band], 24 per buffer; K1 -> plan; kept zero), OgBandPlan (plan -> frame kernel).
#define OG_WORK_STRIDE (1920 * 4 + 48 * 4 + sizeof(OgBandPlan))
#define OG_WORK_BYTES(nframes) ((size_t)(nframes) * OG_WORK_STRIDE)
OG_DEV float *og_work_coef(void *work, uint32_t frame) { return (float *)((char *)work + (size_t)frame * OG_WORK_STRIDE); }
OG_DEV uint32_t *og_work_cm(void *work, uint32_t frame) { return (uint32_t *)((char *)work + (size_t)frame * OG_WORK_STRIDE + 1920 * 4); }
OG_DEV OgBandPlan *og_work_plan(void *work, uint32_t frame)
{
    return (OgBandPlan *)((char *)work + (size_t)frame * OG_WORK_STRIDE + 1920 * 4 + 48 * 4);
}

// The plan of frame f, start to end by one caller (CPU builds): consumes (and re-zeroes) K1's collapse masks.
OG_DEV void og_plan_frame(const OgCeltHdr *h, const OgCeltLeaf *leaves, uint32_t f, const OgCeltConst &K, const uint32_t *la,
                          const uint32_t *lc, void *work)
{
    OgPlanShared S;
    uint32_t *cm = og_work_cm(work, f);
    og_band_plan(h, leaves + h->leaf_begin, K, la, lc, cm, og_work_coef(work, f), og_work_plan(work, f), S);
    for (int i = 0; i < 48; i++) cm[i] = 0;
}

// The same by the one-lane form (og_band_plan_lane); sc: OG_PLAN_SCRATCH cells.  STAGED: cm_staged holds the frame's 48
// collapse-mask words, already fetched (and re-zeroed in `work`) by the caller.
template <bool STAGED = false>
OG_DEV void og_plan_frame_lane(const OgCeltHdr *h, const OgCeltLeaf *leaves, uint32_t f, const OgCeltConst &K, const uint32_t *la,
                               const uint32_t *lc, void *work, OgLaneScratch sc, const uint32_t *cm_staged = nullptr)
{
    if (STAGED) {
        og_band_plan_lane(h, leaves + h->leaf_begin, K, la, lc, cm_staged, og_work_coef(work, f), og_work_plan(work, f), sc);
    } else {
        uint32_t *cm = og_work_cm(work, f);
        og_band_plan_lane(h, leaves + h->leaf_begin, K, la, lc, cm, og_work_coef(work, f), og_work_plan(work, f), sc);
        for (int i = 0; i < 48; i++) cm[i] = 0;
    }
}
