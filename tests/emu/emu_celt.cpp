// TEST FIXTURE: g++ build of the CELT kernel body (opus_amd/csrc/hip/celt_frame.h) in which a workgroup's
// lanes run one after the other (see hip/wg.h).  It lets the CPU-only CI exercise the kernel logic against
// the oracle.  It is not part of libopusgpu.so and nothing under opus_amd/ loads it.
#include <cstring>
#include <vector>
#include "../../opus_amd/csrc/hip/celt_frame.h"
#include "../../opus_amd/csrc/host/celt_tables.hpp"

extern "C" int emu_launch_celt(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                               OgCeltState *state, float *pcm, size_t pcm_stride, const OgCeltDebug *dbg)
{
    static OgCeltTables T;
    static bool init = false;
    if (!init) { og_fill_celt_tables(&T); init = true; }
    static OgCeltLds L;
    static OgTailLds Lt;
    // K1: every PVQ leaf of every frame (order is irrelevant to the result), the band plans, then K2+K3 frame by frame
    // the scratch starts out as garbage (NaN bit patterns) except the collapse-mask words, which the launch contract wants zero
    std::vector<float> work(OG_WORK_BYTES(nframes) / 4 + 4);
    std::memset(work.data(), 0xFF, work.size() * 4);
    for (uint32_t f = 0; f < nframes; f++) std::memset(og_work_cm(work.data(), f), 0, 48 * 4);
    {
        uint32_t total = 0;
        for (uint32_t f = 0; f < nframes; f++) total += hdr[f].nleaves;
        std::vector<OgLeafRef> refs(total + 1);
        uint32_t classes[4];
        const uint32_t nref = opusgpu_sort_pvq_leaves(hdr, nframes, leaves, refs.data(), total + 1, classes);
        for (uint32_t r = 0; r < nref; r++) {           // order is irrelevant to the result
            float x[176];
            if (leaves[refs[r].leaf].n > 24)             // as the kernel: the two large size classes take two zeros per step,
                og_pvq_lane<0>(refs[r], leaves[refs[r].leaf], work.data(), T.pvq_u2, x, 1, dbg ? dbg->pulses : nullptr, nullptr);
            else if (leaves[refs[r].leaf].n <= 8)        // the smallest one probes four table entries per step
                og_pvq_lane<4>(refs[r], leaves[refs[r].leaf], work.data(), T.pvq_u2, x, 1, dbg ? dbg->pulses : nullptr, nullptr);
            else
                og_pvq_lane(refs[r], leaves[refs[r].leaf], work.data(), T.pvq_u2, x, 1, dbg ? dbg->pulses : nullptr, nullptr);
        }
    }
    // the band plan in both its forms (three steps / one lane start to end, celt_plan.h): they must agree on everything the
    // frame kernel reads; the frames then go on with one or the other, alternately
    int plan_mismatches = 0;
    for (uint32_t f = 0; f < nframes; f++) {
        uint32_t cm[48];
        std::memcpy(cm, og_work_cm(work.data(), f), sizeof cm);
        og_plan_frame(hdr + f, leaves, f, T.k, T.lcg_a, T.lcg_c, work.data());
        static OgBandPlan A;
        OgBandPlan *P = og_work_plan(work.data(), f);
        std::memcpy(&A, P, sizeof A);
        std::memset(P, 0xFF, sizeof *P);
        std::memcpy(og_work_cm(work.data(), f), cm, sizeof cm);
        uint16_t sc[OG_PLAN_SCRATCH];
        og_plan_frame_lane(hdr + f, leaves, f, T.k, T.lcg_a, T.lcg_c, work.data(), OgLaneScratch{sc, 1});
        int ntasks = 0;
        for (int t = 0; t < 42; t++) ntasks += (A.task[t].flags & OG_T_ACTIVE) != 0;
        bool same = std::memcmp(A.task, P->task, sizeof A.task) == 0 && std::memcmp(A.collapse, P->collapse, sizeof A.collapse) == 0 &&
                    A.seed_out == P->seed_out && A.nops == P->nops && A.merge_cnt == P->merge_cnt && A.nlevels == P->nlevels &&
                    A.merge_level == P->merge_level && A.plain[0] == P->plain[0] && A.plain[1] == P->plain[1] && A.joint == P->joint &&
                    A.keep[0] == P->keep[0] && A.keep[1] == P->keep[1] &&
                    std::memcmp(A.task_start, P->task_start, A.nlevels + 3) == 0 &&
                    std::memcmp(A.order, P->order, A.task_start[A.nlevels + 2]) == 0 && A.task_start[A.nlevels + 2] <= ntasks &&
                    std::memcmp(A.op, P->op, sizeof(OgFillOp) * A.nops) == 0;
        if (!same) plan_mismatches++;
        if (f & 1) std::memcpy(P, &A, sizeof A);
    }
    for (uint32_t f = 0; f < nframes; f++) {
        std::memset(&L, 0xFF, sizeof L);               // LDS starts undefined on the device
        OgCeltDebug d = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (dbg) {
            d.norm = dbg->norm ? dbg->norm + (size_t)f * 1920 : nullptr;
            d.freq = dbg->freq ? dbg->freq + (size_t)f * 1920 : nullptr;
            d.collapse = dbg->collapse ? dbg->collapse + (size_t)f * 42 : nullptr;
        }
        og_celt_spectrum(hdr + f, og_work_plan(work.data(), f), &T, L, d, og_work_coef(work.data(), f), 0, (int)f);
    }
    std::memset(&Lt, 0xFF, sizeof Lt);
    std::memcpy(Lt.window, T.window, sizeof Lt.window);
    std::memcpy(Lt.taps, T.k.pf_taps, sizeof Lt.taps);
    const uint32_t fpg = frames_per_group ? frames_per_group : 1;
    for (uint32_t f = 0; f < nframes; f += fpg)         // the tail kernel: one workgroup per group, frames in order
        og_celt_tail_group(hdr + f, nframes - f < fpg ? nframes - f : fpg, state, pcm, pcm_stride, Lt, nullptr, work.data(), f);
    return plan_mismatches;
}
extern "C" size_t emu_lds_bytes(void) { return sizeof(OgCeltLds); }
extern "C" size_t emu_tail_lds_bytes(void) { return sizeof(OgTailLds); }

// Diagnostic: band-plan statistics of one frame (K1 collapse masks are not needed for levels / op counts).
extern "C" void emu_plan_stats(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t f, int32_t out[8])
{
    static OgCeltTables T;
    static bool init = false;
    if (!init) { og_fill_celt_tables(&T); init = true; }
    std::vector<float> work(OG_WORK_BYTES(f + 1) / 4 + 4, 0.f);
    og_plan_frame(hdr + f, leaves, f, T.k, T.lcg_a, T.lcg_c, work.data());
    const OgBandPlan *P = og_work_plan(work.data(), f);
    out[0] = P->nlevels; out[1] = P->nops; out[2] = P->merge_level;
    int xf0 = 0, xfn = 0, act = 0;
    for (int t = 0; t < 42; t++) {
        const OgTask &k = P->task[t];
        if (!(k.flags & OG_T_ACTIVE)) continue;
        act++;
        if (k.hm | k.slog) { if (k.level == 0) xf0++; else xfn++; }
    }
    out[3] = act; out[4] = xf0; out[5] = xfn;
    int mc = 0;                                         // most Haar butterflies of any call
    for (int t = 0; t < 42; t++) {
        const OgTask &k = P->task[t];
        const int cnt = (k.rctd & 15) + (k.rctd >> 4);
        if ((k.flags & OG_T_ACTIVE) && k.n != 1 && cnt > mc) mc = cnt;
    }
    out[6] = mc; out[7] = hdr[f].nleaves;
}

// The kernel's cwrsi alone (celt_pvq.h), for known-answer and exhaustive comparisons with the oracle's.
extern "C" uint32_t emu_cwrsi(int n, int k, uint32_t i, int blocks, int32_t *y, int32_t *cm, int probes)
{
    static OgCeltTables T;
    static bool init = false;
    if (!init) { og_fill_celt_tables(&T); init = true; }
    float x[176];
    int16_t p[176];
    int m = 0;
    int logb = 0;
    while ((1 << logb) < blocks) logb++;                // blocks is a power of two
    const uint32_t yy = probes == 0 ? og_cwrsi_sparse(T.pvq_u2, n, k, i, x, 1, logb, &m, p)
                      : probes == 2 ? og_cwrsi(T.pvq_u2, n, k, i, x, 1, logb, &m, p)
                      : probes == 3 ? og_cwrsi_wide<3>(T.pvq_u2, n, k, i, x, 1, logb, &m, p)
                      : probes == 4 ? og_cwrsi_wide<4>(T.pvq_u2, n, k, i, x, 1, logb, &m, p)
                                    : og_cwrsi_wide<2>(T.pvq_u2, n, k, i, x, 1, logb, &m, p);     // (22: the wide form at two probes)
    for (int j = 0; j < n; j++) y[j] = p[j];
    *cm = m;
    return yy;
}
