// C ABI, layer 3b (include/opusgpu.h): raw CELT kernel launch on caller-owned device memory.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <string>
#include "celt_frame.h"
#include "../host/celt_tables.hpp"
#include "device_util.hpp"

// K1: one lane per PVQ leaf, all four size classes in one launch: 256-thread workgroups, each serving one class with as many
// lanes as fit a common LDS budget -- 16 lanes for N <= 176, 32 for N <= 64, 128 for N <= 24, 256 for N <= 8 (12 KB of
// vectors + the 10.6 KB U table); wavefronts without a leaf leave at once.  A wave's vectors are lane-interleaved in LDS
// with the stride of its live lanes (16 / 32 / 64: element i of lane l at x[i * stride + l], one bank per lane,
// conflict-free).  refs are sorted by decreasing size, so a class is a contiguous range, cut into batches of `lanes` leaves.
// OG_K1_BATCHES(c) consecutive batches per workgroup of class c: 1 everywhere.  (Measured, r2n: four batches per workgroup
// for N <= 8 and two for N <= 24 -- table staged once, next batch's records prefetched -- made the kernel 25 % slower, 250 us
// against 201: fewer, longer workgroups balance worse over the CUs than the table staging costs.  Measured again at the
// end of round 2, two batches for the two small classes: 177 us against 160.)
struct OgPvqClasses { uint32_t cnt[4], wg0[4]; };      // leaves per class (largest first); first workgroup of each class
#define OG_K1_BATCHES(c) 1u
#ifndef OG_K1_SMALL_PROBES
#define OG_K1_SMALL_PROBES 4          /* table probes per cwrsi iteration for the leaves of at most 8 bins (celt_pvq.h: og_cwrsi_wide) */
#endif
// Register budget: six wavefronts per SIMD (80 VGPRs, which the kernel meets without spilling now that the rotation works in
// blocks of four coefficients; measured, round 2: 203 us at the compiler's own choice of 110 VGPRs = four per SIMD, 189 us at
// five, 202 us at six while that took 29 spills; with blocks of four: 157 us at six, 160 us at seven = 72 VGPRs with 4 spills).
// The launch is as long as the chains of its largest leaves (tools/k1_classes.sh: the two large classes alone 123 us, the two
// small ones alone 118 us, everything 153-158 us).  Measured without effect on that: s_setprio 3 for the large-class wavefronts
// (153 us either way: they are not short of issue slots), the next rotation block fetched while the current one is rotated
// (7 spills at 80 VGPRs: 162 us).  Four table probes per cwrsi step for N <= 8 (OG_K1_SMALL_PROBES): that class alone 74 -> 66 us.
// The two large size classes: a wavefront holds L = 16 / 32 leaves and the launch is as long as the chain of its largest leaf
// (one lane walks up to 176 bins), so the leaf's other NS = 64 / L lanes -- idle during cwrsi, which is serial -- take part in
// what follows it.  exp_rotation1 with stride s is s independent chains (the residue classes of the position mod s; the
// backward pass of a class touches that class only), and the blocks of a transient frame are independent as well: the
// (block, class) pairs are dealt over the leaf's lanes, each chain run start to end by one lane exactly as og_exp_rotation
// runs it (the same values, bit for bit).  The stride-1 pass of a one-block leaf stays one lane's.  The leaf's values then go
// out in 16-byte pieces dealt over its lanes.
template <int NS>
__device__ void og_coop_rot_pass(float *xl, int xs, int len, int B, int stride, float c, float s, int sub)
{
    const int nf = len - stride, nbk = len - 2 * stride;
    if (nf <= 0) return;
    const int R = stride < nf ? stride : nf;            // classes with at least one step
    const int qf = stride == 1 ? nf : og_idiv_small(nf, stride), remf = nf - qf * stride;
    int qb = 0, remb = 0;
    if (nbk > 0) { qb = stride == 1 ? nbk : og_idiv_small(nbk, stride); remb = nbk - qb * stride; }
    int blk = 0, r = sub;
    while (r >= R && blk < B) { r -= R; blk++; }
    while (blk < B) {
        float *xb = xl + blk * len * xs;
        og_rot_fwd(xb, xs, r, stride, qf + (r < remf), c, s);
        if (r < nbk) {
            const int cnt = qb + (r < remb);
            og_rot_bwd(xb, xs, r + (cnt - 1) * stride, stride, cnt, c, s);
        }
        r += NS;
        while (r >= R && blk < B) { r -= R; blk++; }
    }
}

template <int L>
__device__ __attribute__((noinline)) void og_pvq_coop(const OgLeafRef ref, const OgCeltLeaf lf, bool has, void *work, const uint32_t *U2, float *x, uint32_t tid,
                            int16_t *dbg_pulses, uint64_t *stamps)
{
    constexpr int NS = 64 / L;
    const int owner = (int)tid & (L - 1), sub = (int)tid / L;
    float *xl = x + owner;                              // the leaf's vector: x[j * L + owner]
    const uint32_t frame = ref.frame & OG_REF_FRAME_MASK;
    const int ch = (lf.meta & OG_L_CH_Y) ? 1 : 0, spread = (int)(ref.frame >> OG_REF_SPREAD_SHIFT);
    const bool zeros = lf.k == 0;                       // (see og_pvq_lane)
    uint32_t cm = 0;
    if (has && !zeros) {
        int16_t *dbg = dbg_pulses ? dbg_pulses + (size_t)frame * 1920 + ch * 960 + lf.pos : nullptr;
        cm = og_pvq_leaf<0, false>(U2, lf, spread, xl, L, dbg, stamps);
    }
    // the leaf's lanes learn what they need from its first lane
    const uint32_t w_rot = (has && !zeros) ? (uint32_t)lf.n | (uint32_t)lf.k << 8 | (uint32_t)(lf.meta & OG_L_LOGB_MASK) << 16 | (uint32_t)spread << 20 | 1u << 24 : 0u;
    const uint32_t w_out = has ? (uint32_t)lf.pos | (uint32_t)ch << 16 | (uint32_t)zeros << 17 | (uint32_t)lf.n << 18 | 1u << 26 : 0u;
    const uint32_t rot = (uint32_t)__shfl((int)w_rot, owner), out = (uint32_t)__shfl((int)w_out, owner), fr = (uint32_t)__shfl((int)frame, owner);
    const int len = (int)(rot & 255u), k = (int)((rot >> 8) & 255u), logb = (int)((rot >> 16) & 7u), spr = (int)((rot >> 20) & 3u);
    const bool active = (rot >> 24) != 0 && !(2 * k >= len || spr == 0);
    float c = 0.f, s = 0.f;
    int stride2 = 0;
    const int B = 1 << logb, ln = len >> logb;
    if (active) {                                       // og_exp_rotation's parameters
        const float gain = (float)len * og_frcp((float)(len + (20 - 5 * spr) * k));
        const float theta = 0.5f * gain * gain;
        c = og_cos_halfpi(theta);
        s = og_cos_halfpi(1.f - theta);
        if (len >= 8 * B) {
            stride2 = 1;
            while ((stride2 * stride2 + stride2) * B + (B >> 2) < len) stride2++;
        }
    }
    __syncthreads();                                    // (one wavefront: orders its LDS traffic; the first lanes' vectors are written)
    if (active && stride2) og_coop_rot_pass<NS>(xl, L, ln, B, stride2, s, c, sub);
    __syncthreads();
    if (active) og_coop_rot_pass<NS>(xl, L, ln, B, 1, c, s, sub);
    __syncthreads();
    if (stamps) stamps[2] = (uint64_t)__builtin_readcyclecounter();       // (profiling: [1] .. [2] = scale + rotation, [2] .. [3] = store)
    if (out >> 26) {
        const int pos = (int)(out & 0xFFFFu), n = (int)((out >> 18) & 255u);
        const bool zr = ((out >> 17) & 1u) != 0;
        float *dst = og_work_coef(work, fr) + ((out >> 16) & 1u) * 960 + pos;
#define OG_XV(j) (zr ? 0.f : xl[(j) * L])
        int head = (4 - (pos & 3)) & 3;                 // values in front of the first 16-byte boundary of the row
        if (head > n) head = n;
        const int body = (n - head) >> 2, tail = (n - head) & 3;
        if (sub == 0) {
            for (int i = 0; i < head; i++) dst[i] = OG_XV(i);
            for (int i = head + 4 * body; i < n; i++) dst[i] = OG_XV(i);
        }
        for (int q = sub; q < body; q += NS) {
            const int i = head + 4 * q;
            *(OgF4s *)(dst + i) = OgF4s{OG_XV(i), OG_XV(i + 1), OG_XV(i + 2), OG_XV(i + 3)};
        }
        (void)tail;
#undef OG_XV
    }
    if (has && !zeros) OG_ATOMIC_OR(og_work_cm(work, frame) + ch * 24 + lf.band, cm);
    if (stamps) stamps[3] = (uint64_t)__builtin_readcyclecounter();
}

template <bool DBG>                   // DBG: stage dumps / profiling stamps wanted (the plain instantiation carries none of that code)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void og_pvq_all_kernel(const OgLeafRef *__restrict__ refs, OgPvqClasses cl,
                                                         const OgCeltLeaf *__restrict__ leaves,
                                                         uint32_t nframes, const OgCeltTables *__restrict__ T, void *work,
                                                         int16_t *dbg_pulses_, uint64_t *dbg_cycles_)
{
    int16_t *const dbg_pulses = DBG ? dbg_pulses_ : nullptr;
    uint64_t *const dbg_cycles = DBG ? dbg_cycles_ : nullptr;
    __shared__ __attribute__((aligned(16))) uint32_t U2[15 * OG_U2_STRIDE + 1];      // staged in 16-byte pieces
    __shared__ float x[3072];                           // 176 * 16 = 2816, 64 * 32 = 2048, 24 * 128 = 3072, 8 * 256 = 2048
    // Workgroup order: the two large classes keep one 16 / 32-lane wavefront busy for ~100 us each while holding a full
    // workgroup's LDS, so they are dealt one in three among the small-class workgroups from the start of the grid (every CU
    // then runs a mix) instead of all first.  cl.wg0[] is the sequential numbering; `b` below is the position in it.
    const uint32_t tid = threadIdx.x;
    uint32_t b = blockIdx.x;
    {
        const uint32_t nbig = cl.wg0[2], nall = gridDim.x;
        if (nall - nbig >= 2 * nbig) {
            if (b < 3 * nbig) b = (b % 3 == 0) ? b / 3 : nbig + (b - b / 3 - 1);
            // else: b already counts nbig large + (b - nbig) small ones before it
        }
    }
    const int c = b >= cl.wg0[3] ? 3 : b >= cl.wg0[2] ? 2 : b >= cl.wg0[1] ? 1 : 0;
    const uint32_t lanes = c == 0 ? 16u : c == 1 ? 32u : c == 2 ? 128u : 256u;
    if (tid >= ((lanes + 63u) & ~63u)) return;          // whole wavefronts with nothing to do (no barrier after this point for them)
    uint32_t base = 0;
    for (int k = 0; k < c; k++) base += cl.cnt[k];
    const uint32_t nbat = OG_K1_BATCHES(c), cnt = cl.cnt[c];
    uint32_t g = (b - cl.wg0[c]) * nbat * lanes + tid;  // this lane's leaf of the workgroup's first batch
    const bool live = tid < lanes;
    OgLeafRef ref = {0, 0};
    OgCeltLeaf lf = {};
    if (live && g < cnt) { ref = refs[base + g]; lf = leaves[ref.leaf]; }
    const uint32_t nw = (lanes + 63u) >> 6;             // wavefronts that stay: they share the table
    {
        static_assert((15 * OG_U2_STRIDE + 1) % 4 == 0 && offsetof(OgCeltTables, pvq_u2) % 16 == 0, "the table as 16-byte pieces");
        const uint4 *src = (const uint4 *)T->pvq_u2;
        for (uint32_t p = tid; p < (15 * OG_U2_STRIDE + 1) / 4; p += 64 * nw) ((uint4 *)U2)[p] = src[p];
    }
    __syncthreads();                                    // wavefronts that have ended no longer take part in the barrier
    if (c <= 1 && OG_K1_BATCHES(c) == 1u) {             // the large classes: one wavefront, every lane stays (og_pvq_coop)
        uint64_t *st0 = dbg_cycles && tid == 0 && (b - cl.wg0[c]) < 4096 ? dbg_cycles + (size_t)nframes * 64 + ((size_t)(c == 0 ? 176 : 64) * 4096 + (b - cl.wg0[c])) * 4 : nullptr;
        if (c == 0) og_pvq_coop<16>(ref, lf, live && g < cnt, work, U2, x, tid, dbg_pulses, st0);
        else og_pvq_coop<32>(ref, lf, live && g < cnt, work, U2, x, tid, dbg_pulses, st0);
        return;
    }
    if (!live) return;
    const int nmax = c == 0 ? 176 : c == 1 ? 64 : c == 2 ? 24 : 8;
    uint64_t *stamps = dbg_cycles && tid == 0 && (b - cl.wg0[c]) < 4096 ? dbg_cycles + (size_t)nframes * 64 + ((size_t)nmax * 4096 + (b - cl.wg0[c])) * 4 : nullptr;
    for (uint32_t r = 0; r < nbat && g < cnt; r++, g += lanes) {       // (no barrier inside: the lanes are independent)
        const OgLeafRef cur = ref;
        const OgCeltLeaf clf = lf;
        if (r + 1 < nbat && g + lanes < cnt) { ref = refs[base + g + lanes]; lf = leaves[ref.leaf]; }     // in flight during this batch
        switch (c) {
        case 0: og_pvq_lane<0>(cur, clf, work, U2, x + tid, 16, dbg_pulses, stamps); break;      // (0: the walk for sparse leaves)
        case 1: og_pvq_lane<0>(cur, clf, work, U2, x + tid, 32, dbg_pulses, stamps); break;
        case 2: og_pvq_lane(cur, clf, work, U2, x + (tid >> 6) * (24 * 64) + (tid & 63), 64, dbg_pulses, stamps); break;
        default: og_pvq_lane<OG_K1_SMALL_PROBES>(cur, clf, work, U2, x + (tid >> 6) * (8 * 64) + (tid & 63), 64, dbg_pulses, stamps); break;
        }
        stamps = nullptr;
    }
}

// Band plan (celt_plan.h) of OG_PLAN_FPW frames per 128-lane workgroup, in three steps: a work item per (frame, band) for what
// the bands do not share, a lane per frame for the recurrences over the bands (the only serial part: the other wavefronts
// wait at the barrier), the (frame, band) work items again for the records out.  A work item keeps its band's two task
// records in registers from the first step to the last.  What the steps read and index sits in LDS: the head and the band
// records of the workgroup's headers (93 of a header's 188 dwords), the small constant tables, an OgPlanShared per frame:
// 18.2 KB, eight workgroups per CU.
// (Round 2, one lane per frame for everything -- og_plan_lane_kernel below: 94 us per 32 768 frames, 76 us per 4 096, the
// length of one lane's instruction chain.  The split cuts that chain to the recurrences: 74 and 47 us.  What it took to win at
// the large size as well: all 2 048 workgroups resident at once, i.e. the 18 KB of LDS and two wavefronts per workgroup;
// with the task records in LDS -- 31 KB, five workgroups per CU -- it took 111 us; a 64-lane workgroup 5 us more than 128.)
#define OG_PLAN_FPW 16
#define OG_PLAN_WG 128                /* two wavefronts: eight workgroups per CU are then four wavefronts per SIMD (109 VGPRs) */
#define OG_PLAN_ITEMS ((OG_PLAN_FPW * 21 + OG_PLAN_WG - 1) / OG_PLAN_WG)      /* (frame, band) work items per lane */
#define OG_PLAN_HW 93                 // dwords kept of a header: head (9) + band records (21 x 4)
__global__ __launch_bounds__(OG_PLAN_WG) __attribute__((amdgpu_waves_per_eu(4, 8))) void og_plan_kernel(const OgCeltHdr *__restrict__ hdr, const OgCeltLeaf *__restrict__ leaves,
                                                             uint32_t nframes, const OgCeltTables *__restrict__ T, void *work,
                                                             int stop /* profiling aid: leave after step `stop` (1 = A, 2 = B) */)
{
    static_assert(sizeof(OgHdrHead) == 36 && sizeof(OgCeltBand) == 16 && offsetof(OgCeltHdr, band) % 4 == 0 && OG_MAX_BANDS == 21, "header pieces as dwords");
    __shared__ uint32_t hs[OG_PLAN_FPW * OG_PLAN_HW];
    __shared__ OgCeltConst K;
    __shared__ uint32_t la[184], lc[184];
    __shared__ OgPlanShared S[OG_PLAN_FPW];
    const uint32_t tid = threadIdx.x;
    const uint32_t f0 = blockIdx.x * OG_PLAN_FPW, nf = nframes - f0 < OG_PLAN_FPW ? nframes - f0 : OG_PLAN_FPW;
    constexpr uint32_t BOFF = offsetof(OgCeltHdr, band) / 4;
    {
        // every fetch first (one memory round trip for all of them), then the LDS stores
        constexpr int NH = (OG_PLAN_FPW * OG_PLAN_HW + OG_PLAN_WG - 1) / OG_PLAN_WG, NK = (sizeof(OgCeltConst) / 4 + OG_PLAN_WG - 1) / OG_PLAN_WG;
        static_assert(2 * OG_PLAN_WG >= 184, "the LCG maps in two pieces per lane");
        uint32_t hv[NH], kv[NK], av[2], cv[2];
#pragma unroll
        for (int r = 0; r < NH; r++) {
            const uint32_t p = tid + r * OG_PLAN_WG, q = p < nf * OG_PLAN_HW ? p : 0u;
            const uint32_t fr = q / OG_PLAN_HW, j = q - fr * OG_PLAN_HW;
            hv[r] = ((const uint32_t *)(hdr + f0 + fr))[j < 9 ? j : BOFF + (j - 9)];
        }
#pragma unroll
        for (int r = 0; r < NK; r++) { const uint32_t p = tid + r * OG_PLAN_WG; kv[r] = ((const uint32_t *)&T->k)[p < sizeof(OgCeltConst) / 4 ? p : 0u]; }
#pragma unroll
        for (int r = 0; r < 2; r++) { const uint32_t p = tid + r * OG_PLAN_WG; av[r] = T->lcg_a[p < 184u ? p : 0u]; cv[r] = T->lcg_c[p < 184u ? p : 0u]; }
#pragma unroll
        for (int r = 0; r < NH; r++) { const uint32_t p = tid + r * OG_PLAN_WG; if (p < nf * OG_PLAN_HW) hs[p] = hv[r]; }
#pragma unroll
        for (int r = 0; r < NK; r++) { const uint32_t p = tid + r * OG_PLAN_WG; if (p < sizeof(OgCeltConst) / 4) ((uint32_t *)&K)[p] = kv[r]; }
#pragma unroll
        for (int r = 0; r < 2; r++) { const uint32_t p = tid + r * OG_PLAN_WG; if (p < 184u) { la[p] = av[r]; lc[p] = cv[r]; } }
        for (uint32_t p = tid; p < nf * (sizeof(OgPlanShared) / 4); p += OG_PLAN_WG) ((uint32_t *)&S[0])[p] = 0;
    }
    __syncthreads();
    if (stop == 3) return;            // (profiling: the fetches alone)
    OgTask tk[OG_PLAN_ITEMS][2];      // the work items' task records
#pragma unroll
    for (int r = 0; r < OG_PLAN_ITEMS; r++) {
        const uint32_t p = tid + r * OG_PLAN_WG;
        if (p < nf * 21) {
            const uint32_t fr = p / 21, i = p - fr * 21;
            const OgHdrHead &h = *(const OgHdrHead *)(hs + fr * OG_PLAN_HW);
            og_plan_band_a(h, (const OgCeltBand *)(hs + fr * OG_PLAN_HW + 9), leaves + h.leaf_begin, K, og_work_cm(work, f0 + fr),
                           og_work_coef(work, f0 + fr), (int)i, S[fr], tk[r]);
        }
    }
    if (stop == 1) return;
    __syncthreads();
    // the frames' chains are dealt over both wavefronts: a wavefront runs the union of its lanes' paths through the bands, so
    // eight lanes each instead of sixteen in one (measured: 68.7 -> 65.0 us; eight frames per workgroup instead, so that four
    // lanes share a wavefront: 91 us in two rounds at four wavefronts per SIMD, 94 us at eight with 94 spilled registers)
    static_assert(OG_PLAN_WG == 128 && OG_PLAN_FPW % 2 == 0, "two wavefronts");
    const uint32_t cf = (tid & 63u) < OG_PLAN_FPW / 2 ? (tid >> 6) * (OG_PLAN_FPW / 2) + (tid & 63u) : 0xFFFFu;
    if (cf < nf) {
        const OgHdrHead &h = *(const OgHdrHead *)(hs + cf * OG_PLAN_HW);
        og_plan_chain(h, (const OgCeltBand *)(hs + cf * OG_PLAN_HW + 9), leaves + h.leaf_begin, K, la, lc, og_work_plan(work, f0 + cf), S[cf]);
    }
    if (stop == 2) return;
    __syncthreads();
    // the tasks' order by level (og_plan_sort), a lane per task: per level in use one vote of the wavefront gives the level's
    // tasks (in band order: the lane order) and their count; each wavefront sorts half of the workgroup's frames
    {
        constexpr uint32_t H = OG_PLAN_FPW / 2;
        const uint32_t fr0 = (tid >> 6) * H, lane = tid & 63u;
        uint32_t lv_[H], mx_[H];
#pragma unroll
        for (uint32_t k = 0; k < H; k++) {              // every fetch first (frames past the launch's end: nothing to sort)
            const bool in = fr0 + k < nf;
            const OgPlanShared &Sf = S[in ? fr0 + k : 0u];
            lv_[k] = in && lane < 42u ? Sf.tl[lane] : 0xFFu;
            mx_[k] = in ? Sf.maxlvl : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < H; k++) {
            if (fr0 + k >= nf) break;                   // (uniform over the wavefront)
            OgPlanShared &Sf = S[fr0 + k];
            const uint32_t lvl = lv_[k];
            const int maxlvl = __builtin_amdgcn_readfirstlane((int)mx_[k]);
            uint32_t n = 0, pos = 0, ts = 0;
            for (int l = 0; l <= maxlvl; l++) {
                const uint64_t m = __ballot(lvl == (uint32_t)l);
                if (lane == (uint32_t)l) ts = n;
                if (lvl == (uint32_t)l) pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                n += (uint32_t)__popcll(m);
            }
            if (lvl != 0xFFu) Sf.order[pos] = (uint8_t)lane;
            if (lane <= (uint32_t)maxlvl + 2u) Sf.task_start[lane] = (uint8_t)(lane <= (uint32_t)maxlvl ? ts : n);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OG_PLAN_ITEMS; r++) {
        const uint32_t p = tid + r * OG_PLAN_WG;
        if (p < nf * 21) {
            const uint32_t fr = p / 21, i = p - fr * 21;
            const OgHdrHead &h = *(const OgHdrHead *)(hs + fr * OG_PLAN_HW);
            og_plan_emit(h, (const OgCeltBand *)(hs + fr * OG_PLAN_HW + 9), (int)i, tk[r], og_work_plan(work, f0 + fr), S[fr]);
        }
    }
}

#define OG_PLAN_LANE_FPW 16
// The same plan by one lane per frame start to end (og_band_plan_lane), OG_PLAN_LANE_FPW frames per wavefront: the form the
// three steps replaced, kept as their cross-check (the CPU emulation runs both on every frame) and behind OPUSGPU_PLAN=lane.
// The per-lane code is one long serial chain, so what it reads and indexes sits in LDS: the workgroup's headers (copied in
// coalesced), the small constant tables, and a row of scratch cells per lane for the arrays it indexes at run time.
__global__ __launch_bounds__(64) void og_plan_lane_kernel(const OgCeltHdr *__restrict__ hdr, const OgCeltLeaf *__restrict__ leaves,
                                                     uint32_t nframes, const OgCeltTables *__restrict__ T, void *work)
{
    static_assert(sizeof(OgCeltHdr) % 4 == 0, "headers are copied as dwords");
    __shared__ uint32_t hs[OG_PLAN_LANE_FPW * sizeof(OgCeltHdr) / 4];
    __shared__ OgCeltConst K;
    __shared__ uint32_t la[184], lc[184];
    __shared__ uint16_t sc[OG_PLAN_SCRATCH * OG_PLAN_LANE_FPW];
    const uint32_t f0 = blockIdx.x * OG_PLAN_LANE_FPW, nf = nframes - f0 < OG_PLAN_LANE_FPW ? nframes - f0 : OG_PLAN_LANE_FPW;
    const uint32_t *src = (const uint32_t *)(hdr + f0);
    for (uint32_t p = threadIdx.x; p < nf * (sizeof(OgCeltHdr) / 4); p += 64) hs[p] = src[p];
    for (int p = threadIdx.x; p < (int)(sizeof(OgCeltConst) / 4); p += 64) ((uint32_t *)&K)[p] = ((const uint32_t *)&T->k)[p];
    for (int p = threadIdx.x; p < 184; p += 64) { la[p] = T->lcg_a[p]; lc[p] = T->lcg_c[p]; }
    __syncthreads();
    // K1's collapse masks (48 words per frame): fetched coalesced, handed back zeroed, and kept over the band energies (energy, prev_min) of
    // the LDS header copies, which the plan does not read (no extra LDS: 2048 workgroups stay resident at once)
    static_assert(offsetof(OgCeltHdr, energy) % 4 == 0 && offsetof(OgCeltHdr, band) - offsetof(OgCeltHdr, energy) >= 48 * 4, "room for the masks (over energy + prev_min)");
    constexpr uint32_t HW = sizeof(OgCeltHdr) / 4, EOFF = offsetof(OgCeltHdr, energy) / 4;
    for (uint32_t p = threadIdx.x; p < nf * 48; p += 64) {
        const uint32_t fr = p / 48, j = p - fr * 48;
        uint32_t *cm = og_work_cm(work, f0 + fr);
        hs[fr * HW + EOFF + j] = cm[j];
        cm[j] = 0;
    }
    __syncthreads();
    const uint32_t f = f0 + threadIdx.x;
    if (threadIdx.x < OG_PLAN_LANE_FPW && f < nframes)
        og_plan_frame_lane<true>((const OgCeltHdr *)hs + threadIdx.x, leaves, f, K, la, lc, work, OgLaneScratch{sc + threadIdx.x, OG_PLAN_LANE_FPW},
                      hs + threadIdx.x * HW + EOFF);
}

// Frame kernels.  og_celt_kernel: one 256-lane workgroup per frame, stateless (band walk .. IMDCT post-twiddle), eight
// workgroups per CU (20.4 KB of LDS, 64 VGPRs).  og_celt_tail_kernel: one 128-lane workgroup per group of
// frames_per_group consecutive records (a stream's frames in time order), which it walks in order carrying the stream's
// state: window + overlap-add, comb filter, de-emphasis, PCM.
__global__ __launch_bounds__(OG_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void og_celt_kernel(const OgCeltHdr *__restrict__ hdr, uint32_t nframes,
                                                                                                   const OgCeltTables *__restrict__ T, void *work,
                                                                                                   OgCeltDebug dbg, int stop)
{
    __shared__ OgCeltLds L;
    // (one frame per workgroup and a grid of exactly nframes workgroups: no bound check, so no wait for that argument.  A
    // workgroup looping over frames -- a launch sized to what the chip holds at once -- was measured in round 2: 582 us against
    // 376, the loop costs the 64-register budget of eight workgroups per CU dearly)
    const uint32_t f = blockIdx.x;
    OgCeltDebug d;
    d.pulses = nullptr;
    d.norm = dbg.norm ? dbg.norm + (size_t)f * 1920 : nullptr;
    d.freq = dbg.freq ? dbg.freq + (size_t)f * 1920 : nullptr;
    d.collapse = dbg.collapse ? dbg.collapse + (size_t)f * 42 : nullptr;
    d.cycles = dbg.cycles ? dbg.cycles + (size_t)f * 64 : nullptr;
    og_celt_spectrum(hdr + f, og_work_plan(work, f), T, L, d, og_work_coef(work, f), stop, (int)f);
}

// Groups: frames_per_group consecutive records each, or -- group_off != null -- records [group_off[g], group_off[g + 1]) for
// workgroup g (a stream's records of a mixed send: their number differs from stream to stream).
__global__ __launch_bounds__(OG_WGT) void og_celt_tail_kernel(const OgCeltHdr *__restrict__ hdr, uint32_t nframes, uint32_t frames_per_group,
                                                              const uint32_t *__restrict__ group_off,
                                                              const OgCeltTables *__restrict__ T, void *work, OgCeltState *state, float *pcm,
                                                              size_t pcm_stride, OgCeltDebug dbg)
{
    __shared__ OgTailLds L;
    for (int p = threadIdx.x; p < 120; p += OG_WGT) L.window[p] = T->window[p];
    if (threadIdx.x < 12) (&L.taps[0][0])[threadIdx.x] = (&T->k.pf_taps[0][0])[threadIdx.x];
    __syncthreads();
    uint32_t f0, nfr;
    if (group_off) {
        f0 = group_off[blockIdx.x];
        const uint32_t f1 = group_off[blockIdx.x + 1];
        if (f0 >= nframes || f1 <= f0) return;
        nfr = (f1 < nframes ? f1 : nframes) - f0;
    } else {
        f0 = blockIdx.x * frames_per_group;
        if (f0 >= nframes) return;
        nfr = nframes - f0 < frames_per_group ? nframes - f0 : frames_per_group;
    }
    og_celt_tail_group(hdr + f0, nfr, state, pcm, pcm_stride, L, dbg.cycles ? dbg.cycles + (size_t)f0 * 64 : nullptr, work, f0);
}

// Scratch of opusgpu_launch_celt_noalloc (include/opusgpu.h): the `work` buffer of a launch of up to OG_NOALLOC_MAX_FRAMES
// frames, held by every GPU the library runs on until the process ends (7.6 MiB).  In host code the array's NAME is only a
// stand-in for the symbol: its device address on a given GPU comes from hipGetSymbolAddress (opusgpu_init, once per device,
// kept in DevCtx::scratch) and that pointer is what every kernel argument and memset below uses.
#define OG_NOALLOC_MAX_FRAMES 256
static __device__ __align__(256) unsigned char og_noalloc_scratch[OG_WORK_BYTES(OG_NOALLOC_MAX_FRAMES)];

namespace og {

thread_local std::string g_last_error;

int device_fail(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return OPUSGPU_DEVICE;
}

// Per-device context, built once: the constant tables.
struct DevCtx {
    OgCeltTables *tables = nullptr;
    void *scratch = nullptr;            // this device's copy of og_noalloc_scratch (hipGetSymbolAddress)
};
static std::mutex g_tables_mu;
static DevCtx g_ctx[64];

static int ctx_for_device(int device, DevCtx **out)
{
    if (device < 0 || device >= 64) return OPUSGPU_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    DevCtx &c = g_ctx[device];
    if (!c.tables) {
        int prev = 0;
        OG_HIP(hipGetDevice(&prev));
        OG_HIP(hipSetDevice(device));
        OgCeltTables *host = new OgCeltTables, *dev = nullptr;
        og_fill_celt_tables(host);
        hipError_t e = hipMalloc(&dev, sizeof(OgCeltTables));
        if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(OgCeltTables), hipMemcpyHostToDevice);
        delete host;
        void *scr = nullptr;                             // each GPU holds its own copy of the array at its own address
        if (e == hipSuccess) e = hipGetSymbolAddress(&scr, HIP_SYMBOL(og_noalloc_scratch));
        (void)hipSetDevice(prev);
        if (e != hipSuccess) { if (dev) (void)hipFree(dev); return device_fail(e, "celt device set-up"); }
        c.scratch = scr;
        c.tables = dev;
    }
    *out = &c;
    return OPUSGPU_OK;
}

int celt_tables_for_device(int device, const OgCeltTables **out)
{
    DevCtx *c;
    int rc = ctx_for_device(device, &c);
    if (rc) return rc;
    *out = c->tables;
    return OPUSGPU_OK;
}

}  // namespace og

extern "C" {

const char *opusgpu_last_error(void) { return og::g_last_error.c_str(); }

int opusgpu_init(int device)
{
    const OgCeltTables *t;
    return og::celt_tables_for_device(device, &t);
}

size_t opusgpu_celt_work_bytes(uint32_t nframes) { return OG_WORK_BYTES(nframes); }

int opusgpu_launch_celt_pvq(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, const OgLeafRef *refs,
                            const uint32_t class_counts[4], void *work, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!hdr || !leaves || !work || !class_counts) return OPUSGPU_BAD_ARG;
    const uint32_t nrefs = class_counts[0] + class_counts[1] + class_counts[2] + class_counts[3];
    if (nrefs && !refs) return OPUSGPU_BAD_ARG;
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    og::DevCtx *ctx;
    int rc = og::ctx_for_device(device, &ctx);
    if (rc) return rc;
    const OgCeltTables *T = ctx->tables;
    OgCeltDebug dbg = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (debug) dbg = *debug;
    hipStream_t st = (hipStream_t)stream;
    if (nrefs) {                                        // K1 (skipped when no frame carries a PVQ leaf)
        OgPvqClasses cl;
        static const uint32_t lanes[4] = {16, 32, 128, 256};
        uint32_t wg = 0;
        // profiling aid: OPUSGPU_K1_CLASSES = bit mask of the size classes to run (the output is then incomplete)
        static const int only = getenv("OPUSGPU_K1_CLASSES") ? atoi(getenv("OPUSGPU_K1_CLASSES")) : 15;
        for (int c = 0; c < 4; c++) {
            cl.cnt[c] = class_counts[c];
            cl.wg0[c] = wg;
            const uint32_t per_wg = lanes[c] * OG_K1_BATCHES(c);
            if ((only >> c) & 1) wg += (class_counts[c] + per_wg - 1) / per_wg;
        }
        if (wg && (dbg.pulses || dbg.cycles)) hipLaunchKernelGGL(og_pvq_all_kernel<true>, dim3(wg), dim3(256), 0, st, refs, cl, leaves, nframes, T, work, dbg.pulses, dbg.cycles);
        else if (wg) hipLaunchKernelGGL(og_pvq_all_kernel<false>, dim3(wg), dim3(256), 0, st, refs, cl, leaves, nframes, T, work, nullptr, nullptr);
    }
    // OPUSGPU_PLAN=lane: the one-lane form instead (profiling: see og_plan_lane_kernel)
    static const bool plan_lane = [] { const char *e = getenv("OPUSGPU_PLAN"); return e && e[0] == 'l'; }();
    if (plan_lane)
        hipLaunchKernelGGL(og_plan_lane_kernel, dim3((nframes + OG_PLAN_LANE_FPW - 1) / OG_PLAN_LANE_FPW), dim3(64), 0, st, hdr, leaves, nframes, T, work);
    else
    {
        static const int plan_stop = getenv("OPUSGPU_PLAN_STOP") ? atoi(getenv("OPUSGPU_PLAN_STOP")) : 0;       // profiling aid
        hipLaunchKernelGGL(og_plan_kernel, dim3((nframes + OG_PLAN_FPW - 1) / OG_PLAN_FPW), dim3(OG_PLAN_WG), 0, st, hdr, leaves, nframes, T, work, plan_stop);
    }
    OG_HIP(hipGetLastError());
    return OPUSGPU_OK;
}

int opusgpu_launch_celt_spectrum(const OgCeltHdr *hdr, uint32_t nframes, void *work, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!hdr || !work) return OPUSGPU_BAD_ARG;
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    const OgCeltTables *T;
    int rc = og::celt_tables_for_device(device, &T);
    if (rc) return rc;
    OgCeltDebug dbg = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (debug) dbg = *debug;
    static const int stop = getenv("OPUSGPU_STOP_AFTER") ? atoi(getenv("OPUSGPU_STOP_AFTER")) : 0;      // profiling aid (tools/stage_insts.sh)
    hipLaunchKernelGGL(og_celt_kernel, dim3(nframes), dim3(OG_WG), 0, (hipStream_t)stream, hdr, nframes, T, work, dbg, stop);
    OG_HIP(hipGetLastError());
    return OPUSGPU_OK;
}

static int launch_tail(const OgCeltHdr *hdr, uint32_t nframes, uint32_t frames_per_group, const uint32_t *group_off, uint32_t ngroups,
                       void *work, OgCeltState *state, float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!hdr || !state || !pcm || !work || (group_off ? ngroups == 0 : frames_per_group == 0)) return OPUSGPU_BAD_ARG;
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    const OgCeltTables *T;
    int rc = og::celt_tables_for_device(device, &T);
    if (rc) return rc;
    OgCeltDebug dbg = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (debug) dbg = *debug;
    const uint32_t groups = group_off ? ngroups : (nframes + frames_per_group - 1) / frames_per_group;
    hipLaunchKernelGGL(og_celt_tail_kernel, dim3(groups), dim3(OG_WGT), 0, (hipStream_t)stream, hdr, nframes, frames_per_group, group_off, T, work,
                       state, pcm, pcm_stride_floats, dbg);
    OG_HIP(hipGetLastError());
    return OPUSGPU_OK;
}

int opusgpu_launch_celt_tail(const OgCeltHdr *hdr, uint32_t nframes, uint32_t frames_per_group, void *work, OgCeltState *state,
                             float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    return launch_tail(hdr, nframes, frames_per_group, nullptr, 0, work, state, pcm, pcm_stride_floats, debug, stream);
}

int opusgpu_launch_celt_tail_runs(const OgCeltHdr *hdr, uint32_t nframes, const uint32_t *run_off, uint32_t nruns, void *work,
                                  OgCeltState *state, float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes && !run_off) return OPUSGPU_BAD_ARG;
    return launch_tail(hdr, nframes, 0, run_off, nruns, work, state, pcm, pcm_stride_floats, debug, stream);
}

int opusgpu_launch_celt_runs(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, const uint32_t *run_off, uint32_t nruns,
                             const OgLeafRef *refs, const uint32_t class_counts[4], void *work, OgCeltState *state, float *pcm,
                             size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!state || !pcm || !run_off || nruns == 0) return OPUSGPU_BAD_ARG;
    int rc = opusgpu_launch_celt_pvq(hdr, leaves, nframes, refs, class_counts, work, debug, stream);
    if (rc) return rc;
    if ((rc = opusgpu_launch_celt_spectrum(hdr, nframes, work, debug, stream))) return rc;
    return opusgpu_launch_celt_tail_runs(hdr, nframes, run_off, nruns, work, state, pcm, pcm_stride_floats, debug, stream);
}

int opusgpu_launch_celt_frames(const OgCeltHdr *hdr, uint32_t nframes, uint32_t frames_per_group, void *work, OgCeltState *state,
                               float *pcm, size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!hdr || !state || !pcm || !work || frames_per_group == 0) return OPUSGPU_BAD_ARG;
    int rc = opusgpu_launch_celt_spectrum(hdr, nframes, work, debug, stream);
    if (rc) return rc;
    return opusgpu_launch_celt_tail(hdr, nframes, frames_per_group, work, state, pcm, pcm_stride_floats, debug, stream);
}

int opusgpu_launch_celt(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                        const OgLeafRef *refs, const uint32_t class_counts[4], void *work, OgCeltState *state, float *pcm,
                        size_t pcm_stride_floats, const OgCeltDebug *debug, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (!state || !pcm || frames_per_group == 0) return OPUSGPU_BAD_ARG;
    int rc = opusgpu_launch_celt_pvq(hdr, leaves, nframes, refs, class_counts, work, debug, stream);
    if (rc) return rc;
    return opusgpu_launch_celt_frames(hdr, nframes, frames_per_group, work, state, pcm, pcm_stride_floats, debug, stream);
}

uint32_t opusgpu_noalloc_max_frames(void) { return OG_NOALLOC_MAX_FRAMES; }

int opusgpu_launch_celt_noalloc(const OgCeltHdr *hdr, const OgCeltLeaf *leaves, uint32_t nframes, uint32_t frames_per_group,
                                const OgLeafRef *refs, const uint32_t class_counts[4], OgCeltState *state, float *pcm,
                                size_t pcm_stride_floats, void *stream)
{
    if (nframes == 0) return OPUSGPU_OK;
    if (nframes > OG_NOALLOC_MAX_FRAMES) {
        og::g_last_error = "opusgpu_launch_celt_noalloc: " + std::to_string(nframes) + " frames exceed the static scratch (" +
                           std::to_string(OG_NOALLOC_MAX_FRAMES) + " frames per call); split the batch or use opusgpu_launch_celt";
        return OPUSGPU_BAD_ARG;
    }
    int device = 0;
    OG_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64) return OPUSGPU_BAD_ARG;
    void *scratch = nullptr;
    {
        std::lock_guard<std::mutex> lk(og::g_tables_mu);
        scratch = og::g_ctx[device].tables ? og::g_ctx[device].scratch : nullptr;
    }
    if (!scratch) {                                     // set-up allocates (the tables): it is opusgpu_init's job, not this call's
        og::g_last_error = "opusgpu_launch_celt_noalloc: call opusgpu_init(device) once before the first launch on a device";
        return OPUSGPU_BAD_ARG;
    }
    // the array keeps the last call's contents; the kernels hand the parts that must start at zero back zeroed, but a call
    // that was cut short (a failed launch) would not have: clear this call's part first
    OG_HIP(hipMemsetAsync(scratch, 0, OG_WORK_BYTES(nframes), (hipStream_t)stream));
    return opusgpu_launch_celt(hdr, leaves, nframes, frames_per_group, refs, class_counts, scratch, state, pcm, pcm_stride_floats,
                               nullptr, stream);
}

}  // extern "C"
