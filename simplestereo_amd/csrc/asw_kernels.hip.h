// K1: ASW cost aggregation + winner-take-all keys, fused (gfx950).
//
// Replaces the reference hot loop _passive.cpp:34-100 (left-referenced) and, by
// the symmetry of the aggregated cost, also the right-referenced loop
// _passive.cpp:191-253:
//     C(y, xl, xr) = sum_t wL[y,xl,t] wR[y,xr,t] TAD(L[t+xl], R[t+xr]) / sum_t wL wR
// is ONE number used by both passes, so a single evaluation feeds two argmins
// (left: over xr for fixed xl; right: over xl for fixed xr).
//
// Work decomposition
//   workgroup  = (image row y, tile of Tx left columns, chunk of Dc disparities)
//   thread     = register tile of RX=8 columns x RD=4 disparities, 2 fp32 accumulators
//                (N, S' -- see below) per (x,d) pair (RX=4 instantiation for small disparity ranges)
//   outer loop = the window rows i (tap row r = y - pad + i).  Per window row the
//                workgroup stages the pixels it needs in LDS (prefetched one row ahead) and
//                builds, in LDS,
//                  wL[j][x]   left support weights  (x in tile, tap column j)
//                  wR[j][xr]  right support weights (xr = x - d over the tile: they
//                             do not depend on x, the reference re-evaluates them
//                             for every (x,d), _passive.cpp:71-74)
//                  e[u][d]    truncated absolute difference of L[r][u], R[r][u-d]
//                             as bytes: it depends on the tap column u = x+j-pad
//                             only, so each value serves up to `win` taps
//   inner loop = tap columns j; per step a thread reads 8 wL + 12 wR values (five
//                ds_read_b128, 16-byte lane stride thanks to the parity-split rows) and
//                ONE new dword of e (the other seven rows of its window slide in
//                registers), then does 32 taps x {v_mul, 2 v_fma} + 4 x {cvt_ubyte, sub}.
// HBM traffic is the pixel records only (16 B/pixel/image, re-read from L2 by
// neighbouring tiles) plus 8-byte WTA keys; everything else lives in LDS/VGPRs.
// Measured (1080p, D 0..192, win 35): 45.6-46.4 ms, 3.5e10 VALU wave-instructions at ~83 % of the
// plain fp32 issue rate, no scratch, LDS ~50 % busy (profiles/).
#pragma once
#include "common.hip.h"

namespace ssamd {

static constexpr int ASW_RX = 8;      // columns per thread of the default register tile (template parameter RX: 8 or 4)
static constexpr int ASW_RD = 4;      // disparities per thread (one packed dword of e per row)
static constexpr int ASW_WB = 4;       // support weights evaluated side by side in the build phase
static constexpr int ASW_MAX_THREADS = 768;
// right weights read per tap column (float4 granules): the RX + RD - 1 centres x - d of the register tile
__host__ __device__ constexpr int asw_nwr(int rx) { return (rx + ASW_RD - 1 + 3) / 4 * 4; }
static_assert(ASW_RD == 4 && asw_nwr(8) == 12 && asw_nwr(4) == 8, "the main loop is unrolled for 8x4 and 4x4 register tiles");

struct AswGeom {
    int Tx, XG, DG, Dc, nchunks, threads;
    int Rx;                      // columns per thread: 8, or 4 for small disparity ranges (twice the threads per column)
    int nL, nR, nRc, SR, Se, emask;
    int SL, hL, hR;              // wL row stride and the half offsets of the parity-split wL / wR rows
    int wseg, wlen;              // weight build: tap columns (of a chunk) split in wseg segments of wlen
    int JC;                      // tap columns staged per chunk (multiple of ASW_RX); >= win: one chunk
    int e2;                      // 1: two e tiles (rows alternate): no barrier between the last chunk of a window row and
                                 //    the e / weight build of the next one (chunked form with >= 2 chunks only)
    int e_bytes;                 // size of one e tile
    int pipe;                    // 1: asw_aggregate_pipe_kernel (asw_pipe_kernel.hip.h): phase-shifted build / aggregation
    int NC, JCmax;               //    chunks per window row (tail shorter than 8 merged into the last) and rows per weight buffer
    int dephase;                 //    1: waves 0-3 build before they aggregate, the others after (0: all after)
    int wave_rx;                 // 8 or 4: asw_aggregate_wave_kernel (asw_wave_kernel.hip.h) with that many columns per lane runs
                                 //    instead (small disparity ranges); the other fields then describe the fallback geometry
    int off_wL, off_wR, off_e, off_labL, off_labR, off_bgrL, off_bgrR, off_bestL, off_bestR, off_cen, off_prox;
    int lds_bytes;
    int lds_bytes_evol;          // phase-shifted kernel with the pre-computed TAD volume: LDS without the staged colour bytes (bgrL / bgrR are last)
};

// exact mode (round 6): the aggregation kernels select the NEAR-TIES of their winners themselves, in the epilogue, and append them
// to a queue that the fp64 tie-break pass (asw_exact_kernels.hip.h) re-evaluates in the reference's arithmetic -- no cost-image
// volume in HBM (round 5 dumped H*W*nD*4 bytes and re-read them: 1.6 GB at 1080p / 193, 9.1 GB at 4K).  entries == nullptr: off.
struct AswExactQueue {
    u64 *entries;                // pix (32) | d (16) << 32 | sides (2) << 48; pix = (output row - row0) * W + LEFT column
    uint32_t *ekeys;             // RAW queue only: the fp32 cost image of each entry (what asw_exact_filter_kernel compares with the final winners)
    unsigned int *counter;       // entries appended (may exceed cap)
    unsigned char *flagL, *flagR;    // [rows][W] the pixel has near-ties (flagR null: no right-referenced pass; both null: RAW queue)
    unsigned char *zeroL, *zeroR;    // [rows][W] the (left / right) pixel has TWO OR MORE candidates whose fp32 cost is exactly 0: asw_exact_zero_kernel
                                     //   (a lone zero -- the true match of a synthetic pair -- needs nothing; zeroR null without a right pass)
    unsigned char *zrow;             // [rows] the row holds such a pixel: only those rows are looked at
    unsigned int W;                  // image width (row of a pixel index)
    unsigned int cap;
    uint32_t tol;                // cost-image ulps
    float sat_abs;               // absolute cost difference below which two saturated candidates are a near-tie of the reference's fp64
    uint32_t deep;               // RAW queue: images at or above this (40 - cost <= sat_abs: every tap with a weight saturated) are NOT queued --
                                 //   they can only tie a winner within 2 sat_abs of 40, and such pixels get ALL their candidates from
                                 //   asw_exact_escalate_kernel.  0xffffffff (direct calls): every valid candidate is tested.
};
// Two uses.  DIRECT calls (one disparity chunk, no right-referenced pass: every pixel is decided by ONE workgroup, whose tile-local
// winner IS the final one): the kernels append to the final queue and flag the pixels.  MERGING calls (several chunks, consistent):
// a workgroup only knows its tile-local winners, which are >= the final ones, so what it selects is a SUPERSET -- it goes, with its
// cost image, to a RAW queue that asw_exact_filter_kernel (asw_exact_kernels.hip.h) re-tests against the final winners once the
// aggregation kernel is done.  (Measured, config 3 consistent, before `deep`: 8.3e6 raw candidates, 99.4 % of them
// with 40 - cost == 0 exactly -- on the bench frame 6 ... 29 % of ALL candidates have every tap saturated, and in a tile that does not
// hold a right pixel's match they all tie the tile's winner.  With `deep` + escalation: profiles/r06_exact_mode_cost.txt.)

struct AswArgs {
    const PixRec *recL, *recR;   // [H][W] pixel records of the (sub-)image
    const float *prox;           // [win*win] proximity weights exp(-|t|/gammaP)
    u64 *keyL;                   // [rows][W] left-referenced WTA keys  (cost, d)
    u64 *keyR;                   // [rows][W] right-referenced WTA keys (cost, xl) or nullptr
    int16_t *disp;               // non-null: ONE disparity chunk and no right pass -- every pixel is decided by exactly one
                                 //   workgroup, which writes the disparity itself (no keys, no atomics, no decode kernel)
    const unsigned char *evol;   // phase-shifted kernel: truncated-absolute-difference volume of asw_tad_volume_kernel (or nullptr:
                                 //   e tiles are built in the kernel), [disparity chunk][image row - erow0][evolW columns][g.Se bytes]
    int erow0, erows, evolW;
    float *costs;                // optional [rows][W][nD] raw cost dump
    int cost_keys;               // 1: the dump holds the 32-bit cost images of asw_cost_key (bit patterns) instead of the costs: the
                                 //    fp64 tie-break pass compares them with the winning keys (asw_exact_kernels.hip.h)
    int H, W, win, pad, minD, maxD, row0, rows;
    int ystep;                   // output row of workgroup row b: row0 + b * ystep (2: alternate-rows mode)
    int yb0;                     // first workgroup row of this launch (the half-width tail launch continues the main launch's rows)
    int yskip_at, yskip;         // ... + yskip for b >= yskip_at: TWO row ranges in one launch (the border rows of a row strip whose
                                 //     interior rows ran while the halo was in flight, strips.py); yskip = 0: one range
    float kC;                    // -log2(e)/gammaC
    AswExactQueue xq;            // exact mode: near-tie queue (entries == nullptr: off)
    AswGeom g;
};

// ---- the phase-shifted kernel's tiles of the headline configurations as COMPILE-TIME geometry (round 6) ------------------------
// asw_aggregate_pipe_kernel<.., SLC, SRC, SEC> used to take only its three LDS strides as constants and read the other ~30 geometry
// values from the kernel arguments: with the pointers and image sizes that is more scalars than the 102 SGPRs hold, and the compiler
// spilled 76-155 of them into VGPR lanes (v_writelane / v_readlane).  The static instantiations now derive EVERYTHING from
// (winSize, XG, DG, JC) at compile time -- a constexpr restatement of the pipe branch of asw_layout_e (ssamd_api.hip), which the
// launcher compares field by field with the geometry it planned before it picks a static instantiation (asw_pipe_geom_matches).
struct AswPipeTileId { int win, XG, DG, JC; };
__host__ __device__ constexpr int asw_cx_round_up(int v, int m) { return (v + m - 1) / m * m; }
__host__ __device__ constexpr int asw_cx_max(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr AswGeom asw_pipe_geom_constexpr(AswPipeTileId t)
{
    AswGeom g{};
    const int win = t.win, p = win / 2;
    g.Rx = 8; g.JC = t.JC; g.pipe = 1; g.wave_rx = 0; g.e2 = 1; g.nchunks = 1;
    g.NC = (win + t.JC / 2) / t.JC;
    g.JCmax = asw_cx_max(t.JC, win - (g.NC - 1) * t.JC);
    g.XG = t.XG; g.DG = t.DG;
    g.Tx = 8 * t.XG; g.Dc = ASW_RD * t.DG;
    g.threads = asw_cx_round_up(t.XG * t.DG, 64);
    g.dephase = g.threads / 64 >= 12 ? 1 : 0;
    g.nL = g.Tx + 2 * p;
    g.nRc = g.Tx + g.Dc - 1;
    g.nR = g.nRc + 2 * p;
    g.hL = 0; g.hR = 0;
    g.SL = asw_cx_round_up(g.Tx, 4);
    g.SR = asw_cx_round_up(g.nRc + 1, 4);
    g.Se = 16 * ((t.DG + 3) / 4);
    g.emask = 0;
    const int wrows = 2 * g.JCmax;
    int off = 0;
    g.off_wL = off; off = (off + wrows * g.SL * 4 + 15) & ~15;
    g.off_wR = off; off = (off + wrows * g.SR * 4 + 15) & ~15;
    g.e_bytes = (g.nL * g.Se + 15) & ~15;
    g.off_e = off; off = (off + g.e_bytes * 2 + 15) & ~15;
    g.off_labL = off; off = (off + g.nL * 16 * 2 + 15) & ~15;
    g.off_labR = off; off = (off + g.nR * 16 * 2 + 15) & ~15;
    g.off_bestL = off; off = (off + g.Tx * 8 + 15) & ~15;
    g.off_bestR = off; off = (off + (g.nRc + 1) * 8 + 15) & ~15;
    g.off_cen = off; off = (off + (g.Tx + g.nRc) * 16 + 15) & ~15;
    g.off_prox = off; off = (off + win * 4 * 2 + 15) & ~15;
    g.lds_bytes_evol = off;
    g.off_bgrL = off; off = (off + g.nL * 4 * 2 + 15) & ~15;
    g.off_bgrR = off; off = (off + g.nR * 4 * 2 + 15) & ~15;
    g.lds_bytes = off;
    return g;
}
// tile of a static instantiation, by its three strides: 120 x 196 (1080p / D 0..192), 88 x 260 (4096 x 2160 / D 0..256), 216 x 68 (D 0..64)
template <int SLC, int SRC, int SEC> struct AswPipeTile { static constexpr AswPipeTileId id{35, 1, 1, 16}; };        // (generic instantiation: never read)
template <> struct AswPipeTile<120, 316, 208> { static constexpr AswPipeTileId id{35, 15, 49, 16}; };
template <> struct AswPipeTile<88, 348, 272> { static constexpr AswPipeTileId id{35, 11, 65, 16}; };
template <> struct AswPipeTile<216, 284, 80> { static constexpr AswPipeTileId id{35, 27, 17, 16}; };
// every field the kernel reads (and the launch uses) equal?
inline bool asw_pipe_geom_matches(const AswGeom &a, const AswGeom &b)
{
    return a.pipe == b.pipe && a.Rx == b.Rx && a.JC == b.JC && a.NC == b.NC && a.JCmax == b.JCmax && a.XG == b.XG && a.DG == b.DG && a.Tx == b.Tx &&
           a.Dc == b.Dc && a.threads == b.threads && a.dephase == b.dephase && a.nL == b.nL && a.nRc == b.nRc && a.nR == b.nR && a.SL == b.SL &&
           a.SR == b.SR && a.Se == b.Se && a.e_bytes == b.e_bytes && a.off_wL == b.off_wL && a.off_wR == b.off_wR && a.off_e == b.off_e &&
           a.off_labL == b.off_labL && a.off_labR == b.off_labR && a.off_bestL == b.off_bestL && a.off_bestR == b.off_bestR &&
           a.off_cen == b.off_cen && a.off_prox == b.off_prox && a.off_bgrL == b.off_bgrL && a.off_bgrR == b.off_bgrR &&
           a.lds_bytes == b.lds_bytes && a.lds_bytes_evol == b.lds_bytes_evol && a.e2 == b.e2;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// output row (of the sub-image) of workgroup row b
template <typename Args>
__device__ __forceinline__ int asw_out_row(const Args &A, int b)
{
    b += A.yb0;
    return A.row0 + b * A.ystep + (b >= A.yskip_at ? A.yskip : 0);
}

// Matching-cost cap of the reference (std::min(40, ...), _passive.cpp:77).
static constexpr float ASW_TAD_CAP = 40.0f;

// Each (x,d) pair accumulates TWO weighted sums over the window taps,
//     N  = sum w * e            and      S' = sum w * (40 - e),      w = wL*wR,
// instead of the reference's (sum w*e, sum w).  N + S' = 40 * sum w, so
//     cost = 40 N / (N + S')     and     40 - cost = 40 S' / (N + S').
// fp32 keeps full RELATIVE precision on whichever of N, S' is small: costs near 0
// and costs near the truncation value 40 (where the reference's candidates differ
// by 1e-6 and less: occlusions, textureless areas) are both resolved, which a
// fp32 (sum w*e)/(sum w) cannot do.  The WTA key is built from whichever form is
// accurate (asw_cost_key).
struct AswRow {                 // one e row of the register window: e and 40-e for the thread's disparities
    float e[ASW_RD], c[ASW_RD];
};

__device__ __forceinline__ void asw_row_unpack(AswRow &row, const uint32_t packed)
{
#pragma unroll
    for (int di = 0; di < ASW_RD; ++di) {
        const float e = (float)((packed >> (8 * di)) & 0xffu);   // v_cvt_f32_ubyteN
        row.e[di] = e;
        row.c[di] = ASW_TAD_CAP - e;
    }
}

// RX*RD taps of one tap column.  ROT: window slot of the thread's first column (slots rotate
// by one per tap column; the rotation is resolved at compile time by unrolling RX columns).
template <int RX, int ROT>
__device__ __forceinline__ void asw_taps(float (&accN)[RX][ASW_RD], float (&accS)[RX][ASW_RD],
                                         const float (&wl)[RX], const float (&wr)[asw_nwr(RX)],
                                         const AswRow (&win)[RX])
{
#pragma unroll
    for (int xi = 0; xi < RX; ++xi) {
        const AswRow &row = win[(ROT + xi) % RX];
#pragma unroll
        for (int di = 0; di < ASW_RD; ++di) {
            const float w = wl[xi] * wr[xi - di + ASW_RD - 1];
            accN[xi][di] = fmaf(w, row.e[di], accN[xi][di]);
            accS[xi][di] = fmaf(w, row.c[di], accS[xi][di]);
        }
    }
}

// Order-preserving 32-bit image of the aggregated cost of one (x,d) pair, and the
// cost itself.  cost <= 20: bits(cost); cost > 20: 0xC0000000 - bits(40 - cost), which
// is > bits(20.0f) and decreasing in (40 - cost): a monotone map of the cost that keeps
// the resolution of the accurate operand.
__device__ __forceinline__ uint32_t asw_cost_key(const float n, const float s, float &cost)
{
    const float t40 = n + s;
    if (n <= s) {
        cost = ASW_TAD_CAP * n / t40;
        return __float_as_uint(cost);
    }
    const float inv = ASW_TAD_CAP * s / t40;
    cost = ASW_TAD_CAP - inv;
    return 0xC0000000u - __float_as_uint(inv);
}

// ---- exact mode: near-tie selection (shared by the four kernel families) ------------------------------------------------
static constexpr unsigned EXACT_SIDE_L = 1u, EXACT_SIDE_R = 2u;
// hint bit of a queue entry: the candidate's fp32 cost image is 0 (N = 0 exactly).  Black margins of rectified frames tie thousands
// of such candidates per row; the fp64 cost of one is EXACTLY 0 iff every in-image tap has TAD = 0 (weights are positive in
// fp64), which asw_exact_eval_kernel checks with integer compares before it spends ~2 500 fp64 exp / sqrt / div on the entry
static constexpr unsigned EXACT_HINT_ZERO = 4u;
// cost images (asw_cost_key) at or above this hold 40 - cost (cost > 20)
static constexpr uint32_t EXACT_KEY_HIGH = 0xC0000000u - 0x41A00000u;      // 0x41A00000 = bits of 20.0f

// is candidate image `key` a near-tie of the better image `kb` (kb <= key)?  (a) within `tol` ulps of it -- or (b), both on the
// saturated side of the image (cost > 20: the image holds 40 - cost), within `sat_abs` in ABSOLUTE terms: the (N, S') pair
// resolves 40 - 1e-30 from 40 - 0, but the reference's fp64 quotient carries a rounding noise of up to ~(win^2) ulps of 40
// (2e-11 for a 35 x 35 window), so among candidates closer than that its first minimum is decided by that noise and has to be
// recomputed -- or (c) the two lie either side of cost = 20, where the image changes form and ulps do not compare.
// MONOTONE in kb: near(key, kb) implies near(key, kb') for every kb <= kb' <= key, except where (c) held for kb and kb' is on
// the other side of 20 -- exact_near_local below closes that gap, so a workgroup may test against its tile-local winner
// (>= the final one) and queue a superset.
__device__ __forceinline__ bool exact_near(uint32_t key, uint32_t kb, uint32_t tol, float sat_abs)
{
    if (key - kb <= tol) return true;
    if (key >= EXACT_KEY_HIGH) {
        const float inv = __uint_as_float(0xC0000000u - key);                                   // 40 - cost of the candidate
        if (kb >= EXACT_KEY_HIGH) return __uint_as_float(0xC0000000u - kb) - inv <= sat_abs;
        return (40.0f - inv) - __uint_as_float(kb) <= 20.0f * 1.1920929e-7f * (float)tol;
    }
    return false;
}

// ... against a winner kb' that may still be displaced by a better one: also true when both images are saturated-side and the
// candidate is inside the band above cost 20 that case (c) spans (a final winner just below 20 would make it a near-tie)
__device__ __forceinline__ bool exact_near_local(uint32_t key, uint32_t kb, uint32_t tol, float sat_abs)
{
    if (key - kb <= tol) return true;
    // quick reject (the common case of the epilogue scan: a wrong candidate, cost > 20, of a pixel whose winner is well below 20):
    // cases (b), (c) and the band all need the winner within 20 * 2^-23 * tol <= 2.4 (tol <= 1e6) of cost = 20 or above it
    if (key < EXACT_KEY_HIGH || kb < 0x418C0000u) return false;          // 0x418C0000 = bits of 17.5f
    if (exact_near(key, kb, tol, sat_abs)) return true;
    return kb >= EXACT_KEY_HIGH && __uint_as_float(0xC0000000u - key) >= 20.0f - 20.0f * 1.1920929e-7f * (float)tol;
}

__device__ __forceinline__ u64 exact_entry(uint32_t pix, int d, unsigned sides)
{
    return (u64)pix | ((u64)(uint32_t)d << 32) | ((u64)sides << 48);
}

// Append (pix, d, sides) for the lanes that `want` it.  Wave-aggregated: ONE atomicAdd on the queue counter per wave and call
// (a flat or saturated frame makes every candidate a near-tie: per-lane atomics on one address would serialise the whole grid).
// Every lane that reaches the call takes part; lanes that left the kernel earlier are simply not in the ballot.
__device__ __forceinline__ void asw_exact_push_wave(const AswExactQueue &q, bool want, uint32_t pix, int d, unsigned sides, uint32_t key = 0xffffffffu)
{
    if (key == 0u) sides |= EXACT_HINT_ZERO;                  // (key unknown = 0xffffffff: no hint)
    const u64 mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0) return;
    const int leader = (int)__builtin_ctzll(mask);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(q.counter, (unsigned)__builtin_popcountll(mask));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    if (want) {
        const unsigned slot = base + (unsigned)__builtin_popcountll(mask & (((u64)1 << lane) - 1));
        if (slot < q.cap) {
            q.entries[slot] = exact_entry(pix, d, sides);
            if (q.ekeys) q.ekeys[slot] = key;
        }
        if ((sides & EXACT_SIDE_L) && q.flagL) q.flagL[pix] = 1;
        if ((sides & EXACT_SIDE_R) && q.flagR) q.flagR[pix - (uint32_t)d] = 1;
    }
}

// Epilogue step 1 (after the barrier that completes the tile-local winners bL / bR in LDS): every candidate of the thread's
// register tile that is a near-tie of its pixel's LOCAL winner -- and is not that winner -- is queued.  Local winners are >= the
// final ones, so this is a superset of the near-ties of the final winners (exact_near_local).  kk = the cost images of the register
// tile as the first epilogue pass computed them (0xffffffff: not a candidate the reference evaluates).
//   bL = &bestL[first column of the thread], bR = &bestR[slot of (first column, first disparity + RD - 1)] or nullptr,
//   xb / db = first column / disparity of the tile, rowpix = (output row - row0) * W.
// `live` = the lane holds candidates; EVERY lane of the wave calls this (wave-aggregated queue slots).
template <int RX, int RD> struct AswKeyTile { uint32_t v[RX][RD]; };
// (A __noinline__ form -- own register allocation, the 32 cost images by value -- was tried when the scan pushed three of the
//  phase-shifted kernel's VGPRs into scratch: the by-value tile travels THROUGH scratch, + 0.9 ms per 1080p launch.  Inlined it is.)
template <int RX, int RD>
__device__ __forceinline__ void asw_exact_select(const AswExactQueue &q, bool live, const AswKeyTile<RX, RD> &kt,
                                                 const u64 *bL, const u64 *bR, int xb, int db, uint32_t rowpix)
{
    const uint32_t (&kk)[RX][RD] = kt.v;
    static_assert(RX * RD <= 32, "one bit per candidate of the register tile");
    // Branch-free per candidate (a first form with short-circuit tests compiled to ~300 exec-mask branches per thread, a second one
    // with a per-candidate slow path cost + 1 ms on a 37 ms launch: a pixel without a good match has most of its candidates on
    // that path).  Rule (a) of exact_near for every candidate; rules (b), (c) and the local band only for the columns whose local
    // winner costs 17.5 or more -- one branch per column, skipped by waves that hold no such column.  Invalid candidates carry
    // 0xffffffff and are masked out, and so are, in a merging call, candidates at or above q.deep (see AswExactQueue) -- and candidates
    // whose image is 0 (cost exactly 0: they can only tie a winner that costs 0 too; the black margins of a rectified 1080p frame are
    // 3e7 of them; asw_exact_zero_kernel settles those pixels with integer compares).
    uint32_t mL = 0, mR = 0;
    if (live) {
        uint32_t zl = 0, zr = 0;
        const uint32_t tol = q.tol, deep = q.deep;
        const float ctol = 20.0f * 1.1920929e-7f * (float)tol, sat_abs = q.sat_abs;
        auto slow = [&](uint32_t key, uint32_t kh) -> uint32_t {             // exact_near_local without its rule (a), as 0 / 1
            const float inv = __uint_as_float(0xC0000000u - key);
            const uint32_t khigh = kh >= EXACT_KEY_HIGH ? 1u : 0u;
            const uint32_t b_ = khigh & (__uint_as_float(0xC0000000u - kh) - inv <= sat_abs ? 1u : 0u);
            const uint32_t c_ = (khigh ^ 1u) & ((40.0f - inv) - __uint_as_float(kh) <= ctol ? 1u : 0u);
            const uint32_t band = khigh & (inv >= 20.0f - ctol ? 1u : 0u);
            return (key >= EXACT_KEY_HIGH ? 1u : 0u) & (b_ | c_ | band);
        };
#pragma unroll
        for (int xi = 0; xi < RX; ++xi) {
            const u64 kl = bL[xi];
            const uint32_t kh = (uint32_t)(kl >> 32);
            const uint32_t wd = (uint32_t)kl - (uint32_t)db;                    // winner's disparity relative to the tile (== di for the winner)
            uint32_t okm = 0;
#pragma unroll
            for (int di = 0; di < RD; ++di) {
                const uint32_t key = kk[xi][di];
                const uint32_t ok = (key < deep ? 1u : 0u) & (key != 0u ? 1u : 0u) & (wd != (uint32_t)di ? 1u : 0u);
                okm |= ok << di;
                mL |= (ok & (key - kh <= tol ? 1u : 0u)) << (xi * RD + di);
                zl |= ((key == 0u ? 1u : 0u) & (wd != (uint32_t)di ? 1u : 0u)) << xi;       // a second candidate that costs exactly 0
            }
            if (kh >= 0x418C0000u && kh != 0xffffffffu) {                       // 17.5f
#pragma unroll
                for (int di = 0; di < RD; ++di) mL |= (((okm >> di) & 1u) & slow(kk[xi][di], kh)) << (xi * RD + di);
            }
        }
        if (bR) {
#pragma unroll
            for (int k = 0; k < RX + RD - 1; ++k) {
                const u64 kr = bR[k];
                const uint32_t kh = (uint32_t)(kr >> 32);
                const uint32_t wx = (uint32_t)kr - (uint32_t)xb;                // winner's column relative to the tile (== xi for the winner)
                const bool high = kh >= 0x418C0000u && kh != 0xffffffffu;
#pragma unroll
                for (int xi = 0; xi < RX; ++xi) {
                    const int di = xi + RD - 1 - k;
                    if (di >= 0 && di < RD) {                                   // (compile time)
                        const uint32_t key = kk[xi][di];
                        const uint32_t ok = (key < deep ? 1u : 0u) & (key != 0u ? 1u : 0u) & (wx != (uint32_t)xi ? 1u : 0u);
                        uint32_t near = key - kh <= tol ? 1u : 0u;
                        if (high) near |= slow(key, kh);
                        mR |= (ok & near) << (xi * RD + di);
                        zr |= ((key == 0u ? 1u : 0u) & (wx != (uint32_t)xi ? 1u : 0u)) << k;
                    }
                }
            }
        }
        // pixels with two or more zero-cost candidates (black margins): marked for asw_exact_zero_kernel; rare, a few byte stores
        if (zl | zr) {
            q.zrow[rowpix / q.W] = 1;
            for (uint32_t z = zl; z; z &= z - 1) q.zeroL[rowpix + (uint32_t)(xb + __builtin_ctz(z))] = 1;
            for (uint32_t z = zr; z; z &= z - 1) q.zeroR[rowpix + (uint32_t)(xb - db - RD + 1 + __builtin_ctz(z))] = 1;      // right column of slot k: xb + xi - (db + di), k = xi - di + RD - 1
        }
    }
    // Queue slots: ONE returning atomic per wave (a first form took one per round of the busiest lane: the round trips, at the end
    // of a workgroup with nothing to hide them, cost consistent = True 1.7 ms at 1080p).  Per-lane counts n <= 32 are summed over
    // the lanes below by bit plane: six ballots.
    const uint32_t m = mL | mR;
    const uint32_t n = (uint32_t)__builtin_popcount(m);
    if (__builtin_amdgcn_ballot_w64(n != 0) == 0) return;
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const u64 pl = __builtin_amdgcn_ballot_w64((n >> bit) & 1u);
        before += __builtin_amdgcn_mbcnt_hi((uint32_t)(pl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pl, 0u)) << bit;
        total += (uint32_t)__builtin_popcountll(pl) << bit;
    }
    const u64 active = __builtin_amdgcn_ballot_w64(true);
    const int leader = (int)__builtin_ctzll(active);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(q.counter, total);
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    unsigned slot = base + before;
    uint32_t mm = m;
    while (mm) {
        const int b = __builtin_ctz(mm);
        mm &= mm - 1;
        const int xi = b / RD, di = b - xi * RD;
        const unsigned sides = ((mL >> b) & 1u ? EXACT_SIDE_L : 0u) | ((mR >> b) & 1u ? EXACT_SIDE_R : 0u);
        const uint32_t pix = rowpix + (uint32_t)(xb + xi);
        const int d = db + di;
        if (slot < q.cap) {
            uint32_t key = 0;
#pragma unroll
            for (int a = 0; a < RX; ++a)
#pragma unroll
                for (int c2 = 0; c2 < RD; ++c2)
                    if (a * RD + c2 == b) key = kk[a][c2];
            q.entries[slot] = exact_entry(pix, d, sides | (key == 0u ? EXACT_HINT_ZERO : 0u));
            if (q.ekeys) q.ekeys[slot] = key;
        }
        if ((sides & EXACT_SIDE_L) && q.flagL) q.flagL[pix] = 1;
        if ((sides & EXACT_SIDE_R) && q.flagR) q.flagR[pix - (uint32_t)d] = 1;
        ++slot;
    }
}

// Epilogue step 2: `mine` (a tile-local winner) has just been merged into the pixel's global key with old = atomicMin(&key, mine).
// Whichever of the two loses is queued if it is a near-tie of the other: atomics on one address are totally ordered, every
// local winner meets the running minimum exactly once, and the running minimum is >= the final one -- so every local winner
// that is a near-tie of the FINAL winner is queued, by its own workgroup or by the one that displaced it; the final winner
// itself never is (asw_exact_winners_kernel adds it for flagged pixels).  RIGHT: keys of a right pixel xcol, low word = left column.
template <bool RIGHT>
__device__ __forceinline__ void asw_exact_merge(const AswExactQueue &q, bool have, u64 mine, u64 old, uint32_t rowpix, int xcol)
{
    bool want = false;
    uint32_t pix = 0, key = 0;
    int d = 0;
    if (have && old != KEY_NONE) {
        const u64 lo = mine < old ? mine : old, hi = mine < old ? old : mine;
        if ((uint32_t)(hi >> 32) < q.deep && (uint32_t)(hi >> 32) != 0u && exact_near_local((uint32_t)(hi >> 32), (uint32_t)(lo >> 32), q.tol, q.sat_abs)) {
            want = true;
            key = (uint32_t)(hi >> 32);
            if (RIGHT) { const int xl = (int)(uint32_t)hi; pix = rowpix + (uint32_t)xl; d = xl - xcol; }
            else { pix = rowpix + (uint32_t)xcol; d = (int)(uint32_t)hi; }
        }
    }
    if (have && old != KEY_NONE && (uint32_t)(old >> 32) == 0u && (uint32_t)(mine >> 32) == 0u) {
        // two tiles' winners both cost exactly 0: the pixel has two zero-cost candidates (asw_exact_zero_kernel)
        (RIGHT ? q.zeroR : q.zeroL)[rowpix + (uint32_t)xcol] = 1;
        q.zrow[rowpix / q.W] = 1;
    }
    asw_exact_push_wave(q, want, pix, d, RIGHT ? EXACT_SIDE_R : EXACT_SIDE_L, key);
}

// wL / wR rows are stored with their even and odd 16-byte blocks in two halves ("parity split"):
// element c lives at  ((c >> 2) & 1) * half + ((c >> 3) << 2) + (c & 3).  A thread reads RX = 8
// consecutive columns = one even + one odd block, so for each ds_read_b128 the lanes of a wave
// (consecutive column groups) are 16 bytes apart instead of 32: no 2-way bank conflicts.
__device__ __forceinline__ int asw_split_pos(int c, int half)
{
    return ((c >> 2) & 1) * half + ((c >> 3) << 2) + (c & 3);
}

// e tile addressing: rows of Se bytes (Se = 4 * power of two >= DG), one dword (RD = 4 disparities)
// per disparity group; the dword slot of group dg in row ul is XOR-swizzled with (ul / RX) so that
// the lanes of a wave (consecutive xg, rows RX apart) read distinct banks.
template <int RX>
__device__ __forceinline__ int asw_e_offset(int ul, int slot, int Se, int emask)
{
    return ul * Se + ((slot ^ ((ul / RX) & emask)) << 2);
}

// CHUNKED: the tap columns of a window row are staged g.JC at a time (see the loop over jc below).
// RX: columns per thread.  8 is the throughput tile (168 VGPRs, 3 waves per SIMD).  4 halves the columns and
// the accumulators per thread: twice the threads per tile column and 4 waves per SIMD, for small disparity
// ranges where few threads share a weight row and LDS capacity, not VGPRs, limits the resident waves.
template <bool WITH_COSTS, bool CHUNKED, int RX = ASW_RX>
__global__ __launch_bounds__(ASW_MAX_THREADS, RX == 8 ? 3 : 4) void asw_aggregate_kernel(const AswArgs A)
{
    constexpr int NWR = asw_nwr(RX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AswGeom &g = A.g;
    float *const wL = reinterpret_cast<float *>(smem + g.off_wL);
    float *const wR = reinterpret_cast<float *>(smem + g.off_wR);
    unsigned char *const eT0 = reinterpret_cast<unsigned char *>(smem + g.off_e);
    float4 *const labL = reinterpret_cast<float4 *>(smem + g.off_labL);
    float4 *const labR = reinterpret_cast<float4 *>(smem + g.off_labR);
    uint32_t *const bgrL = reinterpret_cast<uint32_t *>(smem + g.off_bgrL);
    uint32_t *const bgrR = reinterpret_cast<uint32_t *>(smem + g.off_bgrR);
    u64 *const bestL = reinterpret_cast<u64 *>(smem + g.off_bestL);
    u64 *const bestR = reinterpret_cast<u64 *>(smem + g.off_bestR);
    float4 *const cenLab = reinterpret_cast<float4 *>(smem + g.off_cen);
    float *const proxS = reinterpret_cast<float *>(smem + g.off_prox);

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = A.W, win = A.win, p = A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nR = g.nR, nRc = g.nRc, SR = g.SR, Se = g.Se, emask = g.emask;
    // XCD-aware tile order: workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, a
    // speed assumption only).  When the row has a multiple of 8 x tiles, tile slot b of every row lands on
    // XCD b % 8; give each XCD a run of ADJACENT tiles (slots b, b+8, ... -> tiles m*(b%8) + b/8, m = tiles/8)
    // so that the halo columns neighbouring tiles share are served by one L2 instead of two.
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int x0 = bx * Tx;
    const int y = asw_out_row(A, blockIdx.y);
    const int dlo = A.minD + blockIdx.z * Dc;
    const int dhi = dlo + Dc - 1;
    // no (x,d) pair of this tile has x-d >= 0 (left image border): nothing to aggregate; the empty candidate
    // loop of the reference leaves dBest = 0, i.e. the output x (_passive.cpp:54,98)
    if (min(x0 + Tx - 1, W - 1) - dlo < 0) {
        if (A.disp)
            for (int k = threadIdx.x; k < Tx && x0 + k < W; k += blockDim.x)
                A.disp[(size_t)(y - A.row0) * W + x0 + k] = (int16_t)(x0 + k);
        return;
    }

    const int segL_lo = x0 - p;        // first tap column staged from the left image
    const int xrc_lo = x0 - dhi;       // first right-image window centre of the tile
    const int segR_lo = xrc_lo - p;    // first tap column staged from the right image

    // lanes of a wave run along x (xg fastest): their wL / wR reads are consecutive 16-byte
    // slots (conflict-free ds_read_b128 for any lane grouping)

    float accN[RX][ASW_RD], accS[RX][ASW_RD];
#pragma unroll
    for (int a = 0; a < RX; ++a)
#pragma unroll
        for (int b = 0; b < ASW_RD; ++b) { accN[a][b] = 0.f; accS[a][b] = 0.f; }

    for (int k = tid; k < Tx; k += nthr) bestL[k] = KEY_NONE;
    for (int k = tid; k <= nRc; k += nthr) bestR[k] = KEY_NONE;

    // window centres (row y) of the tile: left columns x0.., then right columns xrc_lo..
    for (int c = tid; c < Tx + nRc; c += nthr) {
        const bool isL = c < Tx;
        const int ccol = isL ? x0 + c : xrc_lo + (c - Tx);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ccol < (unsigned)W) {
            const PixRec q = (isL ? A.recL : A.recR)[(size_t)y * W + ccol];
            v = make_float4(q.L, q.a, q.b, 1.f);
        }
        cenLab[c] = v;
    }
    // e-build task walk: task t -> (ul = t % nL, dq = t / nL), advanced incrementally

    const int i_lo = max(0, p - y), i_hi = min(win, A.H + p - y);
    // stage the pixels of image row r this tile touches into staging buffer `buf`
    // (coalesced 16 B loads; columns outside the image become zero records)
    auto stage_row = [&](int r, int buf) {
        int tids = threadIdx.x;          // opaque copy: keeps the staging indices from being hoisted out of the
        asm volatile("" : "+v"(tids));   // row loop and held (or spilled) across the aggregation
        for (int k = tids; k < win; k += nthr) proxS[buf * win + k] = A.prox[(r - (y - p)) * win + k];
        const PixRec *const rowL = A.recL + (size_t)r * W;
        const PixRec *const rowR = A.recR + (size_t)r * W;
        for (int k = tids; k < nL + nR; k += nthr) {
            const bool isL = k < nL;
            const int idx = isL ? k : k - nL;
            const int col = (isL ? segL_lo : segR_lo) + idx;
            PixRec v;
            v.L = v.a = v.b = 0.f;
            v.bgrx = 0u;
            if ((unsigned)col < (unsigned)W) v = (isL ? rowL : rowR)[col];
            (isL ? labL + buf * nL : labR + buf * nR)[idx] = make_float4(v.L, v.a, v.b, 0.f);
            (isL ? bgrL + buf * nL : bgrR + buf * nR)[idx] = v.bgrx;
        }
    };
    // Tap columns are staged JC at a time when the whole window row of weights does not leave room for
    // enough resident waves (small disparity ranges: few threads share a weight row); chunk buffers
    // alternate so that one barrier per chunk suffices.  JC == win: a single chunk, rows used in place.
    const int JC = CHUNKED ? g.JC : win;
    int cb = 0;
    for (int i = i_lo; i < i_hi; ++i) {
        const int r = y - p + i;

        // per-phase thread indices are re-derived from an opaque copy of the thread id so that
        // they are not kept live across the aggregation loop (168-VGPR budget, no scratch spills)
        int tidb = threadIdx.x;
        asm volatile("" : "+v"(tidb));
        // ---- pixels of image row r were staged into buffer (i & 1) during the previous
        //      iteration's aggregation (prologue for the first row): global latency is hidden
        float4 *const labLc = labL + (i & 1) * nL, *const labRc = labR + (i & 1) * nR;
        uint32_t *const bgrLc = bgrL + (i & 1) * nL, *const bgrRc = bgrR + (i & 1) * nR;
        if (i == i_lo) stage_row(r, i & 1);
        // staged pixels visible; every thread is done with main(i-1).  With two e tiles the barrier is only needed
        // for the first row: the pixels of later rows were staged before an earlier chunk barrier of the previous row,
        // this row's e tile and first weight chunk go to the buffers the previous row's last chunk does not read,
        // and stragglers of that chunk are waited for at this row's first chunk barrier.
        if (!(CHUNKED && g.e2) || i == i_lo) __syncthreads();
        unsigned char *const eT = eT0 + ((CHUNKED && g.e2) ? (i & 1) * g.e_bytes : 0);

        // ---- truncated absolute differences e[ul][d] = min(40, |dB|+|dG|+|dR|) (_passive.cpp:77-79);
        //      pixel bytes are B,G,R,0 so v_sad_u8 sums the 3 channels.  Task = (tap column ul,
        //      pair of disparity groups = 8 disparities); consecutive lanes = consecutive ul.
        {
            const int nsp = (g.DG + 1) >> 1, e_q = nthr / nL, e_r = nthr - e_q * nL;
            int sp = tidb / nL, ul = tidb - sp * nL;
            while (sp < nsp) {
                const uint32_t lp = bgrLc[ul];
                const uint32_t *const rp = bgrRc + (ul + (Dc - 1) - 8 * sp);   // R[u-d] for d = dlo + 8*sp
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo |= min(__builtin_amdgcn_sad_u8(lp, rp[-k], 0u), 40u) << (8 * k);
                    hi |= min(__builtin_amdgcn_sad_u8(lp, rp[-4 - k], 0u), 40u) << (8 * k);
                }
                *reinterpret_cast<uint32_t *>(eT + asw_e_offset<RX>(ul, 2 * sp, Se, emask)) = lo;
                if (2 * sp + 1 < g.DG) *reinterpret_cast<uint32_t *>(eT + asw_e_offset<RX>(ul, 2 * sp + 1, Se, emask)) = hi;
                ul += e_r; sp += e_q;
                if (ul >= nL) { ul -= nL; ++sp; }
            }
        }

        // ---- per-row aggregation state (register window of e rows, running e pointer)
        int tidm = threadIdx.x;
        asm volatile("" : "+v"(tidm));
        // a register tile contributes only if some (x,d) of it is a candidate the reference evaluates
        // (x - d >= 0, d <= maxDisparity, x < W); waves whose lanes are all outside (left image border,
        // padded disparities) skip the aggregation -- wave-uniform, decided once per window row
        bool run;
        {
            const int xg = tidm % g.XG, dg = tidm / g.XG;
            const bool tile_live = tidm < g.XG * g.DG && x0 + RX * xg < W && dlo + ASW_RD * dg <= A.maxD &&
                                   x0 + RX * xg + RX - 1 - (dlo + ASW_RD * dg) >= 0;
            run = __builtin_amdgcn_ballot_w64(tile_live) != 0 && tidm < g.XG * g.DG;
        }
        AswRow ew[RX];      // the only per-row state (besides the accumulators) that lives across the chunks' build phases

        for (int jc = 0; jc < win; jc += JC) {
            const int jend = min(win, jc + JC);
            const int rb = CHUNKED ? cb * JC : 0;          // first buffer row of this chunk
            // ---- support weights of window row i, tap columns [jc, jend) (_passive.cpp:47-50 and 71-74;
            //      exp(-dist/gammaC) = exp2(dist*kC)).  Task = (window centre c, segment of the tap
            //      columns); consecutive lanes take consecutive centres: conflict-free ds_read_b128 of the
            //      staged pixels and coalesced LDS writes.  Branch-free: taps or centres outside the image
            //      get weight 0 through a bit mask.
            {
                int tidw = threadIdx.x;
                asm volatile("" : "+v"(tidw));
                const float *const prow = proxS + (i & 1) * win;    // proximity weights of window row i, staged in LDS
                const int ncen = Tx + nRc;
                for (int t = tidw; t < ncen * g.wseg; t += nthr) {
                    const int sgm = t / ncen, c = t - sgm * ncen;
                    const bool isL = c < Tx;
                    const int cc = isL ? c : c - Tx;
                    const float4 cen = cenLab[c];                      // centre pixel (row y); .w = inside image
                    const float4 *const seg = (isL ? labLc : labRc) + cc;
                    const int stride = isL ? g.SL : SR;
                    float *const wout = (isL ? wL : wR) + asw_split_pos(cc, isL ? g.hL : g.hR) + (rb - jc) * stride;
                    const int col0 = (isL ? x0 : xrc_lo) + cc - p;
                    const int j1 = min(jend, jc + (sgm + 1) * g.wlen);
                    const uint32_t cmask = cen.w != 0.f ? 0xffffffffu : 0u;
                    // batches of ASW_WB independent evaluations: all LDS reads first, then the dependent
                    // chains (sub, fma, v_sqrt, v_exp, mul) side by side -- a single chain is ~150 cycles of
                    // latency, and during this phase every wave of the group is in the same loop.  Whole
                    // batches walk running pointers (staged pixels, proximity row, output column); only the
                    // last, partial batch of a segment pays for clamped indices.
                    int j = jc + sgm * g.wlen;
                    const float4 *sp = seg + j;
                    const float *pp = prow + j;
                    float *wp = wout + j * stride;
                    int col = col0 + j;
                    const int stride4 = ASW_WB * stride;
                    for (; j + ASW_WB <= j1; j += ASW_WB, sp += ASW_WB, pp += ASW_WB, wp += stride4, col += ASW_WB) {
                        float4 tp[ASW_WB];
                        float pr[ASW_WB], wv[ASW_WB];
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) { tp[u] = sp[u]; pr[u] = pp[u]; }
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) {
                            const float dL = tp[u].x - cen.x, da = tp[u].y - cen.y, db = tp[u].z - cen.z;
                            wv[u] = fmaf(db, db, fmaf(da, da, dL * dL));
                        }
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) wv[u] = __builtin_amdgcn_sqrtf(wv[u]);
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) wv[u] = asw_weight_finish(wv[u], A.kC, pr[u]);
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) {
                            const uint32_t m = (unsigned)(col + u) < (unsigned)W ? cmask : 0u;
                            wp[u * stride] = __uint_as_float(__float_as_uint(wv[u]) & m);
                        }
                    }
                    if (j < j1) {
                        float4 tp[ASW_WB];
                        float pr[ASW_WB], wv[ASW_WB];
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) {
                            const int jj = min(j + u, j1 - 1);
                            tp[u] = seg[jj];
                            pr[u] = prow[jj];
                        }
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) {
                            const float dL = tp[u].x - cen.x, da = tp[u].y - cen.y, db = tp[u].z - cen.z;
                            wv[u] = fmaf(db, db, fmaf(da, da, dL * dL));
                        }
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) wv[u] = __builtin_amdgcn_sqrtf(wv[u]);
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) wv[u] = asw_weight_finish(wv[u], A.kC, pr[u]);
#pragma unroll
                        for (int u = 0; u < ASW_WB; ++u) {  // past the segment end the clamped tap is simply rewritten
                            const int jj = min(j + u, j1 - 1);
                            const uint32_t m = (unsigned)(col0 + jj) < (unsigned)W ? cmask : 0u;
                            wout[jj * stride] = __uint_as_float(__float_as_uint(wv[u]) & m);
                        }
                    }
                }
            }
            __syncthreads();   // e tile and this chunk of wL, wR ready; every thread is done with the previous chunk
            if (jc == 0 && i + 1 < i_hi) stage_row(r + 1, (i + 1) & 1);   // prefetch: overlaps with the aggregation below

            // ---- aggregation over the tap columns of this chunk
            if (run) {
                // thread coordinates, e-row pointer and swizzle state are re-derived per chunk from an opaque thread
                // id: nothing but the e window and the accumulators stays live across a weight-build phase
                int tida = threadIdx.x;
                asm volatile("" : "+v"(tida));
                const int xg = tida % g.XG, dg = tida / g.XG;
                // the next e row to load is ul0 + RX - 1 + jc (ul0 + 0 before the priming of the first chunk); the
                // swizzled dword slot of the thread's disparity group depends on row / RX only, i.e. it changes
                // once per RX tap columns
                const unsigned char *erow = eT + (RX * xg + (jc == 0 ? 0 : RX - 1 + jc)) * Se;
                int q = xg + jc / RX;
                int slotA = (dg ^ (q & emask)) << 2;
                if (jc == 0) {
#pragma unroll
                    for (int n = 0; n < RX - 1; ++n) {
                        asw_row_unpack(ew[n], *reinterpret_cast<const uint32_t *>(erow + slotA));
                        erow += Se;
                    }
                }
                // running LDS pointers (advanced by one tap column per step) keep the address arithmetic at
                // ~4 VALU ops per step and nothing step-specific live across the loop
                // RX = 8: even block of the thread's columns, odd block at + hL;  RX = 4: the thread's single block
                const float *wlp = wL + rb * g.SL + (RX == 8 ? RX / 2 * xg : (xg & 1) * g.hL + ((xg >> 1) << 2));
                // right weights: NWR/4 consecutive 4-float blocks starting at block b0 (parity-split rows)
                const int b0 = (RX * xg - ASW_RD * dg + Dc - ASW_RD) >> 2;
                const float *wrp0 = wR + rb * SR + (b0 & 1) * g.hR + ((b0 >> 1) << 2);
                const float *wrp1 = wR + rb * SR + ((b0 + 1) & 1) * g.hR + (((b0 + 1) >> 1) << 2);
                const float *wrp2 = wR + rb * SR + (b0 & 1) * g.hR + (((b0 + 2) >> 1) << 2);
                for (int j0 = jc; j0 < jend; j0 += RX) {
                    const int slotB = (dg ^ ((q + 1) & emask)) << 2;
#define SSAMD_STEP(JJ, SLOT)                                                                        \
    if (j0 + (JJ) < jend) {                                                                         \
        asw_row_unpack(ew[((JJ) + RX - 1) % RX], *reinterpret_cast<const uint32_t *>(erow + (SLOT))); \
        erow += Se;                                                                                 \
        float wl[RX], wr[NWR];                                                              \
        {                                                                                           \
            const float4 v0 = *reinterpret_cast<const float4 *>(wlp);                              \
            wl[0] = v0.x; wl[1] = v0.y; wl[2] = v0.z; wl[3] = v0.w;                                 \
            if constexpr (RX == 8) {                                                                \
                const float4 v1 = *reinterpret_cast<const float4 *>(wlp + g.hL);                   \
                wl[RX - 4] = v1.x; wl[RX - 3] = v1.y; wl[RX - 2] = v1.z; wl[RX - 1] = v1.w;         \
            }                                                                                       \
            const float4 r0 = *reinterpret_cast<const float4 *>(wrp0);                             \
            const float4 r1 = *reinterpret_cast<const float4 *>(wrp1);                             \
            wr[0] = r0.x; wr[1] = r0.y; wr[2] = r0.z; wr[3] = r0.w;                                 \
            wr[4] = r1.x; wr[5] = r1.y; wr[6] = r1.z; wr[7] = r1.w;                                 \
            if constexpr (RX == 8) {                                                                \
                const float4 r2 = *reinterpret_cast<const float4 *>(wrp2);                         \
                wr[NWR - 4] = r2.x; wr[NWR - 3] = r2.y; wr[NWR - 2] = r2.z; wr[NWR - 1] = r2.w;     \
            }                                                                                       \
        }                                                                                           \
        wlp += g.SL; wrp0 += SR; wrp1 += SR; wrp2 += SR;                                            \
        asw_taps<RX, (JJ)>(accN, accS, wl, wr, ew);                                                 \
    }
                    // the row loaded at step JJ is row j + RX - 1: (row / RX) == q for JJ = 0, q + 1 afterwards
                    SSAMD_STEP(0, slotA) SSAMD_STEP(1, slotB) SSAMD_STEP(2, slotB) SSAMD_STEP(3, slotB)
                    if constexpr (RX == 8) {
                        SSAMD_STEP(RX - 4, slotB) SSAMD_STEP(RX - 3, slotB) SSAMD_STEP(RX - 2, slotB) SSAMD_STEP(RX - 1, slotB)
                    }
#undef SSAMD_STEP
                    slotA = slotB;
                    ++q;
                }
            }
            if (CHUNKED) cb ^= 1;
        }
    }

    // ---- weighted average (_passive.cpp:88) and the two WTA reductions
    int tidf = threadIdx.x;
    asm volatile("" : "+v"(tidf));
    AswKeyTile<RX, ASW_RD> kt;                      // cost images of the register tile (exact mode re-reads them after the barrier)
    if (tidf < g.XG * g.DG) {
        const int xg = tidf % g.XG, dg = tidf / g.XG;
        u64 diag[RX + ASW_RD - 1];
#pragma unroll
        for (int k = 0; k < RX + ASW_RD - 1; ++k) diag[k] = KEY_NONE;
#pragma unroll
        for (int xi = 0; xi < RX; ++xi) {
            const int x = x0 + RX * xg + xi;
            u64 bl = KEY_NONE;
#pragma unroll
            for (int di = 0; di < ASW_RD; ++di) {
                const int d = dlo + ASW_RD * dg + di;
                const bool valid = (x < W) && (d <= A.maxD) && (x - d >= 0);
                kt.v[xi][di] = 0xffffffffu;
                if (valid) {
                    float c;
                    const u64 hi = (u64)asw_cost_key(accN[xi][di], accS[xi][di], c) << 32;
                    kt.v[xi][di] = (uint32_t)(hi >> 32);
                    bl = min(bl, hi | (u64)(uint32_t)d);
                    diag[xi - di + ASW_RD - 1] = min(diag[xi - di + ASW_RD - 1], hi | (u64)(uint32_t)x);
                    if (WITH_COSTS)
                        A.costs[((size_t)(y - A.row0) * W + x) * (A.maxD - A.minD + 1) + (d - A.minD)] = A.cost_keys ? __uint_as_float((uint32_t)(hi >> 32)) : c;
                }
            }
            if (bl != KEY_NONE) atomicMin(&bestL[RX * xg + xi], bl);
        }
        if (A.keyR) {
            const int base = RX * xg - ASW_RD * dg + Dc - ASW_RD;
#pragma unroll
            for (int k = 0; k < RX + ASW_RD - 1; ++k)
                if (diag[k] != KEY_NONE) atomicMin(&bestR[base + k], diag[k]);
        }
    }
    __syncthreads();
    const size_t orow = (size_t)(y - A.row0) * W;
    const bool xq = !WITH_COSTS && A.xq.entries != nullptr;          // exact mode: near-ties of the winners go to the fp64 pass's queue
    if (xq) {
        const bool live = tidf < g.XG * g.DG;
        const int xg = live ? tidf % g.XG : 0, dg = live ? tidf / g.XG : 0;
        asw_exact_select<RX, ASW_RD>(A.xq, live, kt, bestL + RX * xg, A.keyR ? bestR + (RX * xg - ASW_RD * dg + Dc - ASW_RD) : nullptr,
                                     x0 + RX * xg, dlo + ASW_RD * dg, (uint32_t)orow);
    }
    if (A.disp) {
        for (int k = tid; k < Tx; k += nthr) {
            const int x = x0 + k;
            if (x < W) A.disp[orow + x] = bestL[k] == KEY_NONE ? (int16_t)x : (int16_t)(uint32_t)bestL[k];
        }
        return;
    }
    if (xq) {
        // tile-local winners meet the pixels' running minima: the loser of each meeting is queued if it is a near-tie (uniform trip counts)
        for (int k0 = 0; k0 < Tx; k0 += nthr) {
            const int k = k0 + tid, x = x0 + k;
            const bool have = k < Tx && x < W && bestL[k < Tx ? k : 0] != KEY_NONE;
            const u64 mine = have ? bestL[k] : KEY_NONE;
            const u64 old = have ? atomicMin(&A.keyL[orow + x], mine) : KEY_NONE;
            asw_exact_merge<false>(A.xq, have, mine, old, (uint32_t)orow, x);
        }
        if (A.keyR)
            for (int k0 = 0; k0 < nRc; k0 += nthr) {
                const int k = k0 + tid, xr = xrc_lo + k;
                const bool have = k < nRc && (unsigned)xr < (unsigned)W && bestR[k < nRc ? k : 0] != KEY_NONE;
                const u64 mine = have ? bestR[k] : KEY_NONE;
                const u64 old = have ? atomicMin(&A.keyR[orow + xr], mine) : KEY_NONE;
                asw_exact_merge<true>(A.xq, have, mine, old, (uint32_t)orow, xr);
            }
        return;
    }
    for (int k = tid; k < Tx; k += nthr) {
        const int x = x0 + k;
        if (x < W && bestL[k] != KEY_NONE) atomicMin(&A.keyL[orow + x], bestL[k]);
    }
    if (A.keyR) {
        for (int k = tid; k < nRc; k += nthr) {
            const int xr = xrc_lo + k;
            if ((unsigned)xr < (unsigned)W && bestR[k] != KEY_NONE) atomicMin(&A.keyR[orow + xr], bestR[k]);
        }
    }
}

#ifndef SSAMD_KERNEL_TU          // (the translation units that only instantiate one aggregation kernel family: asw_pipe_tu.hip, asw_wave6_tu.hip)
// K2a: decode left keys (non-consistent mode).  disparity = d of the best key, or x
// when the candidate loop was empty (dBest stays 0, _passive.cpp:54,98).
// right_keys != 0: the keys are right-referenced (low word = best LEFT column of the right pixel, 0 when its
// candidate loop was empty, _passive.cpp:209) -- only used by the verification dump ssamd_asw_argmins.
__global__ __launch_bounds__(256) void wta_decode_kernel(const u64 *__restrict__ keyL, int16_t *__restrict__ disp,
                                                         int rows, int W, int right_keys)
{
    const long long n = (long long)rows * W;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) {
        const u64 k = keyL[idx];
        const int x = right_keys ? 0 : (int)(idx % W);
        disp[idx] = (k == KEY_NONE) ? (int16_t)x : (int16_t)(uint32_t)k;
    }
}

// K2b: left-right check + occlusion filling, one workgroup per image row
// (_passive.cpp:250-285; GSW 661-696).  keyR low word = best left column for the
// right pixel, 0 when its candidate loop was empty (dBest stays 0, :209).
// A left pixel is invalidated iff some right pixel selects it while the left
// disparity disagrees; this is order independent, unlike the reference's
// sequential formulation.  Runs of invalid pixels take min(left,right) valid
// neighbour, or the single valid neighbour at the image border.  A fully invalid
// row keeps -1 (the reference reads out of bounds there).
__global__ __launch_bounds__(256) void lr_check_fill_kernel(const u64 *__restrict__ keyL, const u64 *__restrict__ keyR,
                                                            int16_t *__restrict__ disp, int rows, int W)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t *d = reinterpret_cast<int16_t *>(smem);
    unsigned char *inv = reinterpret_cast<unsigned char *>(smem + (((size_t)W * 2 + 15) & ~(size_t)15));
    const int y = blockIdx.x;
    const u64 *kl = keyL + (size_t)y * W, *kr = keyR + (size_t)y * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const u64 k = kl[x];
        d[x] = (k == KEY_NONE) ? (int16_t)x : (int16_t)(uint32_t)k;
        inv[x] = 0;
    }
    __syncthreads();
    for (int xr = threadIdx.x; xr < W; xr += blockDim.x) {
        const u64 k = kr[xr];
        const int best = (k == KEY_NONE) ? 0 : (int)(uint32_t)k;
        if ((int)d[best] != best - xr) inv[best] = 1;
    }
    __syncthreads();
    int16_t *out = disp + (size_t)y * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int16_t v = d[x];
        if (inv[x]) {
            int lo = x - 1, hi = x + 1;
            while (lo >= 0 && inv[lo]) --lo;
            while (hi < W && inv[hi]) ++hi;
            if (lo < 0 && hi >= W) v = -1;
            else if (lo < 0) v = d[hi];
            else if (hi >= W) v = d[lo];
            else v = min(d[lo], d[hi]);
        }
        out[x] = v;
    }
}

#endif  // SSAMD_KERNEL_TU

}  // namespace ssamd
