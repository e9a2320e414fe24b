// K1p: the phase-shifted ("pipelined") form of the ASW aggregation kernel (gfx950).
//
// Same algebra, same per-thread 8 x 4 register tile, same tap order and therefore bit-identical sums as
// asw_aggregate_kernel (asw_kernels.hip.h); what changes is WHEN each wave does its build work.
//
// asw_aggregate_kernel<CHUNKED> runs every wave in lockstep:  build(c) | barrier | aggregate(c) | build(c+1) | barrier ...
// The build work (support weights: LDS read -> 3 sub, 3 fma -> v_sqrt -> v_exp -> LDS write; e tiles; pixel staging)
// is a dependent chain of ~200 cycles with little to overlap it when all twelve waves of the CU are in it at the same
// time: measured, the build phases alone cost 13.8 ms of the 44.5 ms of a 1080p / D 0..192 / win 35 frame although
// their VALU work is worth ~4 ms.
//
// Here the build for the NEXT chunk only touches buffers the current chunk does not read (second weight buffer,
// second e tile, other pixel-staging buffer), so a wave may do it at any point between the chunk's two barriers.
// The twelve waves of a workgroup sit three per SIMD (waves w, w+4, w+8 share a SIMD) and are given three different
// orders, one per SIMD-mate:
//     waves 0-3  : build(next) , aggregate(chunk)
//     waves 4-11 : aggregate(chunk) , build(next)
// so that on every SIMD the first wave's latency-bound build chains are covered by the FMA streams of its two
// mates, and theirs by the first wave's (it starts aggregating later and is still at it when they build).
// One barrier per chunk, as before.
//
// Chunks: tap columns [c*JC, (c+1)*JC) with JC = 8 or 16 and a tail shorter than 8 merged into the last chunk
// (win 35, JC 8: 8, 8, 8, 11), so that the work hidden under a chunk and the chunk itself stay comparable.
// build(next) of chunk c of window row i is
//     c == 0        : stage the pixels of image row i+1 (global -> LDS) + weights of chunk 1
//     0 < c < NC-1  : weights of chunk c+1                              + a share of the e tile of row i+1
//     c == NC-1     : weights of chunk 0 of row i+1                     + the last share of that e tile
// Needs NC >= 2 chunks (the pixels staged under chunk 0 are first read under chunk 1) and two e tiles.
//
// LDS layout ("plain rows", AswGeom::pipe): the lanes of a wave run along the DISPARITY groups of one column group
// (thread = xg * DG + dg), not along x as in asw_aggregate_kernel.  The right-weight blocks a thread reads start at
// float 8 xg - 4 dg + Dc - 4 of a weight row, so consecutive lanes read consecutive 16-byte blocks of a PLAIN row
// (the hardware serves a ds_read_b128 in groups of 16 lanes whose 16-byte slots must differ mod 16: any run of 32
// consecutive blocks does); the left-weight blocks are the same for all lanes of a column group (broadcast); the e
// dwords of a lane group are consecutive in a plain e row.  rocprofv3 on the x-fastest layout with its 15-wide
// thread rows: 21 bank-conflict cycles per 26-cycle aggregation step (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.37, LDS 55 % busy); see profiles/r02_*.  It also needs one running pointer per operand instead of three for
// wR and no swizzle arithmetic.
#pragma once
#include "asw_kernels.hip.h"
#ifndef SSAMD_KERNEL_TU
#include "lab_kernels.hip.h"
#endif

#ifndef SSAMD_PIPE_SENTINEL       // 0: the round-2 weight build with masks (A/B builds of tools/build_variants.sh)
#define SSAMD_PIPE_SENTINEL 1
#endif

namespace ssamd {

#ifndef SSAMD_KERNEL_TU
// K0e: the truncated absolute differences e[r][u][d] = min(40, |dB|+|dG|+|dR|) of L[r][u] and R[r][u-d]
// (_passive.cpp:77-79) for every image row the launch touches, as bytes in exactly the layout of the kernel's e
// tiles: [disparity chunk z][row][column u + pad][Se bytes = one dword per 4 disparities, padded].  e depends on
// (r, u, d) only, so each value is needed by the up to winSize window rows of winSize output rows: built once here
// (H*W*nD bytes: 0.4 GB at 1080p / 193, a 0.1 ms HBM-bound kernel) instead of winSize times inside the aggregation
// kernel, whose workgroups then fetch their tiles with LDS-DMA (global_load_lds_dwordx4: no VALU work, no
// registers).  Columns or disparities outside the image get 0; their taps carry weight 0.
// Workgroup = one image row x 64 columns x one disparity chunk.  The pixels are staged in LDS (coalesced record
// reads), every thread then produces whole dwords (4 disparities) of one column, and the 64 x Se byte block -- a
// contiguous piece of the volume -- is written with consecutive lanes on consecutive dwords.
static constexpr int TADV_COLS = 64;
// One tile (bx, by, bz) of the volume.  BYTES: the pixels come straight from the caller's uint8 [H][W][3] images instead of the
// records -- the volume then does not depend on the Lab conversion and both run in ONE launch (asw_prepass_kernel below).
template <bool BYTES>
__device__ __forceinline__ void asw_tad_tile(const void *__restrict__ srcL, const void *__restrict__ srcR, unsigned char *__restrict__ evol,
                                             int W, int pad, int minD, int Dc, int Se, int erow0, int erows, int evolW, int rd, long long npix_total,
                                             int bx, int by, int bz, char *smem)
{
    uint32_t *const sL = reinterpret_cast<uint32_t *>(smem);                 // [TADV_COLS]
    uint32_t *const sR = sL + TADV_COLS;                                     // [TADV_COLS + Dc]: right columns u0 - dhi .. u0 + 63 - dlo
    const int r = erow0 + by, z = bz, uc0 = bx * TADV_COLS;
    const int P = Se >> 2, dlo = minD + z * Dc, dhi = dlo + Dc - 1;
    const int u0 = uc0 - pad, xr0 = u0 - dhi;
    auto pixel = [&](const void *src, int col) -> uint32_t {
        const size_t q = (size_t)r * W + col;
        if constexpr (BYTES) {
            typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
            const uint8_t *const b = reinterpret_cast<const uint8_t *>(src) + 3 * q;
            // (the 4-byte read of the image's last pixel would run one byte past the buffer)
            return (long long)q + 1 < npix_total ? (*reinterpret_cast<const u32_unaligned *>(b) & 0xffffffu)
                                                 : ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16));
        } else {
            return reinterpret_cast<const PixRec *>(src)[q].bgrx;
        }
    };
    for (int k = threadIdx.x; k < 2 * TADV_COLS + Dc; k += blockDim.x) {
        const bool isL = k < TADV_COLS;
        const int col = isL ? u0 + k : xr0 + (k - TADV_COLS);
        // bit 31 marks a column outside the image (pixel bytes never set it): its e values are 0
        const uint32_t v = (unsigned)col < (unsigned)W ? pixel(isL ? srcL : srcR, col) : 0x80000000u;
        (isL ? sL : sR)[isL ? k : k - TADV_COLS] = v;
    }
    __syncthreads();
    const int ncols = min(TADV_COLS, evolW - uc0);
    uint32_t *const out = reinterpret_cast<uint32_t *>(evol + (((size_t)z * erows + by) * (size_t)evolW + uc0) * Se);
    // rd = 4: dword `slot` of a column holds the disparities dlo + 4 slot + 0..3;  rd = 6 (asw_wave6_kernel.hip.h): a
    // disparity group is an 8-byte slot, dword 2 g holds dlo + 6 g + 0..3 and dword 2 g + 1 holds dlo + 6 g + 4, 5
    auto dword = [&](int c, int slot, uint32_t lp) {
        uint32_t v = 0;
        const int d0 = rd == 6 ? 6 * (slot >> 1) + 4 * (slot & 1) : 4 * slot, nv = rd == 6 && (slot & 1) ? 2 : 4;
        if (!(lp >> 31) && d0 < Dc) {
            // R[u - d] for d = dlo + d0 + q  ->  staged index c + (Dc - 1) - d0 - q
            const uint32_t *const rp = sR + c + (Dc - 1) - d0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t rv = q < nv && d0 + q < Dc ? rp[-q] : 0x80000000u;
                if (!(rv >> 31)) v |= min(__builtin_amdgcn_sad_u8(lp, rv, 0u), 40u) << (8 * q);
            }
        }
        return v;
    };
    if ((P & 3) == 0) {
        // rows of whole 16-byte blocks (the phase-shifted kernel's layout): a thread produces 16 disparities and one
        // 16-byte store -- 1 KiB per wave and store instruction instead of 256 B
        const int P4 = P >> 2;
        uint4 *const out4 = reinterpret_cast<uint4 *>(out);
        for (int k = threadIdx.x; k < ncols * P4; k += blockDim.x) {
            const int c = k / P4, s4 = 4 * (k - c * P4);
            const uint32_t lp = sL[c];
            out4[k] = make_uint4(dword(c, s4, lp), dword(c, s4 + 1, lp), dword(c, s4 + 2, lp), dword(c, s4 + 3, lp));
        }
        return;
    }
    for (int k = threadIdx.x; k < ncols * P; k += blockDim.x) {
        const int c = k / P, slot = k - c * P;
        out[k] = dword(c, slot, sL[c]);
    }
}

__global__ __launch_bounds__(256) void asw_tad_volume_kernel(const PixRec *__restrict__ recL, const PixRec *__restrict__ recR,
                                                             unsigned char *__restrict__ evol, int W, int pad, int minD, int Dc,
                                                             int Se, int erow0, int erows, int evolW, int rd = 4)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asw_tad_tile<false>(recL, recR, evol, W, pad, minD, Dc, Se, erow0, erows, evolW, rd, 0, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// K0 + K0e in ONE launch (round 5): the first `lab_blocks` workgroups convert the pixels of the rows [erow0, erow0 + erows) of both
// images into records (bgr2lab_records_pair_kernel's job, same code), the others each build one tile of the volume from the
// images' bytes.  The two jobs do not depend on each other, so a small-frame call is two dependent launches instead of three
// (a launch-to-launch dependency costs more than either kernel at Tsukuba size).  bgrL / bgrR: the sub-image's row 0.
struct AswPrepassArgs {
    const uint8_t *bgrL, *bgrR;
    PixRec *recL, *recR;
    unsigned char *evol;
    long long npix_total;            // pixels of the whole sub-image (bounds of the 4-byte pixel reads)
    int W, pad, minD, Dc, Se, erow0, erows, evolW, rd;
    int lab_blocks, ex, ey;          // grid: lab_blocks + ex * ey * ez workgroups
};
__global__ __launch_bounds__(256) void asw_prepass_kernel(const AswPrepassArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < P.lab_blocks) {
        SSAMD_LAB_TABLES_IN_LDS(T)
        const long long np = (long long)P.erows * P.W, first = (long long)P.erow0 * P.W;
        const uint8_t *const bl = P.bgrL + 3 * first, *const br = P.bgrR + 3 * first;
        PixRec *const rl = P.recL + first, *const rr = P.recR + first;
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < 2 * np; q += (long long)P.lab_blocks * blockDim.x) {
            const bool right = q >= np;
            const long long p = right ? q - np : q;
            const uint8_t *const bgr = right ? br : bl;
            uint32_t B, G, R;
            if (p + 1 < np) {
                const uint32_t v = *reinterpret_cast<const u32_unaligned *>(bgr + 3 * p);
                B = v & 0xff; G = (v >> 8) & 0xff; R = (v >> 16) & 0xff;
            } else {
                B = bgr[3 * p]; G = bgr[3 * p + 1]; R = bgr[3 * p + 2];
            }
            PixRec o;
            bgr_to_lab(B, G, R, o.L, o.a, o.b, T);
            o.bgrx = B | (G << 8) | (R << 16);
            (right ? rr : rl)[p] = o;
        }
        return;
    }
    const int b = (int)blockIdx.x - P.lab_blocks, bx = b % P.ex, by = (b / P.ex) % P.ey, bz = b / (P.ex * P.ey);
    asw_tad_tile<true>(P.bgrL, P.bgrR, P.evol, P.W, P.pad, P.minD, P.Dc, P.Se, P.erow0, P.erows, P.evolW, P.rd, P.npix_total, bx, by, bz, smem);
}

#endif  // SSAMD_KERNEL_TU

// (A 6-column register tile -- 48 accumulators, 128 VGPRs, FOUR waves per SIMD in 1024-thread groups, right weights
// read as 8-byte pairs -- was built and measured in round 2: bit-identical maps, 43.6 ms against 38.2 ms for this
// kernel on 1080p / 193 / 35.  A third more LDS traffic per tap outweighs the fourth wave; code removed.)
// SLC, SRC, SEC (round 3): the strides of the left / right weight rows (floats) and of the e rows (bytes) as
// compile-time constants for the launch geometries of the headline configurations (0: read from the geometry at run
// time).  The eight steps of a trip then address their operands with immediate offsets from one pointer per array,
// advanced once per trip: 3 address instructions per 8 steps instead of 24 (107 -> 104.4 VALU instructions per step).
// CG (round 6): the static instantiation takes the WHOLE tile geometry and the window from compile-time constants (see AswPipeTile);
// false = strides only (rounds 3-5).  Both are built so that they can be compared in one process (SSAMD_ASW_STATIC=2 / 1).
template <bool WITH_COSTS, int SLC = 0, int SRC = 0, int SEC = 0, bool CG = false>
__global__ __launch_bounds__(ASW_MAX_THREADS, 3) void asw_aggregate_pipe_kernel(const AswArgs A)
{
    constexpr bool STATIC = SLC > 0;
    static_assert(!CG || STATIC, "compile-time geometry belongs to a static tile");
    constexpr int RX = ASW_RX;
    constexpr int NWR = asw_nwr(RX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // static instantiations: the whole tile geometry and the window are compile-time constants (the launcher checked A.g against
    // them: asw_pipe_geom_matches); the generic instantiation reads them from the arguments
    constexpr AswGeom SG = asw_pipe_geom_constexpr(AswPipeTile<SLC, SRC, SEC>::id);
    static_assert(!STATIC || (SG.SL == SLC && SG.SR == SRC && SG.Se == SEC), "the strides name the tile");
    const AswGeom &g = CG ? SG : A.g;
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
    const int W = A.W, win = CG ? AswPipeTile<SLC, SRC, SEC>::id.win : A.win, p = CG ? AswPipeTile<SLC, SRC, SEC>::id.win / 2 : A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nR = g.nR, nRc = g.nRc, SR = g.SR, Se = g.Se;
    const int JC = g.JC, NC = g.NC;
    int bx = blockIdx.x;                          // XCD-aware tile order, see asw_aggregate_kernel
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int x0 = bx * Tx;
    const int y = asw_out_row(A, blockIdx.y);
    const int dlo = A.minD + blockIdx.z * Dc;
    const int dhi = dlo + Dc - 1;
    if (min(x0 + Tx - 1, W - 1) - dlo < 0) {      // no candidate the reference evaluates in this tile
        if (A.disp)
            for (int k = threadIdx.x; k < Tx && x0 + k < W; k += blockDim.x)
                A.disp[(size_t)(y - A.row0) * W + x0 + k] = (int16_t)(x0 + k);
        return;
    }
    const int segL_lo = x0 - p, xrc_lo = x0 - dhi, segR_lo = xrc_lo - p;
    // which of the two orders this wave follows (0: build first): wave-uniform, kept in a scalar register
    const int phase = g.dephase ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 1;

    float accN[RX][ASW_RD], accS[RX][ASW_RD];
#pragma unroll
    for (int a = 0; a < RX; ++a)
#pragma unroll
        for (int b = 0; b < ASW_RD; ++b) { accN[a][b] = 0.f; accS[a][b] = 0.f; }

    for (int k = tid; k < Tx; k += nthr) bestL[k] = KEY_NONE;
    for (int k = tid; k <= nRc; k += nthr) bestR[k] = KEY_NONE;
    for (int c = tid; c < Tx + nRc; c += nthr) {   // window centres (row y): left columns x0.., right columns xrc_lo..
        const bool isL = c < Tx;
        const int ccol = isL ? x0 + c : xrc_lo + (c - Tx);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ccol < (unsigned)W) {
            const PixRec q = (isL ? A.recL : A.recR)[(size_t)y * W + ccol];
            v = make_float4(q.L, q.a, q.b, 1.f);
        }
        cenLab[c] = v;
    }

    const int i_lo = max(0, p - y), i_hi = min(win, A.H + p - y);

    // ---- build pieces.  Every thread index they use is re-derived from an opaque copy of threadIdx.x so that
    //      nothing of a build stays live across the aggregation (168-VGPR budget, no scratch).
    // pixels (and the proximity row) of window row i -> staging buffer i & 1
    auto stage_row = [&](int i) {
        int tids = threadIdx.x;
        asm volatile("" : "+v"(tids));
        const int buf = i & 1, r = y - p + i;
        for (int k = tids; k < win; k += nthr) proxS[buf * win + k] = A.prox[i * win + k];
        const PixRec *const rowL = A.recL + (size_t)r * W;
        const PixRec *const rowR = A.recR + (size_t)r * W;
        for (int k = tids; k < nL + nR; k += nthr) {
            const bool isL = k < nL;
            const int idx = isL ? k : k - nL;
            const int col = (isL ? segL_lo : segR_lo) + idx;
            // a tap column outside the image gets L = +inf: colour distance +inf, exp2(-inf) = +0 -- the weight 0 the
            // reference's bounds test gives (_passive.cpp:60-62), exactly and without a mask in the weight build
            PixRec v;
            v.L = SSAMD_PIPE_SENTINEL ? __builtin_inff() : 0.f;
            v.a = v.b = 0.f;
            v.bgrx = 0u;
            if ((unsigned)col < (unsigned)W) v = (isL ? rowL : rowR)[col];
            (isL ? labL + buf * nL : labR + buf * nR)[idx] = make_float4(v.L, v.a, v.b, 0.f);
            if (!A.evol) (isL ? bgrL + buf * nL : bgrR + buf * nR)[idx] = v.bgrx;      // (only the in-kernel e tiles read them; not allocated otherwise)
        }
    };
    // tasks [t_lo, t_hi) of the e tile of window row i (task = tap column ul x pair of disparity groups, flat index
    // sp * nL + ul): e[ul][d] = min(40, |dB|+|dG|+|dR|) (_passive.cpp:77-79) from the staged bytes
    auto build_e = [&](int i, int t_lo, int t_hi) {
        int tidb = threadIdx.x;
        asm volatile("" : "+v"(tidb));
        unsigned char *const eT = eT0 + (i & 1) * g.e_bytes;
        const uint32_t *const bgrLc = bgrL + (i & 1) * nL, *const bgrRc = bgrR + (i & 1) * nR;
        const int e_q = nthr / nL, e_r = nthr - e_q * nL;
        int t = t_lo + tidb;
        int sp = t / nL, ul = t - sp * nL;
        while (t < t_hi) {
            const uint32_t lp = bgrLc[ul];
            const uint32_t *const rp = bgrRc + (ul + (Dc - 1) - 8 * sp);   // R[u-d] for d = dlo + 8*sp
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo |= min(__builtin_amdgcn_sad_u8(lp, rp[-k], 0u), 40u) << (8 * k);
                hi |= min(__builtin_amdgcn_sad_u8(lp, rp[-4 - k], 0u), 40u) << (8 * k);
            }
            uint32_t *const dst = reinterpret_cast<uint32_t *>(eT + ul * Se) + 2 * sp;
            dst[0] = lo;
            if (2 * sp + 1 < g.DG) dst[1] = hi;
            t += nthr; ul += e_r; sp += e_q;
            if (ul >= nL) { ul -= nL; ++sp; }
        }
    };
    // the e tile of window row i fetched from the pre-computed volume (A.evol) by LDS-DMA: the tile is ONE contiguous
    // block of nL columns x Se bytes there; every wave moves 1 KiB pieces (64 lanes x 16 B) straight into LDS
    auto load_e = [&](int i) {
        int tidd = threadIdx.x;
        asm volatile("" : "+v"(tidd));
        const int r = y - p + i;
        const unsigned char *const src = A.evol + (((size_t)blockIdx.z * A.erows + (r - A.erow0)) * (size_t)A.evolW + x0) * Se;
        unsigned char *const dst = eT0 + (i & 1) * g.e_bytes;
        const int bytes = nL * Se, lane16 = (tidd & 63) * 16;
        for (int k = __builtin_amdgcn_readfirstlane(tidd >> 6) * 1024; k < bytes; k += (nthr >> 6) * 1024)
            if (k + lane16 < bytes)
                __builtin_amdgcn_global_load_lds((const void *)(src + k + lane16),
                                                 (__attribute__((address_space(3))) void *)(dst + k), 16, 0, 0);
    };
    // support weights of window row i, tap columns [jb, je), into weight buffer rows rb.. (_passive.cpp:47-50, 71-74;
    // exp(-dist/gammaC) = exp2(dist*kC)).  One thread per window centre, all columns of the chunk: batches of ASW_WB
    // independent chains walking running pointers; tap columns outside the image are staged with L = +inf and get weight +0
    // (centres outside the image only feed candidates the winner-take-all never looks at).
    auto build_weights = [&](int i, int jb, int je, int rb) {
        int tidw = threadIdx.x;
        asm volatile("" : "+v"(tidw));
        const float4 *const labLc = labL + (i & 1) * nL, *const labRc = labR + (i & 1) * nR;
        const float *const prow = proxS + (i & 1) * win;
        const int ncen = Tx + nRc;
        // tasks: one centre per thread with all columns of the chunk while whole rounds of nthr centres last; the
        // centres left over (fewer than nthr) are cut into column segments so that the last round is spread over
        // the threads instead of leaving most of them idle behind a few (tiles with more centres than threads)
        const int full = ncen / nthr * nthr, rest = ncen - full;
        int slen = je - jb, nseg = 1;
        if (rest > 0 && full > 0) {
            slen = max(ASW_WB, (je - jb + nthr / rest - 1) / (nthr / rest));
            slen = (slen + ASW_WB - 1) / ASW_WB * ASW_WB;
            nseg = (je - jb + slen - 1) / slen;
        }
        const int ntasks = full + rest * nseg;
        for (int t = tidw; t < ntasks; t += nthr) {
            int c = t, jb_t = jb, je_t = je;
            if (t >= full && nseg > 1) {
                const int q = t - full, sgm = q / rest;
                c = full + q - sgm * rest;
                jb_t = jb + sgm * slen;
                je_t = min(je, jb_t + slen);
            }
            const bool isL = c < Tx;
            const int cc = isL ? c : c - Tx;
            const float4 cen = cenLab[c];
            const float4 *const seg = (isL ? labLc : labRc) + cc;
            const int stride = isL ? g.SL : SR;
            float *const wout = (isL ? wL : wR) + cc + (rb - jb) * stride;
            const int col0 = (isL ? x0 : xrc_lo) + cc - p;
            const uint32_t cmask = cen.w != 0.f ? 0xffffffffu : 0u;
            int j = jb_t;
            const float4 *sp = seg + j;
            const float *pp = prow + j;
            float *wp = wout + j * stride;
            int col = col0 + j;
            const int stride4 = ASW_WB * stride;
            for (; j + ASW_WB <= je_t; j += ASW_WB, sp += ASW_WB, pp += ASW_WB, wp += stride4, col += ASW_WB) {
                float4 tp[ASW_WB];
                float pr[ASW_WB], wv[ASW_WB];
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) { tp[u] = sp[u]; pr[u] = pp[u]; }
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) {
                    asm volatile("" ::"v"(tp[u].w));     // keeps the read a ds_read_b128 (4 LDS cycles; the 12-byte form takes 8)
                    const float dL = tp[u].x - cen.x, da = tp[u].y - cen.y, db = tp[u].z - cen.z;
                    wv[u] = fmaf(db, db, fmaf(da, da, dL * dL));
                }
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) wv[u] = __builtin_amdgcn_sqrtf(wv[u]);
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) wv[u] = asw_weight_finish(wv[u], A.kC, pr[u]);
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) {
                    if constexpr (SSAMD_PIPE_SENTINEL) {
                        wp[u * stride] = wv[u];
                    } else {
                        const uint32_t m = (unsigned)(col + u) < (unsigned)W ? cmask : 0u;
                        wp[u * stride] = __uint_as_float(__float_as_uint(wv[u]) & m);
                    }
                }
            }
            if (j < je_t) {
                float4 tp[ASW_WB];
                float pr[ASW_WB], wv[ASW_WB];
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) {
                    const int jj = min(j + u, je_t - 1);
                    tp[u] = seg[jj];
                    pr[u] = prow[jj];
                }
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) {
                    asm volatile("" ::"v"(tp[u].w));
                    const float dL = tp[u].x - cen.x, da = tp[u].y - cen.y, db = tp[u].z - cen.z;
                    wv[u] = fmaf(db, db, fmaf(da, da, dL * dL));
                }
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) wv[u] = __builtin_amdgcn_sqrtf(wv[u]);
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) wv[u] = asw_weight_finish(wv[u], A.kC, pr[u]);
#pragma unroll
                for (int u = 0; u < ASW_WB; ++u) {  // past the segment end the clamped tap is simply rewritten
                    const int jj = min(j + u, je_t - 1);
                    if constexpr (SSAMD_PIPE_SENTINEL) {
                        wout[jj * stride] = wv[u];
                    } else {
                        const uint32_t m = (unsigned)(col0 + jj) < (unsigned)W ? cmask : 0u;
                        wout[jj * stride] = __uint_as_float(__float_as_uint(wv[u]) & m);
                    }
                }
            }
        }
    };
    auto chunk_end = [&](int c) { return c == NC - 1 ? win : (c + 1) * JC; };
    const int nE = nL * ((g.DG + 1) >> 1);             // tasks of one e tile
    // everything chunk c of window row i can hide: see the file header.  cb = weight buffer chunk c reads.
    // (SSAMD_ABLATE_*: phase-ablation builds of tools/build_variants.sh, never defined in the product)
    auto build_next = [&](int i, int c, int cb) {
        const bool more_rows = i + 1 < i_hi;
#ifdef SSAMD_PIPE_BUILD_PRIO          // experiment (tools/build_variants.sh): the latency-bound build chains at a raised issue priority
        __builtin_amdgcn_s_setprio(SSAMD_PIPE_BUILD_PRIO);
#endif
#ifndef SSAMD_ABLATE_STAGE
        if (c == 0 && more_rows) stage_row(i + 1);
#endif
        // issued after the staged pixels are in LDS (their global loads are waited for with vmcnt(0), which would
        // also wait for these); lands under the chunk's taps, complete at the next barrier
        if (c == 0 && more_rows && A.evol) load_e(i + 1);
#ifndef SSAMD_ABLATE_WEIGHTS
        if (c + 1 < NC) build_weights(i, (c + 1) * JC, chunk_end(c + 1), (cb ^ 1) * g.JCmax);
        else if (more_rows) build_weights(i + 1, 0, chunk_end(0), (cb ^ 1) * g.JCmax);
#endif
#ifndef SSAMD_ABLATE_E
        if (c >= 1 && more_rows && !A.evol) build_e(i + 1, (int)((long long)nE * (c - 1) / (NC - 1)), (int)((long long)nE * c / (NC - 1)));
#endif
#ifdef SSAMD_PIPE_BUILD_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- prologue: first window row staged, its e tile and its first weight chunk built (not overlapped: 1 / win of the work)
    stage_row(i_lo);
    if (A.evol) load_e(i_lo);
    __syncthreads();
    if (!A.evol) build_e(i_lo, 0, nE);
    build_weights(i_lo, 0, chunk_end(0), 0);

    // Thread -> (column group xg, disparity group dg), lanes along the disparity groups.  Round 3: a tile at the left
    // image border only has candidates up to d = x (x - d >= 0, _passive.cpp:56), so its threads are dealt over the
    // DGe <= DG disparity groups that hold any: the waves beyond XG * DGe threads skip the tap loops altogether (x tile 0
    // of a 1080p / D 0..192 frame: 30 of 49 groups, 8 of 12 waves; the round-2 mapping kept every wave busy with dead
    // candidates there).  The LDS layout does not depend on the mapping.  (xg, dg) are worked out ONCE and carried in one
    // register: two integer divisions per window row were 1.3 % of the kernel's instructions.
    const int DGe = max(1, min(g.DG, (min(min(x0 + Tx, W) - 1, x0 + Tx - 1) - dlo) / ASW_RD + 1));
    const int nact = g.XG * DGe;
    int pk_xd;
    bool run;          // does this wave hold any candidate the reference evaluates?  (wave-uniform, see asw_aggregate_kernel)
    {
        const int tidm = threadIdx.x;
        int xg, dg;
#ifdef SSAMD_PIPE_PLAIN_LANES          // round-2 order (A/B builds of tools/build_variants.sh): thread = xg * DGe + dg
        xg = tidm / DGe; dg = tidm - xg * DGe;
#else
        // Lane order (round 3).  The LDS serves a ds_read_b128 in groups of 16 lanes and a ds_read_b32 in groups of 32;
        // with thread = xg * DGe + dg and DGe no multiple of 16 (49 at 1080p / D 0..192) a third of the 16-lane groups
        // straddle two column groups, whose right-weight blocks and e dwords then collide (rocprofv3: 23 % of the LDS
        // cycles were bank conflicts).  Here every 16-lane group holds 16 consecutive disparity groups of ONE column
        // group (F = DGe / 16 such groups per column group, in reversed order for odd column groups: the e rows of
        // neighbouring column groups are half the banks apart, so pairs of 16-lane groups stay on disjoint banks);
        // the DGe % 16 left-over disparity groups of all column groups follow at the end.
        const int F = DGe >> 4, Lo = DGe & 15, nfull = g.XG * F;
        const int grp16 = tidm >> 4, r16 = tidm & 15;
        if (grp16 < nfull) {
            xg = grp16 / F;
            int gi = grp16 - xg * F;
            if (xg & 1) gi = F - 1 - gi;
            dg = 16 * gi + r16;
        } else {
            const int q = tidm - 16 * nfull;
            xg = Lo ? q / Lo : g.XG;                 // (q >= XG * Lo: beyond the active threads)
            dg = 16 * F + (Lo ? q - xg * Lo : 0);
        }
#endif
        pk_xd = xg | (dg << 16);
        const bool tile_live = tidm < nact && x0 + RX * xg < W && dlo + ASW_RD * dg <= A.maxD &&
                               x0 + RX * xg + RX - 1 - (dlo + ASW_RD * dg) >= 0;
        run = __builtin_amdgcn_ballot_w64(tile_live) != 0 && tidm < nact;
    }

    int cb = 0;
#pragma nounroll
    for (int i = i_lo; i < i_hi; ++i) {
        const unsigned char *const eT = eT0 + (i & 1) * g.e_bytes;
        AswRow ew[RX];

#pragma nounroll                      // (NC is a compile-time 2 in the static instantiations: two copies of the tap loop cost VGPRs -> scratch)
        for (int c = 0; c < NC; ++c) {
#ifndef SSAMD_ABLATE_BARRIER          // (ablation build: how much of the time is waiting at this barrier; results are wrong without it)
            __syncthreads();       // buffers of chunk (i, c) complete; every wave is done with chunk (i, c) - 1
#endif
            const int jc = c * JC, jend = chunk_end(c);
            const int rb = cb * g.JCmax;
            // waves 0-3 (one per SIMD) build first, their two SIMD-mates aggregate first and build afterwards
            if (phase == 0) build_next(i, c, cb);
#ifndef SSAMD_ABLATE_AGG
            if (run) {
                // e-row and weight pointers are derived per chunk from the packed thread coordinates: nothing but those,
                // the e window and the accumulators stays live across a build
                const int xg = pk_xd & 0xffff, dg = pk_xd >> 16;
                // the next e row to load is ul0 + RX - 1 + jc (ul0 + 0 before the priming of the window row)
                const unsigned char *erow = eT + (RX * xg + (jc == 0 ? 0 : RX - 1 + jc)) * Se + 4 * dg;
                if (jc == 0) {
#pragma unroll
                    for (int n = 0; n < RX - 1; ++n) {
                        asw_row_unpack(ew[n], *reinterpret_cast<const uint32_t *>(erow));
                        erow += Se;
                    }
                }
                const float *wlp = wL + rb * g.SL + 8 * xg;
                const float *wrp = wR + rb * SR + (RX * xg - ASW_RD * dg + Dc - ASW_RD);
                for (int j0 = jc; j0 < jend; j0 += RX) {
#define SSAMD_PSTEP(JJ)                                                                             \
    if (j0 + (JJ) < jend) {                                                                         \
        const uint32_t epk = *reinterpret_cast<const uint32_t *>(erow + (STATIC ? (JJ) * SEC : 0)); \
        if constexpr (!STATIC) erow += Se;                                                          \
        float wl[RX], wr[NWR];                                                                      \
        {                                                                                           \
            const float *const wl_ = wlp + (STATIC ? (JJ) * SLC : 0);                              \
            const float *const wr_ = wrp + (STATIC ? (JJ) * SRC : 0);                              \
            const float4 v0 = *reinterpret_cast<const float4 *>(wl_);                              \
            wl[0] = v0.x; wl[1] = v0.y; wl[2] = v0.z; wl[3] = v0.w;                                 \
            const float4 v1 = *reinterpret_cast<const float4 *>(wl_ + 4);                          \
            wl[4] = v1.x; wl[5] = v1.y; wl[6] = v1.z; wl[7] = v1.w;                                 \
            const float4 r0 = *reinterpret_cast<const float4 *>(wr_);                              \
            const float4 r1 = *reinterpret_cast<const float4 *>(wr_ + 4);                          \
            const float4 r2 = *reinterpret_cast<const float4 *>(wr_ + 8);                          \
            asm volatile("" ::"v"(r2.w));      /* unused 12th weight: keeps the read a ds_read_b128 */ \
            wr[0] = r0.x; wr[1] = r0.y; wr[2] = r0.z; wr[3] = r0.w;                                 \
            wr[4] = r1.x; wr[5] = r1.y; wr[6] = r1.z; wr[7] = r1.w;                                 \
            wr[8] = r2.x; wr[9] = r2.y; wr[10] = r2.z; wr[11] = r2.w;                               \
        }                                                                                           \
        if constexpr (!STATIC) { wlp += g.SL; wrp += SR; }                                         \
        /* columns 0 .. RX-2 first: the newest e row (tap row j + RX - 1) is only used by the last column and is \
           unpacked into the registers of the oldest row once column 0 is done with that one */       \
        _Pragma("unroll") for (int xi = 0; xi < RX; ++xi) {                                         \
            if (xi == RX - 1) asw_row_unpack(ew[((JJ) + RX - 1) % RX], epk);                        \
            const AswRow &row_ = ew[((JJ) + xi) % RX];                                              \
            _Pragma("unroll") for (int di = 0; di < ASW_RD; ++di) {                                 \
                const float w_ = wl[xi] * wr[xi - di + ASW_RD - 1];                                 \
                accN[xi][di] = fmaf(w_, row_.e[di], accN[xi][di]);                                  \
                accS[xi][di] = fmaf(w_, row_.c[di], accS[xi][di]);                                  \
            }                                                                                       \
        }                                                                                           \
    }
                    SSAMD_PSTEP(0) SSAMD_PSTEP(1) SSAMD_PSTEP(2) SSAMD_PSTEP(3)
                    SSAMD_PSTEP(4) SSAMD_PSTEP(5) SSAMD_PSTEP(6) SSAMD_PSTEP(7)
#undef SSAMD_PSTEP
                    if constexpr (STATIC) { erow += RX * SEC; wlp += RX * SLC; wrp += RX * SRC; }
                }
            }
#endif
            if (phase != 0) build_next(i, c, cb);
            cb ^= 1;
        }
    }

    // ---- weighted average (_passive.cpp:88) and the two WTA reductions (as asw_aggregate_kernel)
    int tidf = threadIdx.x;
    asm volatile("" : "+v"(tidf));
    AswKeyTile<RX, ASW_RD> kt;                      // cost images of the register tile (exact mode re-reads them after the barrier)
    if (tidf < nact) {
        const int xg = pk_xd & 0xffff, dg = pk_xd >> 16;
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
        const int xg = pk_xd & 0xffff, dg = pk_xd >> 16;
        asw_exact_select<RX, ASW_RD>(A.xq, tidf < nact, kt, bestL + RX * xg, A.keyR ? bestR + (RX * xg - ASW_RD * dg + Dc - ASW_RD) : nullptr,
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

}  // namespace ssamd
