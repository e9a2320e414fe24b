// K1w: ASW aggregation for SMALL disparity ranges (the class default maxDisparity = 16; gfx950).
//
// Same algebra, same 8 x 4 register tile, same tap order -- therefore the same sums bit for bit -- as
// asw_aggregate_kernel / asw_aggregate_pipe_kernel; what changes is who builds the support weights.
//
// With few disparities the workgroup kernels starve: a tile of Tx columns x Dc disparities has only Tx * Dc / 32
// threads but needs 2 Tx + Dc weights per tap column in LDS, so at Dc = 20 the double-buffered weight rows allow two
// 4-wave groups per CU (two waves per SIMD), and 28 % of the instructions are support weights behind barriers
// (profiles/r02_asw_default_d16_rocprof_summary.txt: 10.1 ms for 1080p / D 0..16 / win 35 against ~4 ms of work).
//
// Here a WAVE is the unit.  Its 64 lanes are NXG = 64 / DG column groups x all DG disparity groups of one output row
// (lane = xg * DG + dg), i.e. a strip of 8 * NXG columns with the whole disparity range.  Per tap column the wave
//   1. evaluates the 2 * Txw + Dc - 1 support weights its own taps need -- one centre per lane, consecutive lanes
//      consecutive centres -- into a wave-private LDS row (no redundancy inside the strip: every weight is used by
//      all of the wave's disparities),
//   2. runs the 32 taps per lane of that column from it.
// LDS traffic from one wave is served in order, so step 2 sees step 1 without any workgroup barrier; waves never
// wait for each other (a workgroup is just four independent waves), and a wave needs ~13 KB of LDS (pixel row,
// centre pixels, e rows, one weight row), so twelve waves per CU stay resident at 163 VGPRs.
// e tiles come from the TAD volume of asw_tad_volume_kernel (LDS-DMA, one image row at a time).
// Used when the whole range fits one chunk of at most 16 disparity groups (nD <= 64 since round 3; 48 in round 2).
#pragma once
#include "asw_kernels.hip.h"
#include <type_traits>

namespace ssamd {

struct AswWaveGeom {
    int RX, DG, NXG, Txw, Dc, lanes;        // columns per lane (8 or 4); disparity groups, column groups and columns per wave, lanes in use
    int nLw, nRcw, nRw;                     // left tap columns, right centres, right tap columns of a wave's strip
    int SLw, SRw, Se;                       // floats per weight row (left / right part), bytes per e column
    int waves;                              // waves per workgroup
    int merged, K;                          // round 3: left and right centres in ONE list of K = ceil((Txw + nRcw) / 64) build rounds
    int RD;                                 // disparities per lane: 4, or 6 (asw_wave6_kernel.hip.h: 8-byte e slots, Se = 8 * odd)
    int creg;                               // round 4: 1 = the window centres live in registers (no cen array in LDS; straight-line merged build only)
    int off_w, off_cen, off_pixL, off_pixR, off_e, off_bestL, off_bestR;     // offsets inside a wave's LDS slice
    int wave_lds;                           // bytes of LDS per wave
};

struct AswWaveArgs {
    const PixRec *recL, *recR;
    const float *prox;
    u64 *keyL, *keyR;
    int16_t *disp;               // non-null: no right-referenced pass -> the wave writes the disparities itself
    float *costs;                // optional raw cost dump
    int cost_keys;               // 1: cost images instead of costs (AswArgs::cost_keys)
    const unsigned char *evol;   // TAD volume (required)
    int erow0, erows, evolW;
    int H, W, win, pad, minD, maxD, row0, rows, ystep;
    int yb0, yskip_at, yskip;    // first workgroup row; second row range (AswArgs::yb0, yskip)
    float kC;
    AswExactQueue xq;            // exact mode: near-tie queue (AswArgs::xq)
    AswWaveGeom g;
};

// wave-local ordering of LDS traffic: nothing may be moved across by the compiler, everything issued has completed
__device__ __forceinline__ void asw_wave_sync()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef float asw_v2f __attribute__((ext_vector_type(2)));

// LDS instructions of one wave execute in issue order, so a wave's ds_read sees its own earlier ds_write without
// waiting for it; only the compiler must keep the order.
__device__ __forceinline__ void asw_wave_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

#ifndef SSAMD_WAVE4_OCC          // experiment builds (tools/build_variants.sh): waves per SIMD of the 4-column tile,
#define SSAMD_WAVE4_OCC 4        // and whether its build keeps two rounds' reads in flight
#endif
#ifndef SSAMD_WAVE4_PAIR
#define SSAMD_WAVE4_PAIR 1
#endif
#ifndef SSAMD_WAVE8_OCC
#define SSAMD_WAVE8_OCC 3
#endif
#ifndef SSAMD_WAVE_PKMUL         // 1: the tap products as v_pk_mul_f32 pairs (measured: 1.5-4 % slower)
#define SSAMD_WAVE_PKMUL 0
#endif
// KL, KR: build rounds (64 centres each) of the left and of the right part when the host knows them at compile time
// (the build is then straight-line code with immediate offsets); 0: counted at run time.
// KM (round 3): rounds of the MERGED build -- the strip's Txw left and nRcw right centres as one list of Txw + nRcw
// entries dealt to the lanes 64 at a time, instead of ceil(Txw / 64) + ceil(nRcw / 64) rounds with two part-filled
// last rounds (class default D 0..16, 4-column tile: 48 + 67 centres = 2 rounds instead of 1 + 2).  The two parts
// read different pixel rows; pixR follows pixL in LDS, so the tap address of list entry c is
// pixL + 16 (c + j) + (c < Txw ? 0 : 32 pad): one per-lane constant per round (tapoff), set up once per wave.
// CREG (round 4, with KM > 0): the centres a lane evaluates are the SAME in every build of the kernel (list entries lane,
// lane + 64, ...), so their Lab values are loaded once from the records into 3 KM registers instead of being kept in an LDS
// array and re-read twice per tap-column pair: a third fewer LDS reads per weight, and the wave's LDS slice loses 16 bytes
// per centre -- at D 0..7 / win 35 (124-column strips, 255 centres) that is 13.9 -> 9.9 KB, 11 -> 16 resident waves per CU.
template <bool WITH_COSTS, int RX, int KL = 0, int KR = 0, int KM = 0, bool CREG = false>
__global__ __launch_bounds__(256, RX == 8 ? SSAMD_WAVE8_OCC : SSAMD_WAVE4_OCC) void asw_aggregate_wave_kernel(const AswWaveArgs A)
{
    static_assert(!CREG || KM > 0, "register centres need the straight-line merged build");
    constexpr int NWR = asw_nwr(RX);
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const AswWaveGeom &g = A.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *const smem = smem_all + wave * g.wave_lds;
    float *const wS = reinterpret_cast<float *>(smem + g.off_w);            // [SLw + SRw]: left weights, then right weights
    float4 *const cenLab = reinterpret_cast<float4 *>(smem + g.off_cen);    // [Txw + nRcw]
    float4 *const pixL = reinterpret_cast<float4 *>(smem + g.off_pixL);     // [nLw] Lab of the current image row
    float4 *const pixR = reinterpret_cast<float4 *>(smem + g.off_pixR);     // [nRw]
    unsigned char *const eT = reinterpret_cast<unsigned char *>(smem + g.off_e);   // [nLw][Se]
    u64 *const bestL = reinterpret_cast<u64 *>(smem + g.off_bestL);
    u64 *const bestR = reinterpret_cast<u64 *>(smem + g.off_bestR);

    const int W = A.W, win = A.win, p = A.pad;
    const int Txw = g.Txw, Dc = g.Dc, nLw = g.nLw, nRcw = g.nRcw, nRw = g.nRw, Se = g.Se;
    const int x0 = (blockIdx.x * g.waves + wave) * Txw;
    if (x0 >= W) return;                                         // (no workgroup barrier anywhere: waves are independent)
    const int y = asw_out_row(A, blockIdx.y);
    const int dlo = A.minD, dhi = dlo + Dc - 1;
    const size_t orow = (size_t)(y - A.row0) * W;
    if (min(x0 + Txw - 1, W - 1) - dlo < 0) {                   // no candidate the reference evaluates in this strip
        if (A.disp)
            for (int k = lane; k < Txw && x0 + k < W; k += 64) A.disp[orow + x0 + k] = (int16_t)(x0 + k);
        return;
    }
    const int segL_lo = x0 - p, xrc_lo = x0 - dhi, segR_lo = xrc_lo - p;
    const int ncen = Txw + nRcw;
    const int xg = lane / g.DG, dg = lane - xg * g.DG;
    const bool active = lane < g.lanes;

    float accN[RX][ASW_RD], accS[RX][ASW_RD];
#pragma unroll
    for (int a = 0; a < RX; ++a)
#pragma unroll
        for (int b = 0; b < ASW_RD; ++b) { accN[a][b] = 0.f; accS[a][b] = 0.f; }
    float cenx[CREG ? KM : 1], ceny[CREG ? KM : 1], cenz[CREG ? KM : 1];      // CREG: Lab of the centres lane, lane + 64, ... (row y)
    if constexpr (CREG) {
#pragma unroll
        for (int r = 0; r < KM; ++r) {
            const int c = 64 * r + lane;
            const bool isL = c < Txw;
            const int ccol = isL ? x0 + c : xrc_lo + (c - Txw);
            cenx[r] = ceny[r] = cenz[r] = 0.f;
            if (c < ncen && (unsigned)ccol < (unsigned)W) {
                const PixRec q = (isL ? A.recL : A.recR)[(size_t)y * W + ccol];
                cenx[r] = q.L; ceny[r] = q.a; cenz[r] = q.b;
            }
        }
    } else {
        for (int c = lane; c < ncen; c += 64) {                  // window centres (row y)
            const bool isL = c < Txw;
            const int ccol = isL ? x0 + c : xrc_lo + (c - Txw);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)ccol < (unsigned)W) {
                const PixRec q = (isL ? A.recL : A.recR)[(size_t)y * W + ccol];
                v = make_float4(q.L, q.a, q.b, 1.f);
            }
            cenLab[c] = v;
        }
    }
    // Support weights of one tap column: lane l evaluates the centres l, l + 64, ... of the left and of the right part.
    // A tap column outside the image has L = +inf and so a zero weight; centres outside the image only feed candidates
    // the winner-take-all never looks at.
    // Addresses are LDS byte offsets = a wave-uniform base (SGPR) + the lane's 16 * lane or 4 * lane: the only vector
    // registers the build keeps between steps are those two (pointers per array would not fit next to the accumulators).
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f *lds_v4;
    typedef __attribute__((address_space(3))) float *lds_f1;
    auto ld4 = [](uint32_t a) { const v4f v = *(lds_v4)a; return make_float4(v.x, v.y, v.z, v.w); };
    const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t lane16 = lane * 16, lane4 = lane * 4;
    // (no lane guards: reads up to 127 entries past a part's end stay inside the wave's LDS slice and the weight rows
    // are padded to whole rounds, so the surplus lanes of the last round write weights nobody reads)
    auto weight = [&](const float4 &ce, const float4 &tp, float pj) {
        const float dL = tp.x - ce.x, da = tp.y - ce.y, db = tp.z - ce.z;
        const float dist = __builtin_amdgcn_sqrtf(fmaf(db, db, fmaf(da, da, dL * dL)));
        return asw_weight_finish(dist, A.kC, pj);
    };
    // Two tap columns (j, j + 1) per build: a centre is read once for both, and the wave pays the LDS round trip of a
    // build once per two aggregation steps.  Weight row q = column parity, at wS + q * wrow.
    const int wrow = g.SLw + g.SRw;
    auto build_part = [&](auto rounds, uint32_t tap_b, uint32_t cen_b, uint32_t dst_b, int n, float pj0, float pj1) {
        asm volatile("" : "+s"(tap_b), "+s"(cen_b), "+s"(dst_b));        // opaque: base + lane sums are formed here, per build
        const uint32_t row1 = (uint32_t)wrow * 4;
        constexpr int K = decltype(rounds)::value;
        if constexpr (K > 0) {                           // round count known: one address per array, immediate offsets
            const uint32_t ca = cen_b + lane16, ta = tap_b + lane16, da = dst_b + lane4, db_ = da + row1;
            if constexpr ((RX == 4 || SSAMD_WAVE8_OCC == 2) && K <= 3) {           // registers to spare: all reads of the part in flight together
                float4 ce[K], ta_[K], tb[K];
#pragma unroll
                for (int r = 0; r < K; ++r) { ce[r] = ld4(ca + 1024 * r); ta_[r] = ld4(ta + 1024 * r); tb[r] = ld4(ta + 1024 * r + 16); }
#pragma unroll
                for (int r = 0; r < K; ++r) asm volatile("" ::"v"(ce[r].w), "v"(ta_[r].w), "v"(tb[r].w) : "memory");
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    *(lds_f1)(da + 256 * r) = weight(ce[r], ta_[r], pj0);
                    *(lds_f1)(db_ + 256 * r) = weight(ce[r], tb[r], pj1);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const float4 ce0 = ld4(ca + 1024 * r), ta0 = ld4(ta + 1024 * r), tb0 = ld4(ta + 1024 * r + 16);
                asm volatile("" ::"v"(ce0.w), "v"(ta0.w), "v"(tb0.w));   // keeps the reads ds_read_b128
                *(lds_f1)(da + 256 * r) = weight(ce0, ta0, pj0);
                *(lds_f1)(db_ + 256 * r) = weight(ce0, tb0, pj1);
            }
            return;
        }
        int k = 0;
        // 4-column tile: two rounds per trip, all six reads in flight together (the 8-column tile has no registers for that)
        if constexpr (RX == 4 && SSAMD_WAVE4_PAIR) for (; k + 64 < n; k += 128) {
            const uint32_t ca = cen_b + lane16 + k * 16, ta = tap_b + lane16 + k * 16, da = dst_b + lane4 + k * 4;
            const float4 ce0 = ld4(ca), ta0 = ld4(ta), tb0 = ld4(ta + 16);
            const float4 ce1 = ld4(ca + 1024), ta1 = ld4(ta + 1024), tb1 = ld4(ta + 1040);
            asm volatile("" ::"v"(ce0.w), "v"(ta0.w), "v"(tb0.w), "v"(ce1.w), "v"(ta1.w), "v"(tb1.w) : "memory");   // also keeps the reads ds_read_b128
            *(lds_f1)da = weight(ce0, ta0, pj0);
            *(lds_f1)(da + 256) = weight(ce1, ta1, pj0);
            *(lds_f1)(da + row1) = weight(ce0, tb0, pj1);
            *(lds_f1)(da + row1 + 256) = weight(ce1, tb1, pj1);
        }
        for (; k < n; k += 64) {                         // single rounds
            const uint32_t ca = cen_b + lane16 + k * 16, ta = tap_b + lane16 + k * 16, da = dst_b + lane4 + k * 4;
            const float4 ce0 = ld4(ca), ta0 = ld4(ta), tb0 = ld4(ta + 16);
            asm volatile("" ::"v"(ce0.w), "v"(ta0.w), "v"(tb0.w) : "memory");
            *(lds_f1)da = weight(ce0, ta0, pj0);
            *(lds_f1)(da + row1) = weight(ce0, tb0, pj1);
        }
    };
    // merged build: per-lane tap offsets of the compile-time rounds (see the template comment)
    uint32_t tapoff[KM > 0 ? KM : 1];
#pragma unroll
    for (int r = 0; r < (KM > 0 ? KM : 1); ++r) {
        const int c = 64 * r + lane;
        tapoff[r] = 16u * (uint32_t)c + (c < Txw ? 0u : 32u * (uint32_t)p);
    }
    auto build_merged = [&](uint32_t tap_b, uint32_t cen_b, uint32_t dst_b, float pj0, float pj1) {
        asm volatile("" : "+s"(tap_b), "+s"(cen_b), "+s"(dst_b));
        const uint32_t row1 = (uint32_t)wrow * 4;
        if constexpr (CREG) {                            // centres in registers: two tap reads per weight pair, all in flight together
            const uint32_t da = dst_b + lane4, db_ = da + row1;
            float4 ta_[KM], tb[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                const uint32_t ta = tap_b + tapoff[r];
                ta_[r] = ld4(ta); tb[r] = ld4(ta + 16);
            }
#pragma unroll
            for (int r = 0; r < KM; ++r) asm volatile("" ::"v"(ta_[r].w), "v"(tb[r].w) : "memory");
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                const float4 ce = make_float4(cenx[r], ceny[r], cenz[r], 0.f);
                *(lds_f1)(da + 256 * r) = weight(ce, ta_[r], pj0);
                *(lds_f1)(db_ + 256 * r) = weight(ce, tb[r], pj1);
            }
            return;
        }
        if constexpr (KM > 0) {
            const uint32_t ca = cen_b + lane16, da = dst_b + lane4, db_ = da + row1;
            if constexpr (RX == 4 && KM <= 3) {          // registers to spare: all reads of the build in flight together
                float4 ce[KM], ta_[KM], tb[KM];
#pragma unroll
                for (int r = 0; r < KM; ++r) {
                    const uint32_t ta = tap_b + tapoff[r];
                    ce[r] = ld4(ca + 1024 * r); ta_[r] = ld4(ta); tb[r] = ld4(ta + 16);
                }
#pragma unroll
                for (int r = 0; r < KM; ++r) asm volatile("" ::"v"(ce[r].w), "v"(ta_[r].w), "v"(tb[r].w) : "memory");
#pragma unroll
                for (int r = 0; r < KM; ++r) {
                    *(lds_f1)(da + 256 * r) = weight(ce[r], ta_[r], pj0);
                    *(lds_f1)(db_ + 256 * r) = weight(ce[r], tb[r], pj1);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                const uint32_t ta = tap_b + tapoff[r];
                const float4 ce0 = ld4(ca + 1024 * r), ta0 = ld4(ta), tb0 = ld4(ta + 16);
                asm volatile("" ::"v"(ce0.w), "v"(ta0.w), "v"(tb0.w));   // keeps the reads ds_read_b128
                *(lds_f1)(da + 256 * r) = weight(ce0, ta0, pj0);
                *(lds_f1)(db_ + 256 * r) = weight(ce0, tb0, pj1);
            }
            return;
        }
        for (int k = 0; k < ncen; k += 64) {              // rounds counted at run time
            const int c = k + lane;
            const uint32_t ca = cen_b + lane16 + k * 16, ta = tap_b + lane16 + k * 16 + (c < Txw ? 0u : 32u * (uint32_t)p),
                           da = dst_b + lane4 + k * 4;
            const float4 ce0 = ld4(ca), ta0 = ld4(ta), tb0 = ld4(ta + 16);
            asm volatile("" ::"v"(ce0.w), "v"(ta0.w), "v"(tb0.w) : "memory");
            *(lds_f1)da = weight(ce0, ta0, pj0);
            *(lds_f1)(da + row1) = weight(ce0, tb0, pj1);
        }
    };
    auto build = [&](int j, float pj0, float pj1) {
        if (KM > 0 || (KL == 0 && KR == 0 && g.merged)) {
            build_merged(sbase + g.off_pixL + 16 * j, sbase + g.off_cen, sbase + g.off_w, pj0, pj1);
            return;
        }
        build_part(std::integral_constant<int, KL>{}, sbase + g.off_pixL + 16 * j, sbase + g.off_cen, sbase + g.off_w, Txw, pj0, pj1);
        build_part(std::integral_constant<int, KR>{}, sbase + g.off_pixR + 16 * j, sbase + g.off_cen + 16 * Txw, sbase + g.off_w + 4 * g.SLw,
                   nRcw, pj0, pj1);
    };

    const int i_lo = max(0, p - y), i_hi = min(win, A.H + p - y);
    int proxv = 0;
    for (int i = i_lo; i < i_hi; ++i) {
        const int r = y - p + i;
        asw_wave_sync();                 // the previous window row's taps are done with the pixel and e rows
        // ---- this image row: e tile by LDS-DMA (one contiguous block of the volume), Lab of the tap columns
#ifdef SSAMD_WABLATE_STAGE
        if (i == i_lo)
#endif
        {
            const unsigned char *const src = A.evol + (((size_t)(r - A.erow0)) * (size_t)A.evolW + x0) * Se;
            const int bytes = nLw * Se;
            for (int k = 0; k < bytes; k += 1024)
                if (k + lane * 16 < bytes)
                    __builtin_amdgcn_global_load_lds((const void *)(src + k + lane * 16),
                                                     (__attribute__((address_space(3))) void *)(eT + k), 16, 0, 0);
            proxv = __builtin_bit_cast(int, A.prox[i * win + min(lane, win - 1)]);   // lane j: proximity weight of tap column j
            const PixRec *const rowL = A.recL + (size_t)r * W, *const rowR = A.recR + (size_t)r * W;
            for (int k = lane; k < nLw + nRw; k += 64) {
                const bool isL = k < nLw;
                const int idx = isL ? k : k - nLw;
                const int col = (isL ? segL_lo : segR_lo) + idx;
                // a tap column outside the image gets L = +inf: its colour distance is +inf, exp2(-inf) = +0 and the
                // weight is exactly the +0 the other kernels produce with a mask, without an instruction for it
                float4 v = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);
                if ((unsigned)col < (unsigned)W) {
                    const PixRec q = (isL ? rowL : rowR)[col];
                    v = make_float4(q.L, q.a, q.b, 0.f);
                }
                (isL ? pixL : pixR)[idx] = v;
            }
        }
        asw_wave_sync();
        // e window: rows ul = RX xg + n of the tile, dword dg
        const unsigned char *erow = eT + (RX * xg) * Se + 4 * dg;
        AswRow ew[RX];
#pragma unroll
        for (int n = 0; n < RX - 1; ++n) {
            asw_row_unpack(ew[n], *reinterpret_cast<const uint32_t *>(erow));
            erow += Se;
        }
        const float *const wlp = wS + RX * xg;
        const float *const wrp = wS + g.SLw + (RX * xg - ASW_RD * dg + Dc - ASW_RD);

        // (SSAMD_WABLATE_*: phase-ablation builds of tools/build_variants.sh, never defined in the product)
#ifdef SSAMD_WABLATE_TAPS
#define SSAMD_WAVE_TAPS_IF if (i == i_lo && j0 == 0)
#else
#define SSAMD_WAVE_TAPS_IF
#endif
#ifdef SSAMD_WABLATE_BUILD
#define SSAMD_WAVE_BUILD(J) if (i == i_lo && (J) == 0) build(J, __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, J)), \
                                                              __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, (J) + 1)));
#else
#define SSAMD_WAVE_BUILD(J) build(J, __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, J)), \
                                  __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, (J) + 1)));
#endif
        for (int j0 = 0; j0 < win; j0 += RX) {
#define SSAMD_WSTEP(JJ)                                                                             \
    if (j0 + (JJ) < win) {                                                                          \
        const int j = j0 + (JJ);                                                                    \
        /* 1. even j: the support weights of tap columns j, j + 1 for the strip's centres (_passive.cpp:47-50, 71-74) */ \
        if (((JJ) & 1) == 0) { asw_wave_order(); SSAMD_WAVE_BUILD(j) asw_wave_order(); }           \
        const float *const wl_ = wlp + ((JJ) & 1) * wrow, *const wr_ = wrp + ((JJ) & 1) * wrow;     \
        /* 2. the taps of column j (lanes past the last column group read inside the slice and are ignored) */ \
        SSAMD_WAVE_TAPS_IF {                                                                        \
            const uint32_t epk = *reinterpret_cast<const uint32_t *>(erow);                         \
            erow += Se;                                                                             \
            float wl[RX], wr[NWR];                                                                  \
            {                                                                                       \
                const float4 v0 = *reinterpret_cast<const float4 *>(wl_);                          \
                wl[0] = v0.x; wl[1] = v0.y; wl[2] = v0.z; wl[3] = v0.w;                             \
                if constexpr (RX == 8) {                                                            \
                    const float4 v1 = *reinterpret_cast<const float4 *>(wl_ + 4);                  \
                    wl[RX - 4] = v1.x; wl[RX - 3] = v1.y; wl[RX - 2] = v1.z; wl[RX - 1] = v1.w;     \
                }                                                                                   \
                const float4 r0 = *reinterpret_cast<const float4 *>(wr_);                          \
                const float4 r1 = *reinterpret_cast<const float4 *>(wr_ + 4);                      \
                asm volatile("" ::"v"(r1.w));                                                       \
                wr[0] = r0.x; wr[1] = r0.y; wr[2] = r0.z; wr[3] = r0.w;                             \
                wr[4] = r1.x; wr[5] = r1.y; wr[6] = r1.z; wr[7] = r1.w;                             \
                if constexpr (RX == 8) {                                                            \
                    const float4 r2 = *reinterpret_cast<const float4 *>(wr_ + 8);                  \
                    asm volatile("" ::"v"(r2.w));                                                   \
                    wr[NWR - 4] = r2.x; wr[NWR - 3] = r2.y; wr[NWR - 2] = r2.z; wr[NWR - 1] = r2.w; \
                }                                                                                   \
            }                                                                                       \
            _Pragma("unroll") for (int xi = 0; xi < RX; xi += 2) {                                  \
                /* the products of two columns; (xi, di) and (xi + 1, di + 1) share the right weight: with SSAMD_WAVE_PKMUL */ \
                /* one v_pk_mul_f32 with a broadcast operand each (same IEEE products, 5 instead of 8 instructions) */ \
                float w_[2][ASW_RD];                                                                \
                _Pragma("unroll") for (int di = 0; di + 1 < ASW_RD; ++di) {                         \
                    const float s_ = wr[xi - di + ASW_RD - 1];                                      \
                    if constexpr (SSAMD_WAVE_PKMUL) {                                               \
                        const asw_v2f a_ = {wl[xi], wl[xi + 1]};                                    \
                        const asw_v2f b_ = {s_, s_};                                                \
                        const asw_v2f p_ = a_ * b_;                                                 \
                        w_[0][di] = p_.x; w_[1][di + 1] = p_.y;                                     \
                    } else {                                                                        \
                        w_[0][di] = wl[xi] * s_; w_[1][di + 1] = wl[xi + 1] * s_;                   \
                    }                                                                               \
                }                                                                                   \
                w_[0][ASW_RD - 1] = wl[xi] * wr[xi];                                                \
                w_[1][0] = wl[xi + 1] * wr[xi + ASW_RD];                                            \
                _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                     \
                    if (xi + u == RX - 1) asw_row_unpack(ew[((JJ) + RX - 1) % RX], epk);            \
                    const AswRow &row_ = ew[((JJ) + xi + u) % RX];                                  \
                    _Pragma("unroll") for (int di = 0; di < ASW_RD; ++di) {                         \
                        accN[xi + u][di] = fmaf(w_[u][di], row_.e[di], accN[xi + u][di]);           \
                        accS[xi + u][di] = fmaf(w_[u][di], row_.c[di], accS[xi + u][di]);           \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }
            SSAMD_WSTEP(0) SSAMD_WSTEP(1) SSAMD_WSTEP(2) SSAMD_WSTEP(3)
            if constexpr (RX == 8) { SSAMD_WSTEP(4) SSAMD_WSTEP(5) SSAMD_WSTEP(6) SSAMD_WSTEP(7) }
#undef SSAMD_WSTEP
#undef SSAMD_WAVE_BUILD
#undef SSAMD_WAVE_TAPS_IF
        }
    }

    // ---- weighted average (_passive.cpp:88) and the two winner-take-all reductions, inside the wave
    asw_wave_sync();
    for (int k = lane; k < Txw; k += 64) bestL[k] = KEY_NONE;        // (these share the pixel rows' space)
    for (int k = lane; k <= nRcw; k += 64) bestR[k] = KEY_NONE;
    asw_wave_sync();
    AswKeyTile<RX, ASW_RD> kt;                      // cost images of the register tile (exact mode re-reads them after the wave's barrier)
    if (active) {
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
                        A.costs[(orow + x) * (A.maxD - A.minD + 1) + (d - A.minD)] = A.cost_keys ? __uint_as_float((uint32_t)(hi >> 32)) : c;
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
    asw_wave_sync();
    const bool xq = !WITH_COSTS && A.xq.entries != nullptr;          // exact mode: near-ties of the winners go to the fp64 pass's queue
    if (xq)
        asw_exact_select<RX, ASW_RD>(A.xq, active, kt, bestL + RX * xg, A.keyR ? bestR + (RX * xg - ASW_RD * dg + Dc - ASW_RD) : nullptr,
                                 x0 + RX * xg, dlo + ASW_RD * dg, (uint32_t)orow);
    if (A.disp) {
        for (int k = lane; k < Txw; k += 64) {
            const int x = x0 + k;
            if (x < W) A.disp[orow + x] = bestL[k] == KEY_NONE ? (int16_t)x : (int16_t)(uint32_t)bestL[k];
        }
        return;
    }
    if (xq) {
        // strip-local winners meet the pixels' running minima: the loser of each meeting is queued if it is a near-tie
        for (int k0 = 0; k0 < Txw; k0 += 64) {
            const int k = k0 + lane, x = x0 + k;
            const bool have = k < Txw && x < W && bestL[k < Txw ? k : 0] != KEY_NONE;
            const u64 mine = have ? bestL[k] : KEY_NONE;
            const u64 old = have ? atomicMin(&A.keyL[orow + x], mine) : KEY_NONE;
            asw_exact_merge<false>(A.xq, have, mine, old, (uint32_t)orow, x);
        }
        if (A.keyR)
            for (int k0 = 0; k0 < nRcw; k0 += 64) {
                const int k = k0 + lane, xr = xrc_lo + k;
                const bool have = k < nRcw && (unsigned)xr < (unsigned)W && bestR[k < nRcw ? k : 0] != KEY_NONE;
                const u64 mine = have ? bestR[k] : KEY_NONE;
                const u64 old = have ? atomicMin(&A.keyR[orow + xr], mine) : KEY_NONE;
                asw_exact_merge<true>(A.xq, have, mine, old, (uint32_t)orow, xr);
            }
        return;
    }
    for (int k = lane; k < Txw; k += 64) {
        const int x = x0 + k;
        if (x < W && bestL[k] != KEY_NONE) atomicMin(&A.keyL[orow + x], bestL[k]);
    }
    if (A.keyR) {
        for (int k = lane; k < nRcw; k += 64) {
            const int xr = xrc_lo + k;
            if ((unsigned)xr < (unsigned)W && bestR[k] != KEY_NONE) atomicMin(&A.keyR[orow + xr], bestR[k]);
        }
    }
}

}  // namespace ssamd
