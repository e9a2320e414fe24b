// K1w6: asw_aggregate_wave_kernel with SIX disparities per lane (4 columns x 6 disparities), round 3 (gfx950).
//
// Every BASELINE disparity range is 2^k + 1 values (maxDisparity is inclusive); the class default of both reference
// classes and of the reference's only example, maxDisparity = 16, is 17.  With four disparities per lane that needs five
// disparity groups = 20 padded disparities (15 % of the tap work on candidates that do not exist) and 12 column groups
// x 5 = 60 of 64 lanes.  Three groups of SIX cover 18, 21 column groups x 3 = 63 lanes, and a strip of 84 columns
// builds its 84 + 101 support-weight centres in three merged rounds where the 48-column strip needs two: per column
// of a tap column (3 x 12 + 87) / 84 = 1.46 issue slots instead of (2 x 12 + 59) / 48 = 1.73.
//
// Same algebra, same tap order per (x, d) -- window rows, then tap columns -- therefore the same sums bit for bit as the
// other ASW kernels (tests/test_gpu_asw.py compares them).  What differs from asw_aggregate_wave_kernel:
//   * e tile: one 8-byte slot per disparity group and tap column (bytes 0..5 = the six truncated differences, written by
//     asw_tad_volume_kernel in its rd = 6 mode), read with one ds_read_b64;
//   * right weights: nine consecutive floats from an 8-byte-aligned address (the groups are six apart), five ds_read_b64;
//   * the build is always the merged one (one list of left and right centres, asw_wave_kernel.hip.h).
// Shares AswWaveArgs / AswWaveGeom (RD = 6, Se = bytes per e column) and the host path of the wave kernel.
#pragma once
#include "asw_wave_kernel.hip.h"

namespace ssamd {

struct AswRow6 {
    float e[6], c[6];
};

__device__ __forceinline__ void asw_row_unpack6(AswRow6 &row, const uint2 packed)
{
#pragma unroll
    for (int di = 0; di < 6; ++di) {
        const uint32_t wrd = di < 4 ? packed.x : packed.y;
        const float e = (float)((wrd >> (8 * (di & 3))) & 0xffu);   // v_cvt_f32_ubyteN
        row.e[di] = e;
        row.c[di] = ASW_TAD_CAP - e;
    }
}

// KM: build rounds known at compile time (3 for the class default), 0: counted at run time
// CREG: window centres in registers (see asw_aggregate_wave_kernel): no cen array in LDS, two instead of three reads per weight pair.
// (Round 4 also built a 128-VGPR form -- four waves per SIMD, build rounds one after the other -- and an e tile without its spare
//  slot, Se = 24: the 96 registers of accumulators and e window leave too little, 6.49 vs 6.07 ms at 1080p / D 0..16 and 4.15 vs
//  3.45 ms with register centres; Se = 24 bought a resident wave at win >= 31 and cost 10-16 % below.  profiles/r04_wave6_*.txt.)
template <bool WITH_COSTS, int KM, bool CREG = false>
__global__ __launch_bounds__(256, 3) void asw_aggregate_wave6_kernel(const AswWaveArgs A)
{
    static_assert(!CREG || KM > 0, "register centres need the straight-line build");
    constexpr int RX = 4, RD = 6, NWR = 10;          // nine right weights used, read as five pairs
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const AswWaveGeom &g = A.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *const smem = smem_all + wave * g.wave_lds;
    float *const wS = reinterpret_cast<float *>(smem + g.off_w);
    float4 *const cenLab = reinterpret_cast<float4 *>(smem + g.off_cen);
    float4 *const pixL = reinterpret_cast<float4 *>(smem + g.off_pixL);
    float4 *const pixR = reinterpret_cast<float4 *>(smem + g.off_pixR);
    unsigned char *const eT = reinterpret_cast<unsigned char *>(smem + g.off_e);
    u64 *const bestL = reinterpret_cast<u64 *>(smem + g.off_bestL);
    u64 *const bestR = reinterpret_cast<u64 *>(smem + g.off_bestR);

    const int W = A.W, win = A.win, p = A.pad;
    const int Txw = g.Txw, Dc = g.Dc, nLw = g.nLw, nRcw = g.nRcw, nRw = g.nRw, Se = g.Se;
    const int x0 = (blockIdx.x * g.waves + wave) * Txw;
    if (x0 >= W) return;
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

    float accN[RX][RD], accS[RX][RD];
#pragma unroll
    for (int a = 0; a < RX; ++a)
#pragma unroll
        for (int b = 0; b < RD; ++b) { accN[a][b] = 0.f; accS[a][b] = 0.f; }
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
    // merged support-weight build of two tap columns (j, j + 1): see asw_aggregate_wave_kernel
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f *lds_v4;
    typedef __attribute__((address_space(3))) float *lds_f1;
    auto ld4 = [](uint32_t a) { const v4f v = *(lds_v4)a; return make_float4(v.x, v.y, v.z, v.w); };
    const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t lane16 = lane * 16, lane4 = lane * 4;
    auto weight = [&](const float4 &ce, const float4 &tp, float pj) {
        const float dL = tp.x - ce.x, da = tp.y - ce.y, db = tp.z - ce.z;
        const float dist = __builtin_amdgcn_sqrtf(fmaf(db, db, fmaf(da, da, dL * dL)));
        return asw_weight_finish(dist, A.kC, pj);
    };
    const int wrow = g.SLw + g.SRw;
    uint32_t tapoff[KM > 0 ? KM : 1];
#pragma unroll
    for (int r = 0; r < (KM > 0 ? KM : 1); ++r) {
        const int c = 64 * r + lane;
        tapoff[r] = 16u * (uint32_t)c + (c < Txw ? 0u : 32u * (uint32_t)p);
    }
    auto build = [&](int j, float pj0, float pj1) {
        uint32_t tap_b = sbase + g.off_pixL + 16 * j, cen_b = sbase + g.off_cen, dst_b = sbase + g.off_w;
        asm volatile("" : "+s"(tap_b), "+s"(cen_b), "+s"(dst_b));
        const uint32_t row1 = (uint32_t)wrow * 4;
        if constexpr (CREG) {
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

    const int i_lo = max(0, p - y), i_hi = min(win, A.H + p - y);
    int proxv = 0;
    for (int i = i_lo; i < i_hi; ++i) {
        const int r = y - p + i;
        asw_wave_sync();                 // the previous window row's taps are done with the pixel and e rows
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
                float4 v = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);      // outside the image: weight +0 (see the wave kernel)
                if ((unsigned)col < (unsigned)W) {
                    const PixRec q = (isL ? rowL : rowR)[col];
                    v = make_float4(q.L, q.a, q.b, 0.f);
                }
                (isL ? pixL : pixR)[idx] = v;
            }
        }
        asw_wave_sync();
        // e window: rows ul = RX xg + n of the tile, 8-byte slot dg
        // (lanes past the last column group -- lane 63 of the 21 x 3 class-default strip -- read the last group's e rows: the tile has no slack behind it)
        const unsigned char *erow = eT + (RX * min(xg, g.NXG - 1)) * Se + 8 * dg;
        AswRow6 ew[RX];
#pragma unroll
        for (int n = 0; n < RX - 1; ++n) {
            asw_row_unpack6(ew[n], *reinterpret_cast<const uint2 *>(erow));
            erow += Se;
        }
        const float *const wlp = wS + RX * xg;
        const float *const wrp = wS + g.SLw + (RX * xg - RD * dg + Dc - RD);

        for (int j0 = 0; j0 < win; j0 += RX) {
#define SSAMD_W6STEP(JJ)                                                                            \
    if (j0 + (JJ) < win) {                                                                          \
        const int j = j0 + (JJ);                                                                    \
        if (((JJ) & 1) == 0) {                                                                      \
            asw_wave_order();                                                                       \
            build(j, __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, j)),                \
                  __builtin_bit_cast(float, __builtin_amdgcn_readlane(proxv, j + 1)));              \
            asw_wave_order();                                                                       \
        }                                                                                           \
        const float *const wl_ = wlp + ((JJ) & 1) * wrow, *const wr_ = wrp + ((JJ) & 1) * wrow;     \
        const uint2 epk = *reinterpret_cast<const uint2 *>(erow);                                   \
        erow += Se;                                                                                 \
        float wl[RX], wr[NWR];                                                                      \
        {                                                                                           \
            const float4 v0 = *reinterpret_cast<const float4 *>(wl_);                              \
            wl[0] = v0.x; wl[1] = v0.y; wl[2] = v0.z; wl[3] = v0.w;                                 \
            _Pragma("unroll") for (int k = 0; k < NWR / 2; ++k) {                                   \
                const float2 rr = *reinterpret_cast<const float2 *>(wr_ + 2 * k);                   \
                wr[2 * k] = rr.x; wr[2 * k + 1] = rr.y;                                             \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int xi = 0; xi < RX; ++xi) {                                         \
            if (xi == RX - 1) asw_row_unpack6(ew[((JJ) + RX - 1) % RX], epk);                       \
            const AswRow6 &row_ = ew[((JJ) + xi) % RX];                                             \
            _Pragma("unroll") for (int di = 0; di < RD; ++di) {                                     \
                const float w_ = wl[xi] * wr[xi - di + RD - 1];                                     \
                accN[xi][di] = fmaf(w_, row_.e[di], accN[xi][di]);                                  \
                accS[xi][di] = fmaf(w_, row_.c[di], accS[xi][di]);                                  \
            }                                                                                       \
        }                                                                                           \
    }
            SSAMD_W6STEP(0) SSAMD_W6STEP(1) SSAMD_W6STEP(2) SSAMD_W6STEP(3)
#undef SSAMD_W6STEP
        }
    }

    // ---- weighted average (_passive.cpp:88) and the two winner-take-all reductions, inside the wave
    asw_wave_sync();
    for (int k = lane; k < Txw; k += 64) bestL[k] = KEY_NONE;        // (these share the pixel rows' space)
    for (int k = lane; k <= nRcw; k += 64) bestR[k] = KEY_NONE;
    asw_wave_sync();
    AswKeyTile<RX, RD> kt;                      // cost images of the register tile (exact mode re-reads them after the wave's barrier)
    if (active) {
        u64 diag[RX + RD - 1];
#pragma unroll
        for (int k = 0; k < RX + RD - 1; ++k) diag[k] = KEY_NONE;
#pragma unroll
        for (int xi = 0; xi < RX; ++xi) {
            const int x = x0 + RX * xg + xi;
            u64 bl = KEY_NONE;
#pragma unroll
            for (int di = 0; di < RD; ++di) {
                const int d = dlo + RD * dg + di;
                const bool valid = (x < W) && (d <= A.maxD) && (x - d >= 0);
                kt.v[xi][di] = 0xffffffffu;
                if (valid) {
                    float c;
                    const u64 hi = (u64)asw_cost_key(accN[xi][di], accS[xi][di], c) << 32;
                    kt.v[xi][di] = (uint32_t)(hi >> 32);
                    bl = min(bl, hi | (u64)(uint32_t)d);
                    diag[xi - di + RD - 1] = min(diag[xi - di + RD - 1], hi | (u64)(uint32_t)x);
                    if (WITH_COSTS)
                        A.costs[(orow + x) * (A.maxD - A.minD + 1) + (d - A.minD)] = A.cost_keys ? __uint_as_float((uint32_t)(hi >> 32)) : c;
                }
            }
            if (bl != KEY_NONE) atomicMin(&bestL[RX * xg + xi], bl);
        }
        if (A.keyR) {
            const int base = RX * xg - RD * dg + Dc - RD;
#pragma unroll
            for (int k = 0; k < RX + RD - 1; ++k)
                if (diag[k] != KEY_NONE) atomicMin(&bestR[base + k], diag[k]);
        }
    }
    asw_wave_sync();
    const bool xq = !WITH_COSTS && A.xq.entries != nullptr;          // exact mode: near-ties of the winners go to the fp64 pass's queue
    if (xq)
        asw_exact_select<RX, RD>(A.xq, active, kt, bestL + RX * xg, A.keyR ? bestR + (RX * xg - RD * dg + Dc - RD) : nullptr,
                                 x0 + RX * xg, dlo + RD * dg, (uint32_t)orow);
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
