// K3/K4: GSW support weights + cost aggregation + winner-take-all keys (gfx950).
//
// Replaces workerGSW of the reference (_passive.cpp:408-700):
//   * support weights (left pass 433-496, right pass 555-618).  The reference's in-place
//     raster relaxation over the window collapses to a closed form (checked
//     bit-exactly against the reference's outputs by the oracle tests): every window cell the
//     loops reach ends at w = expf(-fl32(||BGR(cell) - BGR(centre)||) / gamma); cells outside
//     the image, or never reached, end at exp(-inf) = 0.  Reached cells: all in-image cells,
//     EXCEPT in the left-referenced pass when the window sticks out on the right
//     (x + pad >= W), where the reference's `break`s (446, 454, 471, 479) leave only the
//     centre -- plus, on image row 0, the in-image cells of the centre row.
//     ||.|| is the square root of an integer s <= 3*255^2, so the weight is a function of s:
//     the host tabulates expf(-(float)sqrt((double)s)/gamma) once per gamma (gswTab), which
//     makes the weights bit-identical to the reference's libm results.
//   * cost(x,d) = sum over in-image taps, in window raster order, of w * min(fMax, ||dBGR||)
//     with an fp32 running sum and UNFUSED multiply/add (513-535, 629-651): reproduced
//     operation by operation, so costs and therefore argmins are bit-exact.
//   * the same kernel serves both passes: reference image / target image swap roles and the
//     target column is ref_col - d (left-referenced) or ref_col + d (right-referenced).
//
// Work decomposition mirrors the ASW kernel: workgroup = (TY consecutive output rows, tile of Tx
// reference columns, chunk of Dc disparities); thread = TY rows x 4 columns x RD disparities.  Per IMAGE
// row the group builds e[u][d] (fp32) once in LDS and the support weights w[t][j][x] of each of its
// output rows; e rows slide through registers and feed the accumulators of all TY output rows.  The
// e tile is the expensive part (a correctly rounded sqrt per element), and an image row serves
// up to win window rows of different outputs: TY = 2 halves the number of times it is rebuilt.  The
// two instantiated tiles, (TY, RD) = (1, 8) and (2, 4), have the same accumulator count (32) and the
// same LDS read volume per tap (three 16-byte reads per 32 multiply-add pairs).
#pragma once
#include "common.hip.h"

namespace ssamd {

static constexpr int GSW_RX = 4;
static constexpr int GSW_MAX_THREADS = 512;
static constexpr int GSW_TAB_SIZE = 3 * 255 * 255 + 1;

struct GswGeom {
    int Tx, XG, DG, Dc, nchunks, threads, Ty, Rd;   // thread tile: Ty output rows x 4 columns x Rd disparities
    int nL, nT, Se, emask;          // Se: floats per e row (32-byte slots, XOR-swizzled)
    int off_w, off_e, off_ref, off_tgt, off_best;
    int lds_bytes;
};

struct GswArgs {
    const uint32_t *ref, *tgt;   // [H][W] packed B | G<<8 | R<<16
    const float *tab;            // [GSW_TAB_SIZE] support weight as a function of s = |dBGR|^2
    u64 *key;                    // [rows][W] WTA keys of this pass
    int H, W, win, pad, minD, maxD, row0, rows;
    int right;                   // 0: left-referenced (target col = ref col - d), 1: right-referenced (+d)
    int iterations;
    float fMax;
    GswGeom g;
};

// fl32(sqrt(s)) for an INTEGER-valued s in [0, 3*255^2]: v_sqrt_f32 (1 ulp) followed by one
// Newton step on the exact fma residual.  Equal to the reference's (float)sqrt((double)s) for every
// such s -- checked exhaustively on the device by tests/test_gpu_gsw.py through ssamd_debug_gsw_sqrt --
// at a third of the instructions of the general correctly-rounded sqrtf expansion.
__device__ __forceinline__ float gsw_sqrt_int(float s)
{
    const float r = __builtin_amdgcn_sqrtf(s);
    const float h = 0.5f * __builtin_amdgcn_rcpf(r);
    const float e = fmaf(-r, r, s);
    const float r1 = fmaf(e, h, r);
    return s > 0.f ? r1 : 0.f;
}

__device__ __forceinline__ float4 bgr_unpack(uint32_t v, float valid)
{
    return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), valid);
}

__device__ __forceinline__ float bgr_dist2f(const float4 a, const float4 b)
{
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z;
    return d0 * d0 + d1 * d1 + d2 * d2;       // integers < 2^24: exact in fp32 whatever the contraction
}

// |a-b|^2 over the three colour bytes, exact in fp32
__device__ __forceinline__ float bgr_dist2(uint32_t a, uint32_t b)
{
    const float d0 = (float)(a & 0xffu) - (float)(b & 0xffu);
    const float d1 = (float)((a >> 8) & 0xffu) - (float)((b >> 8) & 0xffu);
    const float d2 = (float)((a >> 16) & 0xffu) - (float)((b >> 16) & 0xffu);
    return d0 * d0 + d1 * d1 + d2 * d2;
}

__device__ __forceinline__ int gsw_e_offset(int ul, int slot, int Se, int emask)
{
    return ul * Se + ((slot ^ ((ul >> 2) & emask)) << 3);     // in floats; slot = 8 floats
}

// 4 x RD taps of one tap column for one output row: cost = fl(cost + fl(w * e)), no contraction
// (the reference is -O2 x86-64 without FMA)
template <int RD>
__device__ __forceinline__ void gsw_taps(float (&cost)[GSW_RX][RD], const float4 w4, const float (&r0)[RD],
                                         const float (&r1)[RD], const float (&r2)[RD], const float (&r3)[RD])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int di = 0; di < RD; ++di) {
        cost[0][di] = cost[0][di] + w4.x * r0[di];
        cost[1][di] = cost[1][di] + w4.y * r1[di];
        cost[2][di] = cost[2][di] + w4.z * r2[di];
        cost[3][di] = cost[3][di] + w4.w * r3[di];
    }
}

// RD consecutive disparities of e row ul; dg = index of the thread's disparity group (RD = 8: one
// 32-byte slot, RD = 4: half a slot)
template <int RD>
__device__ __forceinline__ void gsw_load_row(float (&row)[RD], const float *e, int ul, int dg, int Se, int emask)
{
    static_assert(RD == 4 || RD == 8, "thread tiles of 4 or 8 disparities");
    if constexpr (RD == 8) {
        const int off = gsw_e_offset(ul, dg, Se, emask);
        const float4 a = *reinterpret_cast<const float4 *>(e + off);
        const float4 b = *reinterpret_cast<const float4 *>(e + off + 4);
        row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w;
        row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(e + gsw_e_offset(ul, dg >> 1, Se, emask) + 4 * (dg & 1));
        row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w;
    }
}

// All tap columns of the current image row for output rows [T0, T1) of the strip.  The e rows slide
// through four register rows (one new row per tap column) and are shared by the output rows.
template <int TY, int RD, int T0, int T1>
__device__ __forceinline__ void gsw_row_taps(float (&cost)[TY][GSW_RX][RD], const float *wS, const float *eT, int win,
                                             int Tx, int xg, int dg, int Se, int emask)
{
    const float *wp = wS + GSW_RX * xg;
    const int ul0 = GSW_RX * xg, wstride = win * Tx;
#define SSAMD_GROW(dst, n) gsw_load_row<RD>(dst, eT, ul0 + (n), dg, Se, emask)
#define SSAMD_GSTEP(j, ra, rb, rc, rd)                                                                        \
    if ((j) < win) {                                                                                          \
        SSAMD_GROW(rd, (j) + 3);                                                                              \
        _Pragma("unroll") for (int t = T0; t < T1; ++t)                                                       \
            gsw_taps<RD>(cost[t], *reinterpret_cast<const float4 *>(wp + t * wstride + (j) * Tx), ra, rb, rc, rd); \
    }
    float e0[RD], e1[RD], e2[RD], e3[RD];
    SSAMD_GROW(e0, 0);
    SSAMD_GROW(e1, 1);
    SSAMD_GROW(e2, 2);
    for (int j0 = 0; j0 < win; j0 += 4) {
        SSAMD_GSTEP(j0, e0, e1, e2, e3)
        SSAMD_GSTEP(j0 + 1, e1, e2, e3, e0)
        SSAMD_GSTEP(j0 + 2, e2, e3, e0, e1)
        SSAMD_GSTEP(j0 + 3, e3, e0, e1, e2)
    }
#undef SSAMD_GSTEP
#undef SSAMD_GROW
}

template <int TY, int RD>
__global__ __launch_bounds__(GSW_MAX_THREADS, 4) void gsw_aggregate_kernel(const GswArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GswGeom &g = A.g;
    float *const wS = reinterpret_cast<float *>(smem + g.off_w);          // [TY][win][Tx]
    float *const eT = reinterpret_cast<float *>(smem + g.off_e);          // [nL][Se]
    float4 *const refS = reinterpret_cast<float4 *>(smem + g.off_ref);    // [nL] {b, g, r, inside image}
    float4 *const tgtS = reinterpret_cast<float4 *>(smem + g.off_tgt);    // [nT]
    u64 *const best = reinterpret_cast<u64 *>(smem + g.off_best);         // [TY][Tx]

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nT = g.nT, Se = g.Se, emask = g.emask;
    const int x0 = blockIdx.x * Tx;
    const int y0 = A.row0 + blockIdx.y * TY;                     // first output row of the strip
    const int ny = min(TY, A.row0 + A.rows - y0);                // output rows of the strip that exist
    const int dlo = A.minD + blockIdx.z * Dc;
    const int dhi = dlo + Dc - 1;
    const bool right = A.right != 0;
    // whole tile without candidates: left pass needs x - d >= 0, right pass x + d <= W-1
    if (!right && min(x0 + Tx - 1, W - 1) - dlo < 0) return;
    if (right && x0 + dlo > W - 1) return;

    const int seg_lo = x0 - p;                                   // first reference tap column
    const int tgt_lo = right ? seg_lo + dlo : seg_lo - dhi;      // first target tap column

    const bool active = tid < g.XG * g.DG;
    const int xg = tid % g.XG, dg = tid / g.XG;

    float cost[TY][GSW_RX][RD];
#pragma unroll
    for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int a = 0; a < GSW_RX; ++a)
#pragma unroll
            for (int b = 0; b < RD; ++b) cost[t][a][b] = 0.f;
    for (int k = tid; k < TY * Tx; k += nthr) best[k] = KEY_NONE;

    // image rows in ascending order: every output row sees its window rows in the reference's raster order
    const int r_lo = max(0, y0 - p), r_hi = min(H - 1, y0 + ny - 1 + p);
    for (int r = r_lo; r <= r_hi; ++r) {
        __syncthreads();                    // previous image row fully consumed
        for (int k = tid; k < nL + nT; k += nthr) {     // staged pixels as floats: converted once per pixel,
            const bool isRef = k < nL;                   // not once per (pixel, disparity) element
            const int idx = isRef ? k : k - nL;
            const int col = (isRef ? seg_lo : tgt_lo) + idx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)col < (unsigned)W) v = bgr_unpack((isRef ? A.ref : A.tgt)[(size_t)r * W + col], 1.f);
            (isRef ? refS : tgtS)[idx] = v;
        }
        __syncthreads();

        // ---- support weights of this image row for the tile's reference pixels, per output row:
        //      image row r is window row i = r - y + pad of output row y
        bool use[TY];
#pragma unroll
        for (int t = 0; t < TY; ++t) {
            const int y = y0 + t, i = r - y + p;
            use[t] = t < ny && (unsigned)i < (unsigned)win;
            if (!use[t]) continue;
            float *const wT = wS + t * win * Tx;
            for (int k = tid; k < Tx * win; k += nthr) {
                const int j = k / Tx, c = k - j * Tx;
                const int x = x0 + c, col = x - p + j;
                float w = 0.f;
                if (x < W && (unsigned)col < (unsigned)W) {
                    const bool centre = (i == p) && (j == p);
                    bool reached = A.iterations > 0;
                    if (!right && x + p >= W) reached = reached && (y == 0) && (i == p);   // left-pass break quirk
                    if (centre) w = 1.0f;                                                   // exp(-0/gamma)
                    else if (reached) {
                        const float4 cpx = bgr_unpack(A.ref[(size_t)y * W + x], 1.f);
                        w = A.tab[(int)bgr_dist2f(refS[c + j], cpx)];
                    }
                }
                wT[k] = w;
            }
        }
        // ---- e[ul][d] = min(fMax, ||ref(r,u) - tgt(r,u -/+ d)||), 0 when the target column is outside.
        //      Task = (column ul, disparity dd) with dd fastest across the lanes of a wave: the e writes of
        //      a wave then fall into consecutive floats (the column-fastest order put 64 lanes on 4 banks),
        //      the reference pixel is a broadcast read and the target pixels are consecutive 16-byte reads.
        //      Task index advanced without divisions; four independent elements per iteration (LDS reads
        //      first, then the sub/fma/sqrt chains).
        {
            const int e_q = nthr / Dc, e_r = nthr - e_q * Dc;
            int ul = tid / Dc, dd = tid - ul * Dc;
            while (ul < nL) {
                int uls[4], dds[4];
                float4 rp[4], tp[4];
                float ev[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uls[u] = min(ul, nL - 1); dds[u] = dd;      // tasks past the end repeat the last column
                    dd += e_r; ul += e_q;
                    if (dd >= Dc) { dd -= Dc; ++ul; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tix = right ? uls[u] + dds[u] : uls[u] + (Dc - 1) - dds[u];
                    rp[u] = refS[uls[u]];
                    tp[u] = tgtS[tix];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) ev[u] = bgr_dist2f(rp[u], tp[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) ev[u] = fminf(A.fMax, gsw_sqrt_int(ev[u]));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    eT[gsw_e_offset(uls[u], dds[u] >> 3, Se, emask) + (dds[u] & 7)] = tp[u].w == 0.f ? 0.f : ev[u];
            }
        }
        __syncthreads();

        if (active) {
            if constexpr (TY == 1) {
                gsw_row_taps<1, RD, 0, 1>(cost, wS, eT, win, Tx, xg, dg, Se, emask);
            } else {
                static_assert(TY == 2, "strips of 1 or 2 output rows");
                if (use[0] && use[1]) gsw_row_taps<2, RD, 0, 2>(cost, wS, eT, win, Tx, xg, dg, Se, emask);
                else if (use[0]) gsw_row_taps<2, RD, 0, 1>(cost, wS, eT, win, Tx, xg, dg, Se, emask);
                else if (use[1]) gsw_row_taps<2, RD, 1, 2>(cost, wS, eT, win, Tx, xg, dg, Se, emask);
            }
        }
    }

    // ---- winner-take-all over this chunk's candidates (strict <, first minimum wins:
    //      _passive.cpp:538-541 / 654-657), then merge across chunks with a global atomic
    if (active) {
#pragma unroll
        for (int t = 0; t < TY; ++t) {
            if (t >= ny) continue;
#pragma unroll
            for (int xi = 0; xi < GSW_RX; ++xi) {
                const int x = x0 + GSW_RX * xg + xi;
                u64 b = KEY_NONE;
#pragma unroll
                for (int di = 0; di < RD; ++di) {
                    const int d = dlo + RD * dg + di;
                    const bool valid = (x < W) && (d <= A.maxD) && (right ? (x + d <= W - 1) : (x - d >= 0));
                    if (valid) b = min(b, make_key(cost[t][xi][di], right ? (uint32_t)(x + d) : (uint32_t)d));
                }
                if (b != KEY_NONE) atomicMin(&best[t * Tx + GSW_RX * xg + xi], b);
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < ny * Tx; k += nthr) {
        const int t = k / Tx, c = k - t * Tx;
        const int x = x0 + c;
        if (x < W && best[k] != KEY_NONE) atomicMin(&A.key[(size_t)(y0 + t - A.row0) * W + x], best[k]);
    }
}

// verification helper: gsw_sqrt_int over s = 0 .. n-1
__global__ __launch_bounds__(256) void gsw_sqrt_probe_kernel(float *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gsw_sqrt_int((float)i);
}

// BGR u8 -> packed dword per pixel (GSW works on raw BGR, _passive.cpp:740-741)
__global__ __launch_bounds__(256) void bgr_pack_kernel(const uint8_t *__restrict__ bgr, uint32_t *__restrict__ out,
                                                       long long npix)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; p < npix; p += stride)
        out[p] = (uint32_t)bgr[3 * p] | ((uint32_t)bgr[3 * p + 1] << 8) | ((uint32_t)bgr[3 * p + 2] << 16);
}

}  // namespace ssamd
