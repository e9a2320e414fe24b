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
// Work decomposition mirrors the ASW kernel: workgroup = (row y, tile of Tx reference
// columns, chunk of Dc disparities); thread = 4 columns x 8 disparities; per window row the
// group builds w[j][x] and e[u][d] (fp32) in LDS; e rows slide through registers.
#pragma once
#include "common.hip.h"

namespace ssamd {

static constexpr int GSW_RX = 4;
static constexpr int GSW_RD = 8;
static constexpr int GSW_MAX_THREADS = 512;
static constexpr int GSW_TAB_SIZE = 3 * 255 * 255 + 1;

struct GswGeom {
    int Tx, XG, DG, Dc, nchunks, threads;
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

// 32 taps of one tap column: cost = fl(cost + fl(w * e)), no contraction (reference is -O2 x86-64)
__device__ __forceinline__ void gsw_taps(float (&cost)[GSW_RX][GSW_RD], const float4 w4, const float (&r0)[GSW_RD],
                                         const float (&r1)[GSW_RD], const float (&r2)[GSW_RD], const float (&r3)[GSW_RD])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int di = 0; di < GSW_RD; ++di) {
        cost[0][di] = cost[0][di] + w4.x * r0[di];
        cost[1][di] = cost[1][di] + w4.y * r1[di];
        cost[2][di] = cost[2][di] + w4.z * r2[di];
        cost[3][di] = cost[3][di] + w4.w * r3[di];
    }
}

__device__ __forceinline__ void gsw_load_row(float (&row)[GSW_RD], const float *e, int off)
{
    const float4 a = *reinterpret_cast<const float4 *>(e + off);
    const float4 b = *reinterpret_cast<const float4 *>(e + off + 4);
    row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w;
    row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
}

__global__ __launch_bounds__(GSW_MAX_THREADS) void gsw_aggregate_kernel(const GswArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GswGeom &g = A.g;
    float *const wS = reinterpret_cast<float *>(smem + g.off_w);          // [win][Tx]
    float *const eT = reinterpret_cast<float *>(smem + g.off_e);          // [nL][Se]
    float4 *const refS = reinterpret_cast<float4 *>(smem + g.off_ref);    // [nL] {b, g, r, inside image}
    float4 *const tgtS = reinterpret_cast<float4 *>(smem + g.off_tgt);    // [nT]
    u64 *const best = reinterpret_cast<u64 *>(smem + g.off_best);         // [Tx]

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nT = g.nT, Se = g.Se, emask = g.emask;
    const int x0 = blockIdx.x * Tx;
    const int y = A.row0 + blockIdx.y;
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

    float cost[GSW_RX][GSW_RD];
#pragma unroll
    for (int a = 0; a < GSW_RX; ++a)
#pragma unroll
        for (int b = 0; b < GSW_RD; ++b) cost[a][b] = 0.f;
    for (int k = tid; k < Tx; k += nthr) best[k] = KEY_NONE;

    const int i_lo = max(0, p - y), i_hi = min(win, H + p - y);
    for (int i = i_lo; i < i_hi; ++i) {
        const int r = y - p + i;
        __syncthreads();                    // previous window row fully consumed
        for (int k = tid; k < nL + nT; k += nthr) {     // staged pixels as floats: converted once per pixel,
            const bool isRef = k < nL;                   // not once per (pixel, disparity) element
            const int idx = isRef ? k : k - nL;
            const int col = (isRef ? seg_lo : tgt_lo) + idx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)col < (unsigned)W) v = bgr_unpack((isRef ? A.ref : A.tgt)[(size_t)r * W + col], 1.f);
            (isRef ? refS : tgtS)[idx] = v;
        }
        __syncthreads();

        // ---- support weights of window row i for the tile's reference pixels
        for (int t = tid; t < Tx * win; t += nthr) {
            const int j = t / Tx, c = t - j * Tx;
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
            wS[j * Tx + c] = w;
        }
        // ---- e[ul][d] = min(fMax, ||ref(r,u) - tgt(r,u -/+ d)||), 0 when the target column is outside.
        //      Task index (column ul fastest, then disparity) advanced without divisions.
        //      Task = (column ul, disparity dd) with dd fastest across the lanes of a wave: the e writes of
        //      a wave then fall into consecutive floats (the column-fastest order put 64 lanes on 4 banks),
        //      the reference pixel is a broadcast read and the target pixels are consecutive 16-byte reads.
        //      Four independent elements per iteration (LDS reads first, then the sub/fma/sqrt chains).
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
            const float *wp = wS + GSW_RX * xg;
            const int ul0 = GSW_RX * xg;
#define SSAMD_GROW(dst, n) gsw_load_row(dst, eT, gsw_e_offset(ul0 + (n), dg, Se, emask))
#define SSAMD_GSTEP(j, ra, rb, rc, rd)                                                       \
    if ((j) < win) {                                                                         \
        SSAMD_GROW(rd, (j) + 3);                                                             \
        gsw_taps(cost, *reinterpret_cast<const float4 *>(wp + (j) * Tx), ra, rb, rc, rd);    \
    }
            float e0[GSW_RD], e1[GSW_RD], e2[GSW_RD], e3[GSW_RD];
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
    }

    // ---- winner-take-all over this chunk's candidates (strict <, first minimum wins:
    //      _passive.cpp:538-541 / 654-657), then merge across chunks with a global atomic
    if (active) {
#pragma unroll
        for (int xi = 0; xi < GSW_RX; ++xi) {
            const int x = x0 + GSW_RX * xg + xi;
            u64 b = KEY_NONE;
#pragma unroll
            for (int di = 0; di < GSW_RD; ++di) {
                const int d = dlo + GSW_RD * dg + di;
                const bool valid = (x < W) && (d <= A.maxD) && (right ? (x + d <= W - 1) : (x - d >= 0));
                if (valid) b = min(b, make_key(cost[xi][di], right ? (uint32_t)(x + d) : (uint32_t)d));
            }
            if (b != KEY_NONE) atomicMin(&best[GSW_RX * xg + xi], b);
        }
    }
    __syncthreads();
    const size_t orow = (size_t)(y - A.row0) * W;
    for (int k = tid; k < Tx; k += nthr) {
        const int x = x0 + k;
        if (x < W && best[k] != KEY_NONE) atomicMin(&A.key[orow + x], best[k]);
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
