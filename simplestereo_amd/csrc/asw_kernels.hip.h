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
//   thread     = register tile of RX=4 columns x RD=8 disparities, 2 fp32
//                accumulators (cost, weight sum) per (x,d) pair
//   outer loop = the window rows i (tap row r = y - pad + i).  Per window row the
//                workgroup stages the pixels it needs in LDS and builds, in LDS,
//                  wL[j][x]   left support weights  (x in tile, tap column j)
//                  wR[j][xr]  right support weights (xr = x - d over the tile: they
//                             do not depend on x, the reference re-evaluates them
//                             for every (x,d), _passive.cpp:71-74)
//                  e[u][d]    truncated absolute difference of L[r][u], R[r][u-d]
//                             as bytes: it depends on the tap column u = x+j-pad
//                             only, so each value serves up to `win` taps
//   inner loop = tap columns j; per step a thread reads 4 wL, 12 wR (b128 reads)
//                and ONE new 8-byte row of e (the other three slide in registers),
//                then does 32 taps x {cvt_ubyte, mul, fma, add}.
// HBM traffic is the pixel records only (16 B/pixel/image, re-read from L2 by
// neighbouring tiles) plus 8-byte WTA keys; everything else lives in LDS/VGPRs.
#pragma once
#include "common.hip.h"

namespace ssamd {

static constexpr int ASW_RX = 4;      // columns per thread
static constexpr int ASW_RD = 8;      // disparities per thread
static constexpr int ASW_MAX_THREADS = 512;

struct AswGeom {
    int Tx, XG, DG, Dc, nchunks, threads;
    int nL, nR, nRc, SR, Se;
    int off_wL, off_wR, off_e, off_labL, off_labR, off_bgrL, off_bgrR, off_bestL, off_bestR;
    int lds_bytes;
};

struct AswArgs {
    const PixRec *recL, *recR;   // [H][W] pixel records of the (sub-)image
    const float *prox;           // [win*win] proximity weights exp(-|t|/gammaP)
    u64 *keyL;                   // [rows][W] left-referenced WTA keys  (cost, d)
    u64 *keyR;                   // [rows][W] right-referenced WTA keys (cost, xl) or nullptr
    float *costs;                // optional [rows][W][nD] raw cost dump
    int H, W, win, pad, minD, maxD, row0, rows;
    float kC;                    // -log2(e)/gammaC
    AswGeom g;
};

// 32 taps of one tap column for the thread's 4x8 register tile.
// r0..r3: the e rows (8 packed bytes = 8 disparities) for the thread's 4 columns.
__device__ __forceinline__ void asw_taps(float (&cost)[ASW_RX][ASW_RD], float (&tot)[ASW_RX][ASW_RD],
                                         const float4 wl4, const float4 wa, const float4 wb, const float4 wc,
                                         const uint2 r0, const uint2 r1, const uint2 r2, const uint2 r3)
{
    const float wl[4] = {wl4.x, wl4.y, wl4.z, wl4.w};
    const float wr[12] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y, wc.z, wc.w};
    const uint2 rows[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int xi = 0; xi < ASW_RX; ++xi) {
#pragma unroll
        for (int di = 0; di < ASW_RD; ++di) {
            const uint32_t word = di < 4 ? rows[xi].x : rows[xi].y;
            const float e = (float)((word >> (8 * (di & 3))) & 0xffu);   // v_cvt_f32_ubyteN
            const float w = wl[xi] * wr[xi - di + 7];
            cost[xi][di] = fmaf(w, e, cost[xi][di]);
            tot[xi][di] += w;
        }
    }
}

template <bool WITH_COSTS>
__global__ __launch_bounds__(ASW_MAX_THREADS, 4) void asw_aggregate_kernel(const AswArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AswGeom &g = A.g;
    float *const wL = reinterpret_cast<float *>(smem + g.off_wL);
    float *const wR = reinterpret_cast<float *>(smem + g.off_wR);
    unsigned char *const eT = reinterpret_cast<unsigned char *>(smem + g.off_e);
    float4 *const labL = reinterpret_cast<float4 *>(smem + g.off_labL);
    float4 *const labR = reinterpret_cast<float4 *>(smem + g.off_labR);
    uint32_t *const bgrL = reinterpret_cast<uint32_t *>(smem + g.off_bgrL);
    uint32_t *const bgrR = reinterpret_cast<uint32_t *>(smem + g.off_bgrR);
    u64 *const bestL = reinterpret_cast<u64 *>(smem + g.off_bestL);
    u64 *const bestR = reinterpret_cast<u64 *>(smem + g.off_bestR);

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = A.W, win = A.win, p = A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nR = g.nR, nRc = g.nRc, SR = g.SR, Se = g.Se;
    const int x0 = blockIdx.x * Tx;
    const int y = A.row0 + blockIdx.y;
    const int dlo = A.minD + blockIdx.z * Dc;
    const int dhi = dlo + Dc - 1;
    // no (x,d) pair of this tile has x-d >= 0 (left image border): nothing to do
    if (min(x0 + Tx - 1, W - 1) - dlo < 0) return;

    const int segL_lo = x0 - p;        // first tap column staged from the left image
    const int xrc_lo = x0 - dhi;       // first right-image window centre of the tile
    const int segR_lo = xrc_lo - p;    // first tap column staged from the right image

    const bool active = tid < g.XG * g.DG;
    const int dg = tid % g.DG, xg = tid / g.DG;

    float cost[ASW_RX][ASW_RD], tot[ASW_RX][ASW_RD];
#pragma unroll
    for (int a = 0; a < ASW_RX; ++a)
#pragma unroll
        for (int b = 0; b < ASW_RD; ++b) { cost[a][b] = 0.f; tot[a][b] = 0.f; }

    for (int k = tid; k < Tx; k += nthr) bestL[k] = KEY_NONE;
    for (int k = tid; k <= nRc; k += nthr) bestR[k] = KEY_NONE;

    // e-build task walk: task t -> (ul = t % nL, dq = t / nL), advanced incrementally
    const int e_q = nthr / nL, e_r = nthr % nL;
    const int e_ul0 = tid % nL, e_dq0 = tid / nL;
    const int nDq = Dc >> 2;

    const int i_lo = max(0, p - y), i_hi = min(win, A.H + p - y);
    for (int i = i_lo; i < i_hi; ++i) {
        const int r = y - p + i;
        const PixRec *const rowL = A.recL + (size_t)r * W;
        const PixRec *const rowR = A.recR + (size_t)r * W;

        // ---- stage the pixels of image row r this tile touches (coalesced 16 B loads)
        for (int k = tid; k < nL + nR; k += nthr) {
            const bool isL = k < nL;
            const int idx = isL ? k : k - nL;
            const int col = (isL ? segL_lo : segR_lo) + idx;
            PixRec v;
            v.L = v.a = v.b = 0.f;
            v.bgrx = 0u;
            if ((unsigned)col < (unsigned)W) v = (isL ? rowL : rowR)[col];
            (isL ? labL : labR)[idx] = make_float4(v.L, v.a, v.b, 0.f);
            (isL ? bgrL : bgrR)[idx] = v.bgrx;
        }
        __syncthreads();   // staged pixels visible; every thread is done with main(i-1)

        // ---- support weights of window row i: one window centre per thread, 'win' taps
        //      (_passive.cpp:47-50 and 71-74; exp(-dist/gammaC) = exp2(dist*kC))
        for (int c = tid; c < Tx + nRc; c += nthr) {
            const bool isL = c < Tx;
            const int cc = isL ? c : c - Tx;
            const int ccol = (isL ? x0 : xrc_lo) + cc;
            const bool cvalid = (unsigned)ccol < (unsigned)W;
            float cL = 0.f, ca = 0.f, cb = 0.f;
            if (cvalid) {
                const PixRec v = (isL ? A.recL : A.recR)[(size_t)y * W + ccol];
                cL = v.L; ca = v.a; cb = v.b;
            }
            const float4 *const seg = (isL ? labL : labR) + cc;
            float *const wout = (isL ? wL : wR) + cc;
            const int stride = isL ? Tx : SR;
            const int col0 = ccol - p;
            const float *const prow = A.prox + i * win;
            for (int j = 0; j < win; ++j) {
                const float4 t = seg[j];
                const float dL = t.x - cL, da = t.y - ca, db = t.z - cb;
                const float dist = __builtin_amdgcn_sqrtf(fmaf(db, db, fmaf(da, da, dL * dL)));
                float w = prow[j] * __builtin_amdgcn_exp2f(dist * A.kC);
                if (!cvalid || (unsigned)(col0 + j) >= (unsigned)W) w = 0.f;   // tap outside the image
                wout[j * stride] = w;
            }
        }

        // ---- truncated absolute differences e[ul][d] = min(40, |dB|+|dG|+|dR|), 4 bytes/task
        //      (_passive.cpp:77-79); pixel bytes are B,G,R,0 so v_sad_u8 sums the 3 channels
        {
            int ul = e_ul0, dq = e_dq0;
            while (dq < nDq) {
                const uint32_t lp = bgrL[ul];
                const int rbase = ul + (Dc - 1) - 4 * dq;     // index of R[u-d] for d = dlo+4dq
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t s = min(__builtin_amdgcn_sad_u8(lp, bgrR[rbase - k], 0u), 40u);
                    packed |= s << (8 * k);
                }
                *reinterpret_cast<uint32_t *>(eT + ul * Se + 4 * dq) = packed;
                ul += e_r; dq += e_q;
                if (ul >= nL) { ul -= nL; ++dq; }
            }
        }
        __syncthreads();   // wL, wR, e ready

        // ---- aggregation over the tap columns j of this window row
        if (active) {
            const float *wLp = wL + ASW_RX * xg;
            const float *wRp = wR + (ASW_RX * xg - ASW_RD * dg + Dc - ASW_RD);
            const unsigned char *ep = eT + (ASW_RX * xg) * Se + ASW_RD * dg;
#define SSAMD_E(n) (*reinterpret_cast<const uint2 *>(ep + (n) * Se))
#define SSAMD_W4(ptr, off) (*reinterpret_cast<const float4 *>((ptr) + (off)))
#define SSAMD_STEP(j, ra, rb, rc, rd)                                                           \
    if ((j) < win) {                                                                            \
        rd = SSAMD_E((j) + 3);                                                                  \
        asw_taps(cost, tot, SSAMD_W4(wLp, (j) * Tx), SSAMD_W4(wRp, (j) * SR),                   \
                 SSAMD_W4(wRp, (j) * SR + 4), SSAMD_W4(wRp, (j) * SR + 8), ra, rb, rc, rd);     \
    }
            uint2 e0 = SSAMD_E(0), e1 = SSAMD_E(1), e2 = SSAMD_E(2), e3;
            for (int j0 = 0; j0 < win; j0 += 4) {
                SSAMD_STEP(j0, e0, e1, e2, e3)
                SSAMD_STEP(j0 + 1, e1, e2, e3, e0)
                SSAMD_STEP(j0 + 2, e2, e3, e0, e1)
                SSAMD_STEP(j0 + 3, e3, e0, e1, e2)
            }
#undef SSAMD_STEP
#undef SSAMD_W4
#undef SSAMD_E
        }
    }

    // ---- weighted average (_passive.cpp:88) and the two WTA reductions
    if (active) {
        u64 diag[ASW_RX + ASW_RD - 1];
#pragma unroll
        for (int k = 0; k < ASW_RX + ASW_RD - 1; ++k) diag[k] = KEY_NONE;
#pragma unroll
        for (int xi = 0; xi < ASW_RX; ++xi) {
            const int x = x0 + ASW_RX * xg + xi;
            u64 bl = KEY_NONE;
#pragma unroll
            for (int di = 0; di < ASW_RD; ++di) {
                const int d = dlo + ASW_RD * dg + di;
                const bool valid = (x < W) && (d <= A.maxD) && (x - d >= 0);
                if (valid) {
                    const float c = cost[xi][di] / tot[xi][di];
                    bl = min(bl, make_key(c, (uint32_t)d));
                    diag[xi - di + 7] = min(diag[xi - di + 7], make_key(c, (uint32_t)x));
                    if (WITH_COSTS)
                        A.costs[((size_t)(y - A.row0) * W + x) * (A.maxD - A.minD + 1) + (d - A.minD)] = c;
                }
            }
            if (bl != KEY_NONE) atomicMin(&bestL[ASW_RX * xg + xi], bl);
        }
        if (A.keyR) {
            const int base = ASW_RX * xg - ASW_RD * dg + Dc - ASW_RD;
#pragma unroll
            for (int k = 0; k < ASW_RX + ASW_RD - 1; ++k)
                if (diag[k] != KEY_NONE) atomicMin(&bestR[base + k], diag[k]);
        }
    }
    __syncthreads();
    const size_t orow = (size_t)(y - A.row0) * W;
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

// K2a: decode left keys (non-consistent mode).  disparity = d of the best key, or x
// when the candidate loop was empty (dBest stays 0, _passive.cpp:54,98).
__global__ __launch_bounds__(256) void wta_decode_kernel(const u64 *__restrict__ keyL, int16_t *__restrict__ disp,
                                                         int rows, int W)
{
    const long long n = (long long)rows * W;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) {
        const u64 k = keyL[idx];
        const int x = (int)(idx % W);
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

}  // namespace ssamd
