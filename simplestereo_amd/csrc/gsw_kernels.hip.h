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
    int Hy;                         // round 3: thread groups of `threads` lanes each; group h owns output rows Ty*h .. Ty*h + Ty - 1 of the
                                    // workgroup's strip of Ty*Hy rows, all groups share the e tile of an image row
    int nL, nT, Se, emask, Ses;     // Se = 1 << Ses: floats per e row (32-byte slots, XOR-swizzled)
    int off_w, off_e, off_ref, off_tgt, off_best, off_cen;
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

// fl32(sqrt(s)) for an INTEGER-valued s in [0, 3*255^2]: r = s * v_rsq_f32(s) (about 1.5 ulp) followed
// by one Newton step on the exact fma residual.  Equal to the reference's (float)sqrt((double)s) for every
// such s -- checked exhaustively on the device by tests/test_gpu_gsw.py through ssamd_debug_gsw_sqrt --
// at a quarter of the instructions of the general correctly-rounded sqrtf expansion.  The clamp keeps
// s = 0 finite: r = 0 * rsq(tiny) = 0, residual 0, result 0.
__device__ __forceinline__ float gsw_sqrt_int(float s)
{
    const float y = __builtin_amdgcn_rsqf(fmaxf(s, 1e-30f));
    const float r = s * y, h = 0.5f * y;
    const float e = fmaf(-r, r, s);
    return fmaf(e, h, r);
}

// A staged pixel: packed colour bytes (B | G<<8 | R<<16), their squared norm, 1.0f / 0.0f for inside / outside the
// image, and the cap of the colour distance towards this pixel: fMax inside the image, 0 outside -- so that
// e = min(cap, distance) is the reference's truncated distance for an in-image target and the +0 of a skipped tap
// otherwise (_passive.cpp:511-512, 522-530), without a multiplication by the inside flag.
struct alignas(16) GswPix {
    uint32_t bgr, norm;
    float cap, inside;
};

__device__ __forceinline__ GswPix gsw_pix(uint32_t v, float inside, float cap)
{
    return GswPix{v, __builtin_amdgcn_udot4(v, v, 0u, false), cap, inside};
}

// |a - b|^2 over the three colour bytes = |a|^2 + |b|^2 - 2 a.b, in integers (one v_dot4_u32_u8)
__device__ __forceinline__ uint32_t gsw_dist2(const GswPix a, const GswPix b)
{
    return a.norm + b.norm - 2u * __builtin_amdgcn_udot4(a.bgr, b.bgr, 0u, false);
}

__device__ __forceinline__ int gsw_e_offset(int ul, int slot, int Ses, int emask)
{
    return (ul << Ses) + ((slot ^ ((ul >> 2) & emask)) << 3);     // in floats; slot = 8 floats
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

// RD consecutive disparities of an e row, from its (swizzled) address
template <int RD>
__device__ __forceinline__ void gsw_load_row(float (&row)[RD], const float *e)
{
    static_assert(RD == 4 || RD == 8, "thread tiles of 4 or 8 disparities");
    const float4 a = *reinterpret_cast<const float4 *>(e);
    row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w;
    if constexpr (RD == 8) {
        const float4 b = *reinterpret_cast<const float4 *>(e + 4);
        row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
    }
}

// All tap columns of the current image row for output rows [T0, T1) of the strip.  The e rows slide
// through four register rows (one new row per tap column) and are shared by the output rows.
// The thread's columns start at ul0 = 4 xg, so e row ul0 + n has swizzle key xg + n/4: the key
// changes once per four tap columns and the address of a row is pointer + n * pitch + key term.
template <int TY, int RD, int T0, int T1>
__device__ __forceinline__ void gsw_row_taps(float (&cost)[TY][GSW_RX][RD], const float *wS, const float *eT, int win,
                                             int Tx, int xg, int dg, int Ses, int emask)
{
    const float *wp = wS + GSW_RX * xg;
    const int wstride = win * Tx, pitch = 1 << Ses;
    const int slot = RD == 8 ? dg : dg >> 1, half = RD == 8 ? 0 : 4 * (dg & 1);    // dg-th group of RD disparities
    auto key = [&](int k) { return ((slot ^ ((xg + k) & emask)) << 3) + half; };
    const float *er = eT + ((GSW_RX * xg) << Ses);       // row ul0 + j0
    int kA = key(0), kB = key(1);
#define SSAMD_GSTEP(j, ra, rb, rc, rd, K)                                                                     \
    if ((j) < win) {                                                                                          \
        gsw_load_row<RD>(rd, er + (3 + (j) - j0) * pitch + K);                                                \
        _Pragma("unroll") for (int t = T0; t < T1; ++t)                                                       \
            gsw_taps<RD>(cost[t], *reinterpret_cast<const float4 *>(wp + t * wstride + (j) * Tx), ra, rb, rc, rd); \
    }
    float e0[RD], e1[RD], e2[RD], e3[RD];
    gsw_load_row<RD>(e0, er + kA);
    gsw_load_row<RD>(e1, er + pitch + kA);
    gsw_load_row<RD>(e2, er + 2 * pitch + kA);
    for (int j0 = 0; j0 < win; j0 += 4) {
        SSAMD_GSTEP(j0, e0, e1, e2, e3, kA)              // rows ul0 + j0 + 3 | + 4 .. + 6: keys j0/4 | j0/4 + 1
        SSAMD_GSTEP(j0 + 1, e1, e2, e3, e0, kB)
        SSAMD_GSTEP(j0 + 2, e2, e3, e0, e1, kB)
        SSAMD_GSTEP(j0 + 3, e3, e0, e1, e2, kB)
        er += 4 * pitch;
        kA = kB;
        kB = key(j0 / 4 + 2);
    }
#undef SSAMD_GSTEP
}

template <int TY, int RD>
__global__ __launch_bounds__(GSW_MAX_THREADS, 4) void gsw_aggregate_kernel(const GswArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GswGeom &g = A.g;
    float *const wS = reinterpret_cast<float *>(smem + g.off_w);          // [TY][win][Tx]
    float *const eT = reinterpret_cast<float *>(smem + g.off_e);          // [nL][Se]
    GswPix *const refS0 = reinterpret_cast<GswPix *>(smem + g.off_ref);   // [2][nL4]: image rows alternate
    GswPix *const tgtS0 = reinterpret_cast<GswPix *>(smem + g.off_tgt);   // [2][nT4]
    u64 *const best = reinterpret_cast<u64 *>(smem + g.off_best);         // [TY * Hy][Tx]
    uint32_t *const cenS = reinterpret_cast<uint32_t *>(smem + g.off_cen); // [TY * Hy][Tx] centre pixels of the strip's outputs

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    const int Tx = g.Tx, Dc = g.Dc, nL = g.nL, nT = g.nT, Ses = g.Ses, emask = g.emask;
    const int nL4 = (nL + 3) & ~3, nT4 = nT + (nL4 - nL);      // staged columns, padded for the 4-column e tasks
    const int x0 = blockIdx.x * Tx;
    // Thread groups (round 3): the workgroup's strip has TY * Hy output rows; group h = tid / g.threads (whole waves)
    // aggregates rows TY h .. TY h + TY - 1.  Staging, the support weights of all rows and the e tile -- the expensive
    // part: a correctly rounded square root per element -- are built by ALL threads once per image row and shared, so
    // an e tile serves win window rows of TY * Hy outputs instead of TY: (win + 3) / 4 = 3.5 rebuilds per output row
    // with four-row strips against 6 with two-row ones.
    const int TYS = TY * g.Hy;
    const int grp = __builtin_amdgcn_readfirstlane(tid / g.threads), gtid = tid - grp * g.threads;
    const int y0 = A.row0 + blockIdx.y * TYS;                    // first output row of the strip
    const int ny = min(TYS, A.row0 + A.rows - y0);               // output rows of the strip that exist
    const int yg = y0 + TY * grp;                                // first output row of this thread group
    const int dlo = A.minD + blockIdx.z * Dc;
    const int dhi = dlo + Dc - 1;
    const bool right = A.right != 0;
    // whole tile without candidates: left pass needs x - d >= 0, right pass x + d <= W-1
    if (!right && min(x0 + Tx - 1, W - 1) - dlo < 0) return;
    if (right && x0 + dlo > W - 1) return;

    const int seg_lo = x0 - p;                                   // first reference tap column
    const int tgt_lo = right ? seg_lo + dlo : seg_lo - dhi;      // first target tap column

    const bool active = gtid < g.XG * g.DG && TY * grp < ny;
    const int xg = gtid % g.XG, dg = gtid / g.XG;

    float cost[TY][GSW_RX][RD];
#pragma unroll
    for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int a = 0; a < GSW_RX; ++a)
#pragma unroll
            for (int b = 0; b < RD; ++b) cost[t][a][b] = 0.f;
    for (int k = tid; k < TYS * Tx; k += nthr) {
        best[k] = KEY_NONE;
        const int t = k / Tx, c = k - t * Tx;
        cenS[k] = (t < ny && x0 + c < W) ? A.ref[(size_t)(y0 + t) * W + x0 + c] : 0u;      // (published by the first barrier of the row loop)
    }

    // image rows in ascending order: every output row sees its window rows in the reference's raster order
    const int r_lo = max(0, y0 - p), r_hi = min(H - 1, y0 + ny - 1 + p);
    // pixels of image row rr into staging buffer `buf` (norm computed once per pixel, not once per (pixel,
    // disparity) element).  Rows are staged one ahead: the global latency sits under the previous row's taps and
    // the barrier that ends those taps also publishes the pixels.
    auto stage_row = [&](int rr, int buf) {
        GswPix *const rS = refS0 + buf * nL4, *const tS = tgtS0 + buf * nT4;
        for (int k = tid; k < nL4 + nT4; k += nthr) {
            const bool isRef = k < nL4;
            const int idx = isRef ? k : k - nL4;
            const int col = (isRef ? seg_lo : tgt_lo) + idx;
            GswPix v = gsw_pix(0u, 0.f, 0.f);
            if ((unsigned)col < (unsigned)W) v = gsw_pix((isRef ? A.ref : A.tgt)[(size_t)rr * W + col], 1.f, A.fMax);
            (isRef ? rS : tS)[idx] = v;
        }
    };
    stage_row(r_lo, r_lo & 1);
    for (int r = r_lo; r <= r_hi; ++r) {
        __syncthreads();                    // previous image row fully consumed; this row's pixels staged
        const GswPix *const refS = refS0 + (r & 1) * nL4, *const tgtS = tgtS0 + (r & 1) * nT4;

        // ---- support weights of this image row for the tile's reference pixels, per output row:
        //      image row r is window row i = r - y + pad of output row y.  A thread keeps one reference
        //      column c (its centre pixel is fetched once) and walks the tap columns j.
        bool use[TY];                       // this thread group's rows
#pragma unroll
        for (int t = 0; t < TY; ++t) use[t] = TY * grp + t < ny && (unsigned)(r - (yg + t) + p) < (unsigned)win;
        // (SSAMD_GABLATE_*: phase-ablation builds of tools/build_variants.sh -- wrong maps by construction, never defined in the product)
#ifdef SSAMD_GABLATE_W
        if (r == r_lo)
#endif
        {
            // a thread keeps one reference column c and walks (output row of the strip, tap column) with its centre pixel
            // fetched once per row.  (A flat deal of (row, tap column) pairs over the threads -- for tall strips of
            // narrow tiles, where a thread per tap column leaves threads idle -- was built and measured in round 3:
            // +2 % on two-row strips from its per-task index arithmetic; dropped together with tall strips as the default.)
            const int lanesX = min(Tx, nthr), qw = nthr / lanesX;
            const int c0 = tid % lanesX, jq = tid / lanesX;
            if (jq < qw) {
                for (int c = c0; c < Tx; c += lanesX) {
                    const int x = x0 + c;
                    for (int t = 0; t < TYS; ++t) {          // the weights of every output row of the strip
                        if (!(t < ny && (unsigned)(r - (y0 + t) + p) < (unsigned)win)) continue;
                        const int y = y0 + t, i = r - y + p;
                        float *const wT = wS + t * win * Tx + c;
                        bool reached = A.iterations > 0 && x < W;
                        if (!right && x + p >= W) reached = reached && (y == 0) && (i == p);   // left-pass break quirk
                        const GswPix cpx = gsw_pix(cenS[t * Tx + c], 1.f, 0.f);
                        // two tap columns per batch, branch-free: their table gathers are in flight together
                        for (int jb = jq; jb < win; jb += 2 * qw) {
                            float w[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int j = min(jb + u * qw, win - 1);
                                const GswPix px = refS[c + j];      // .inside: tap column x - pad + j is in the image
                                const bool centre = i == p && j == p;
                                const bool gather = reached && !centre && px.inside != 0.f;
                                const float tw = A.tab[gather ? gsw_dist2(px, cpx) : 0u];
                                w[u] = gather ? tw : (centre && x < W ? 1.0f : 0.f);      // centre: exp(-0/gamma)
                            }
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                if (jb + u * qw < win) wT[(jb + u * qw) * Tx] = w[u];
                        }
                    }
                }
            }
        }
        // ---- e[ul][d] = min(fMax, ||ref(r,u) - tgt(r,u -/+ d)||), 0 when the target column is outside.
        //      A task is four adjacent columns ul = 4m .. 4m+3 at one disparity dd: they share the swizzle
        //      key, so one address computation serves four independent sub/dot/sqrt chains.  Each thread
        //      keeps its dd and walks m with a constant stride; dd is fastest across the lanes of a wave:
        //      the e writes of a wave fall into consecutive floats, the reference pixels are broadcast
        //      reads and the target pixels consecutive 16-byte reads.  (nL is padded to a multiple of 4;
        //      the padding columns hold outside-the-image pixels and are never read by the taps.)
#ifdef SSAMD_GABLATE_E
        if (r == r_lo)
#endif
        {
            const int lanesD = min(Dc, nthr), q = nthr / lanesD;
            const int dd0 = tid % lanesD, mq = tid / lanesD;
            if (mq < q) {
                for (int dd = dd0; dd < Dc; dd += lanesD) {
                    const int tofs = right ? dd : (Dc - 1) - dd;
                    const int eofs = dd & 7, slot = dd >> 3;
                    // running pointers: per task one add each for the pixels and the e rows, and the swizzled slot
                    // (gsw_e_offset: (ul << Ses) + ((slot ^ ((ul >> 2) & emask)) << 3) with ul = 4 m)
                    const int pitch = 1 << Ses, rstep = 4 * q;
                    const GswPix *rp = refS + 4 * mq, *tp = tgtS + 4 * mq + tofs;
                    float *erow = eT + ((4 * mq) << Ses) + eofs;
                    for (int m = mq; 4 * m < nL; m += q, rp += rstep, tp += rstep, erow += rstep << Ses) {
                        int swz = (slot ^ (m & emask)) << 3;
                        asm volatile("" : "+v"(swz));          // one base, the three other rows at + k * pitch (not four running pointers)
                        float *const ep = erow + swz;
                        GswPix rv[4], tv[4];
                        float ev[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { rv[u] = rp[u]; tv[u] = tp[u]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(tv[u].inside));      // keeps the target reads ds_read_b128 (the 12-byte form is twice as slow)
#pragma unroll
                        for (int u = 0; u < 4; ++u) ev[u] = (float)gsw_dist2(rv[u], tv[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float d = gsw_sqrt_int(ev[u]);
                            // v_min_f32 as written: fminf() would first canonicalise the cap it has just read from LDS (one more instruction per element)
                            asm("v_min_f32 %0, %1, %2" : "=v"(ev[u]) : "v"(tv[u].cap), "v"(d));
                        }
                        ep[0] = ev[0]; ep[pitch] = ev[1]; ep[2 * pitch] = ev[2]; ep[3 * pitch] = ev[3];
                    }
                }
            }
        }
        __syncthreads();
        if (r < r_hi) stage_row(r + 1, (r + 1) & 1);       // prefetch: its global latency sits under the taps below

#ifdef SSAMD_GABLATE_TAPS
        if (r == r_lo)
#endif
        if (active) {
            const float *const wG = wS + (TY * grp) * win * Tx;       // this group's weight rows
            if constexpr (TY == 1) {
                if (use[0]) gsw_row_taps<1, RD, 0, 1>(cost, wG, eT, win, Tx, xg, dg, Ses, emask);
            } else {
                static_assert(TY == 2, "thread tiles of 1 or 2 output rows");
                if (use[0] && use[1]) gsw_row_taps<2, RD, 0, 2>(cost, wG, eT, win, Tx, xg, dg, Ses, emask);
                else if (use[0]) gsw_row_taps<2, RD, 0, 1>(cost, wG, eT, win, Tx, xg, dg, Ses, emask);
                else if (use[1]) gsw_row_taps<2, RD, 1, 2>(cost, wG, eT, win, Tx, xg, dg, Ses, emask);
            }
        }
    }

    // ---- winner-take-all over this chunk's candidates (strict <, first minimum wins:
    //      _passive.cpp:538-541 / 654-657), then merge across chunks with a global atomic
    if (active) {
#pragma unroll
        for (int t = 0; t < TY; ++t) {
            if (TY * grp + t >= ny) continue;
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
                if (b != KEY_NONE) atomicMin(&best[(TY * grp + t) * Tx + GSW_RX * xg + xi], b);
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

// verification helper: gsw_sqrt_int over s = 0 .. n-1 (what = 1: the bare v_sqrt_f32, for the record of why it is not used)
__global__ __launch_bounds__(256) void gsw_sqrt_probe_kernel(float *__restrict__ out, int n, int what)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = what == 1 ? __builtin_amdgcn_sqrtf((float)i) : gsw_sqrt_int((float)i);
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
