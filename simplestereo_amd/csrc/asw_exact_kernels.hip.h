// K1x: fp64 tie-break pass of the ASW matchers (opt-in `exact` mode; gfx950).
//
// The reference aggregates in double (_passive.cpp:23, 56-95); the aggregation kernels accumulate (N, S') in fp32 and
// resolve costs to ~1e-6 relative.  Where two candidates of a pixel are closer than that, the fp32 argmin may pick the
// other one (0.0004 % of the pixels of the bench frame, 0.09 % on the class-default lawn photograph).  This pass makes the
// argmin the reference's wherever fp64 can tell the candidates apart:
//
//   1. the aggregation kernel also dumps the 32-bit cost image of every candidate it evaluates (the `costs` dump the
//      verification entry points already have, holding asw_cost_key() images instead of floats: H*W*nD*4 bytes of HBM,
//      1.6 GB at 1080p / 193 -- 0.4 ms to write at HBM speed next to a 36 ms aggregation; 288 GB make that affordable);
//   2. asw_exact_flag_kernel reads the volume once, coalesced: a candidate whose key image is within `tol` ulps of its
//      pixel's winning key (left-referenced: keyL[y][x]; right-referenced: keyR[y][x - d]) and is not the winner itself
//      is appended to a queue, and the pixel is flagged; asw_exact_winners_kernel appends the winners of flagged pixels;
//   3. asw_exact_eval_kernel: one WAVE per queue entry re-evaluates that candidate's cost in fp64 with the reference's
//      own expression and summation order (window-row-major taps, `cost += w1*w2*TAD; tot += w1*w2`, no contraction:
//      _passive.cpp:57-88; the tap products side by side on the lanes, the two running sums by one lane in the reference's
//      order) from fp64 Lab values (colorconversion.hpp:67-69) and a proximity table built by the host's libm
//      (_passive.cpp:360-364), and atomicMin's the cost's bit pattern into the pixel's slot;
//   4. asw_exact_resolve_kernel: among the entries whose cost EQUALS the slot's minimum the smallest index wins (the
//      reference's strict `<` scan keeps the first minimum, _passive.cpp:90-93 / 243-246);
//   5. asw_exact_patch_kernel writes the winning index into the low word of the flagged pixels' WTA keys; decode,
//      left-right check and occlusion filling then run unchanged.
//
// Bit-identity (round 5): the weights are the reference's to the bit -- fp64 Lab values through glibc's powf, exp through
// glibc's exp (both restated in glibc_math.hip.h and proven equal to the running libm by oracle/libm_check.c), IEEE sqrt and
// division, the host libm's proximity table, no contraction -- so even candidates that differ in the last ulp of 40 (1 - k ulp)
// (every tap saturated: profiles/r05_exact_mode_audit.txt) resolve as in the reference, on hosts whose libm is the FMA build of
// glibc >= 2.28 (every x86-64 CPU since 2013).  What can still differ: a candidate the fp32 kernels put more than `tol` ulps
// from their winner although it wins in fp64 (none seen), and a queue overflow (counted).
#pragma once
#include "asw_kernels.hip.h"
#include "lab_kernels.hip.h"
#include "glibc_math.hip.h"

namespace ssamd {

struct AswExactArgs {
    const PixRec *recL, *recR;       // [H][W] records (bgrx = the raw bytes)
    const double *labL, *labR;       // [H][W][3] fp64 CIELab (rows the call touches)
    const double *prox;              // [win*win] proximity weights, host libm
    const uint32_t *kvol;            // [rows][W][nD] cost images of every candidate (undefined where x - d < 0)
    u64 *keyL, *keyR;                // [rows][W] WTA keys of the aggregation (keyR may be null)
    unsigned char *flagL, *flagR;    // [rows][W] pixel has near-ties
    u64 *entries;                    // queue: pix (32) | d (16) << 32 | sides (2) << 48
    unsigned int *counter;           // [0] entries appended (may exceed cap), [1] flagged left pixels, [2] flagged right pixels
    unsigned int cap;
    double *ecost;                   // [cap] fp64 cost of each entry
    u64 *costL, *costR;              // [rows][W] minimum fp64 cost bits over the pixel's entries
    uint32_t *idxL, *idxR;           // [rows][W] smallest index among the entries at that minimum
    int H, W, win, pad, minD, maxD, row0, rows;
    uint32_t tol;                    // key ulps
    float sat_abs;                   // absolute cost difference below which two saturated candidates are a near-tie of the reference's fp64
    double gammaC;
};

// sqrt as the host computes it (IEEE, correctly rounded): the device's f64 square root (v_rsq_f64 + Goldschmidt steps) is within an
// ulp but not always the correctly rounded one; with the exact residual e = x - y0^2 (one fma) the root is y0 + e / (2 y0) up to
// a relative 1e-32, and that sum rounds correctly unless it falls within 1e-16 ulp of a rounding boundary.
__device__ __forceinline__ double exact_sqrt(double x)
{
#pragma clang fp contract(off)
    const double y0 = sqrt(x);
    if (!(y0 > 0.0) || !(y0 < 1.0e300)) return y0;                   // 0, nan, inf
    const double e = fma(-y0, y0, x);
    return y0 + e / (2.0 * y0);
}

static constexpr unsigned EXACT_SIDE_L = 1u, EXACT_SIDE_R = 2u;
// cost images (asw_cost_key) at or above this hold 40 - cost (cost > 20)
static constexpr uint32_t EXACT_KEY_HIGH = 0xC0000000u - 0x41A00000u;      // 0x41A00000 = bits of 20.0f

// is candidate image `key` a near-tie of the winning image `kb` (kb <= key)?
__device__ __forceinline__ bool exact_near(uint32_t key, uint32_t kb, uint32_t tol, float sat_abs)
{
    if (key - kb <= tol) return true;
    if (key >= EXACT_KEY_HIGH) {
        const float inv = __uint_as_float(0xC0000000u - key);                                   // 40 - cost of the candidate
        if (kb >= EXACT_KEY_HIGH) return __uint_as_float(0xC0000000u - kb) - inv <= sat_abs;
        // the two images lie either side of cost = 20 (the winner's holds its cost, the candidate's 40 - cost): ulps do not compare
        return (40.0f - inv) - __uint_as_float(kb) <= 20.0f * 1.1920929e-7f * (float)tol;
    }
    return false;
}

__device__ __forceinline__ u64 exact_entry(uint32_t pix, int d, unsigned sides)
{
    return (u64)pix | ((u64)(uint32_t)d << 32) | ((u64)sides << 48);
}

// fp64 Lab of the rows [r0, r0 + npix / W) from the records' bytes, both images in one launch
__global__ __launch_bounds__(256) void bgr2lab_f64_pair_kernel(const PixRec *__restrict__ recL, const PixRec *__restrict__ recR,
                                                               double *__restrict__ labL, double *__restrict__ labR, long long npix)
{
    SSAMD_LAB_TABLES_IN_LDS(T)
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; q < 2 * npix; q += stride) {
        const bool right = q >= npix;
        const long long p = right ? q - npix : q;
        const uint32_t v = (right ? recR : recL)[p].bgrx;
        double L, a, b;
        bgr_to_lab_f64(v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff, L, a, b, T);
        double *const o = (right ? labR : labL) + 3 * p;
        o[0] = L; o[1] = a; o[2] = b;
    }
}

// 2. one thread per (x, d) element of an output row's slice of the key volume
__global__ __launch_bounds__(256) void asw_exact_flag_kernel(const AswExactArgs A)
{
    const int nD = A.maxD - A.minD + 1;
    const long long per_row = (long long)A.W * nD;
    const int yr = blockIdx.y;
    const uint32_t *const krow = A.kvol + (size_t)yr * per_row;
    const u64 *const kl = A.keyL + (size_t)yr * A.W, *const kr = A.keyR ? A.keyR + (size_t)yr * A.W : nullptr;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < per_row; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e / nD), d = A.minD + (int)(e - (long long)x * nD);
        if (x - d < 0) continue;                                   // not a candidate the reference evaluates (_passive.cpp:56)
        const uint32_t key = krow[e];
        unsigned sides = 0;
        // near-tie: within `tol` ulps of the winner's cost image -- or, on the saturated side of the image (cost > 20: the image
        // holds 40 - cost), within `sat_abs` of it in ABSOLUTE terms: the (N, S') pair resolves 40 - 1e-30 from 40 - 0, but the
        // reference's fp64 quotient carries a rounding noise of up to ~(win^2) ulps of 40 (2e-11 for a 35 x 35 window), so among
        // candidates closer than that its first minimum is decided by that noise and has to be recomputed
        const u64 bl = kl[x];
        if (bl != KEY_NONE && (int)(uint32_t)bl != d && exact_near(key, (uint32_t)(bl >> 32), A.tol, A.sat_abs)) sides |= EXACT_SIDE_L;
        if (kr) {
            const u64 br = kr[x - d];
            if (br != KEY_NONE && (int)(uint32_t)br != x && exact_near(key, (uint32_t)(br >> 32), A.tol, A.sat_abs)) sides |= EXACT_SIDE_R;
        }
        if (!sides) continue;
        const uint32_t pix = (uint32_t)yr * (uint32_t)A.W + (uint32_t)x;
        if (sides & EXACT_SIDE_L) A.flagL[pix] = 1;
        if (sides & EXACT_SIDE_R) A.flagR[pix - (uint32_t)d] = 1;
        const unsigned slot = atomicAdd(A.counter, 1u);
        if (slot < A.cap) A.entries[slot] = exact_entry(pix, d, sides);
    }
}

// ... and the winners of the flagged pixels themselves
__global__ __launch_bounds__(256) void asw_exact_winners_kernel(const AswExactArgs A)
{
    const long long n = (long long)A.rows * A.W;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        if (A.flagL[q]) {
            atomicAdd(A.counter + 1, 1u);
            const unsigned slot = atomicAdd(A.counter, 1u);
            if (slot < A.cap) A.entries[slot] = exact_entry((uint32_t)q, (int)(uint32_t)A.keyL[q], EXACT_SIDE_L);
        }
        if (A.keyR && A.flagR[q]) {
            atomicAdd(A.counter + 2, 1u);
            const int xr = (int)(q % A.W), xl = (int)(uint32_t)A.keyR[q];
            const unsigned slot = atomicAdd(A.counter, 1u);
            if (slot < A.cap) A.entries[slot] = exact_entry((uint32_t)(q + (xl - xr)), xl - xr, EXACT_SIDE_R);
        }
    }
}

// 3. the reference's cost of one candidate, in its arithmetic (_passive.cpp:37-50, 57-88).  One WAVE per queue entry: per
// window row the lanes evaluate the tap products w1*w2 and w1*w2*TAD side by side (two exp, two sqrt, two divisions in fp64
// each -- the expensive part, and independent of each other) into LDS, then lane 0 adds them to `cost` and `tot` one by one in
// the reference's order (window-row-major, left to right): the sums round exactly as a sequential loop's would, at 1 / 50 of
// a single thread's latency (a 35 x 35 window is 2450 dependent exp / sqrt chains: 1.1 ms for ONE entry on one lane).
static constexpr int EXACT_WAVES = 4;          // entries in flight per workgroup
__global__ __launch_bounds__(64 * EXACT_WAVES) void asw_exact_eval_kernel(const AswExactArgs A)
{
#pragma clang fp contract(off)
    __shared__ double s_w[EXACT_WAVES][256], s_c[EXACT_WAVES][256];       // (winSize <= 255)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *const sw = s_w[wv], *const sc = s_c[wv];
    const unsigned n = min(A.counter[0], A.cap);
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    for (unsigned e = blockIdx.x * EXACT_WAVES + wv; e < n; e += gridDim.x * EXACT_WAVES) {
        const u64 ent = A.entries[e];
        const uint32_t pix = (uint32_t)ent;
        const int d = (int)((ent >> 32) & 0xffff);
        const unsigned sides = (unsigned)(ent >> 48) & 3u;
        const int yr = (int)(pix / (uint32_t)W), x = (int)(pix - (uint32_t)yr * (uint32_t)W);
        const int y = A.row0 + yr, xr = x - d;
        const double *const cl = A.labL + 3 * ((size_t)y * W + x), *const cr = A.labR + 3 * ((size_t)y * W + xr);
        const double cl0 = cl[0], cl1 = cl[1], cl2 = cl[2], cr0 = cr[0], cr1 = cr[1], cr2 = cr[2];
        // tap columns j with both the left column x - p + j and the right column xr - p + j inside the image (xr <= x):
        // `continue` below 0, `break` from the width on (_passive.cpp:65-68)
        const int jlo = max(0, p - xr), jhi = min(win, W + p - x);
        double cost = 0.0, tot = 0.0;
        for (int i = 0; i < win; ++i) {
            const int ii = y - p + i;
            if (ii < 0) continue;
            if (ii >= H) break;
            const double *const pr = A.prox + i * win;
            const double *const rowL = A.labL + 3 * (size_t)ii * W, *const rowR = A.labR + 3 * (size_t)ii * W;
            const PixRec *const bL = A.recL + (size_t)ii * W, *const bR = A.recR + (size_t)ii * W;
            for (int j = jlo + lane; j < jhi; j += 64) {
                const int jj = xr - p + j, kk = x - p + j;
                const double *const tl = rowL + 3 * kk, *const tr = rowR + 3 * jj;
                const double a0 = tl[0] - cl0, a1 = tl[1] - cl1, a2 = tl[2] - cl2;
                const double b0 = tr[0] - cr0, b1 = tr[1] - cr1, b2 = tr[2] - cr2;
                const double w1 = pr[j] * glibc_exp(-exact_sqrt(a0 * a0 + a1 * a1 + a2 * a2) / A.gammaC);
                const double w2 = pr[j] * glibc_exp(-exact_sqrt(b0 * b0 + b1 * b1 + b2 * b2) / A.gammaC);
                const int tad = min(40, (int)__builtin_amdgcn_sad_u8(bL[kk].bgrx, bR[jj].bgrx, 0u));
                const double ww = w1 * w2;
                sw[j] = ww;
                sc[j] = ww * tad;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0)
                for (int j = jlo; j < jhi; ++j) {
                    cost += sc[j];
                    tot += sw[j];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            const double c = cost / tot;
            A.ecost[e] = c;
            const u64 bits = (u64)__double_as_longlong(c);           // costs are >= 0: the bit patterns order like the values
            if (sides & EXACT_SIDE_L) atomicMin(A.costL + pix, bits);
            if (sides & EXACT_SIDE_R) atomicMin(A.costR + (pix - (uint32_t)d), bits);
        }
    }
}

// 4. the first minimum wins: smallest disparity (left-referenced), smallest left column (right-referenced)
__global__ __launch_bounds__(256) void asw_exact_resolve_kernel(const AswExactArgs A)
{
    const unsigned n = min(A.counter[0], A.cap);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const u64 ent = A.entries[e];
        const uint32_t pix = (uint32_t)ent;
        const int d = (int)((ent >> 32) & 0xffff);
        const unsigned sides = (unsigned)(ent >> 48) & 3u;
        const u64 bits = (u64)__double_as_longlong(A.ecost[e]);
        if ((sides & EXACT_SIDE_L) && A.costL[pix] == bits) atomicMin(A.idxL + pix, (uint32_t)d);
        if ((sides & EXACT_SIDE_R) && A.costR[pix - (uint32_t)d] == bits) atomicMin(A.idxR + (pix - (uint32_t)d), pix % (uint32_t)A.W);
    }
}

// 5. flagged pixels whose every candidate made it into the queue get the fp64 winner's index in their key
__global__ __launch_bounds__(256) void asw_exact_patch_kernel(const AswExactArgs A)
{
    if (A.counter[0] > A.cap) return;        // queue overflow (counted; the caller reports it): incomplete candidate sets, keys stay as they are
    const long long n = (long long)A.rows * A.W;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        if (A.flagL[q] && A.idxL[q] != 0xffffffffu) A.keyL[q] = (A.keyL[q] & 0xffffffff00000000ull) | (u64)A.idxL[q];
        if (A.keyR && A.flagR[q] && A.idxR[q] != 0xffffffffu) A.keyR[q] = (A.keyR[q] & 0xffffffff00000000ull) | (u64)A.idxR[q];
    }
}

}  // namespace ssamd
