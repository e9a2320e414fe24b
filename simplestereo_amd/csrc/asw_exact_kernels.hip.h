// K1x: fp64 tie-break pass of the ASW matchers (`exact` mode; gfx950).
//
// The reference aggregates in double (_passive.cpp:23, 56-95); the aggregation kernels accumulate (N, S') in fp32 and
// resolve costs to ~1e-6 relative.  Where two candidates of a pixel are closer than that, the fp32 argmin may pick the
// other one (0.006 % of the pixels of the bench frame, 0.09 % on the class-default lawn photograph).  This pass makes the
// argmin the reference's wherever fp64 can tell the candidates apart:
//
//   1. (round 6) the aggregation kernels themselves, in their epilogue, queue every candidate that is a NEAR-TIE of its pixel's
//      winner and flag the pixel (asw_exact_select / asw_exact_merge, asw_kernels.hip.h): the keys are still in registers and
//      the tile-local winners in LDS there.  Round 5 dumped the cost image of EVERY candidate to HBM (H*W*nD*4 bytes: 1.6 GB at
//      1080p / 193, 9.1 GB at 4K) and re-read it in a flag kernel to find 0.006 % of it;
//      -- directly when the workgroup's winner is final (one disparity chunk, no right pass), else into a RAW queue with their
//      cost images, which asw_exact_filter_kernel re-tests against the final winners (tile-local winners are only upper bounds);
//   2. asw_exact_winners_kernel appends the winners of flagged pixels and initialises their result slots;
//   3. asw_exact_eval_kernel: one WAVE per queue entry re-evaluates that candidate's cost in fp64 with the reference's
//      own expression and summation order (window-row-major taps, `cost += w1*w2*TAD; tot += w1*w2`, no contraction:
//      _passive.cpp:57-88; the tap products side by side on the lanes, the two running sums by one lane in the reference's
//      order) from fp64 Lab values (colorconversion.hpp:67-69) and the fp64 proximity table (_passive.cpp:360-364), and
//      atomicMin's the cost's bit pattern into the pixel's slot;
//   4. asw_exact_resolve_kernel: among the entries whose cost EQUALS the slot's minimum the smallest index wins (the
//      reference's strict `<` scan keeps the first minimum, _passive.cpp:90-93 / 243-246);
//   5. asw_exact_patch_kernel writes the winning index into the flagged pixels' WTA keys -- or straight into the disparity map
//      when the aggregation wrote that itself (one disparity chunk, no right pass); decode, left-right check and occlusion
//      filling then run unchanged.
//
// Bit-identity: the weights are the reference's to the bit -- fp64 Lab values through glibc's powf, exp through glibc's exp
// (both restated in glibc_math.hip.h and proven equal to the running libm by oracle/libm_check.c), IEEE sqrt and division, no
// contraction -- so even candidates that differ in the last ulp of 40 (1 - k ulp) (every tap saturated:
// profiles/r05_exact_mode_audit.txt) resolve as in the reference.  What can still differ: a candidate the fp32 kernels put
// more than the near-tie band from their winner although it wins in fp64 (none seen; the band grows with the window,
// ssamd_api.hip), and a queue overflow (counted, reported, the fp32 map is kept).
#pragma once
#include "asw_kernels.hip.h"
#include "lab_kernels.hip.h"
#include "glibc_math.hip.h"

namespace ssamd {

struct AswExactArgs {
    const PixRec *recL, *recR;       // [H][W] records (bgrx = the raw bytes)
    const double *labL, *labR;       // [H][W][3] fp64 CIELab (rows the call touches)
    const double *prox;              // [win*win] fp64 proximity weights
    u64 *keyL, *keyR;                // [rows][W] WTA keys of the aggregation (keyL null: `disp` holds the left winners; keyR may be null)
    int16_t *disp;                   // [rows][W] the map the aggregation kernel wrote itself (keyL == nullptr)
    AswExactQueue q;                 // queue, counter (q.counter[0]; [1] / [2]: flagged left / right pixels), pixel flags
    AswExactQueue raw;               // MERGING calls: what the aggregation kernels selected against tile-local winners, with cost images
                                     //   (raw.entries == nullptr: a direct call, the kernels filled q themselves)
    double *ecost;                   // [cap] fp64 cost of each entry
    u64 *costL, *costR;              // [rows][W] minimum fp64 cost bits over the pixel's entries (initialised for flagged pixels)
    uint32_t *idxL, *idxR;           // [rows][W] smallest index among the entries at that minimum
    int H, W, win, pad, minD, maxD, row0, rows;
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

// has a queue of this call overflowed?  (then candidate sets are incomplete: every later kernel of the pass returns, the fp32 map stays)
__device__ __forceinline__ bool exact_overflowed(const AswExactArgs &A)
{
    return A.q.counter[0] > A.q.cap || (A.raw.entries && A.raw.counter[0] > A.raw.cap);
}

// 1b. MERGING calls: the raw candidates against the FINAL winners.  keyL == nullptr cannot happen here (direct calls have no raw queue).
__global__ __launch_bounds__(256) void asw_exact_filter_kernel(const AswExactArgs A)
{
    const unsigned n = min(A.raw.counter[0], A.raw.cap);
    const unsigned nloop = (n + 255u) & ~255u;                          // uniform trip counts: the push is wave-aggregated
    for (unsigned e0 = blockIdx.x * blockDim.x; e0 < nloop; e0 += gridDim.x * blockDim.x) {
        const unsigned e = e0 + threadIdx.x;
        unsigned s2 = 0;
        uint32_t pix = 0;
        int d = 0;
        if (e < n) {
            const u64 ent = A.raw.entries[e];
            const uint32_t key = A.raw.ekeys[e];
            pix = (uint32_t)ent;
            d = (int)((ent >> 32) & 0xffff);
            const unsigned sides = (unsigned)(ent >> 48) & 3u;
            const uint32_t esc = 0xC0000000u - __float_as_uint(2.0f * A.q.sat_abs);      // (escalated pixels get every candidate from asw_exact_escalate_kernel)
            if (sides & EXACT_SIDE_L) {
                const u64 g = A.keyL[pix];
                if ((int)(uint32_t)g != d && (uint32_t)(g >> 32) < esc && exact_near(key, (uint32_t)(g >> 32), A.q.tol, A.q.sat_abs)) s2 |= EXACT_SIDE_L;
            }
            if ((sides & EXACT_SIDE_R) && A.keyR) {
                const u64 g = A.keyR[pix - (uint32_t)d];
                if ((uint32_t)g != pix % (uint32_t)A.W && (uint32_t)(g >> 32) < esc && exact_near(key, (uint32_t)(g >> 32), A.q.tol, A.q.sat_abs)) s2 |= EXACT_SIDE_R;
            }
        }
        asw_exact_push_wave(A.q, s2 != 0, pix, d, s2);
    }
}

// 1c. MERGING calls: pixels whose FINAL winner is within 2 sat_abs of the cap 40 (every tap with a weight saturated, or nearly) get
// ALL their candidates: the aggregation kernels did not queue the fully saturated ones (AswExactQueue::deep), and any of them may
// be the reference's first minimum here.  One thread per pixel; rare (occlusions, noise).
__global__ __launch_bounds__(256) void asw_exact_escalate_kernel(const AswExactArgs A)
{
    const AswExactQueue &Q = A.q;
    const uint32_t esc = 0xC0000000u - __float_as_uint(2.0f * Q.sat_abs);      // images at or above: 40 - cost <= 2 sat_abs
    const long long n = (long long)A.rows * A.W;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(q % A.W);
        const u64 gl = A.keyL[q];
        if (gl != KEY_NONE && (uint32_t)(gl >> 32) >= esc) {
            const int dwin = (int)(uint32_t)gl, dhi = min(A.maxD, x);            // candidates d = minD .. min(maxD, x) (_passive.cpp:56)
            const int cnt = dhi - A.minD;                                       // all but the winner (asw_exact_winners_kernel adds it)
            if (cnt > 0) {
                unsigned slot = atomicAdd(Q.counter, (unsigned)cnt);
                for (int d = A.minD; d <= dhi; ++d)
                    if (d != dwin) {
                        if (slot < Q.cap) Q.entries[slot] = exact_entry((uint32_t)q, d, EXACT_SIDE_L);
                        ++slot;
                    }
                Q.flagL[q] = 1;
            }
        }
        if (A.keyR) {
            const u64 gr = A.keyR[q];
            if (gr != KEY_NONE && (uint32_t)(gr >> 32) >= esc) {
                const int xwin = (int)(uint32_t)gr;                              // right pixel x: left columns x + d, d = minD .. maxD, below W
                const int dhi = min(A.maxD, A.W - 1 - x), cnt = dhi - A.minD;
                if (cnt > 0) {
                    unsigned slot = atomicAdd(Q.counter, (unsigned)cnt);
                    for (int d = A.minD; d <= dhi; ++d)
                        if (x + d != xwin) {
                            if (slot < Q.cap) Q.entries[slot] = exact_entry((uint32_t)(q + d), d, EXACT_SIDE_R);
                            ++slot;
                        }
                    Q.flagR[q] = 1;
                }
            }
        }
    }
}

// 2. the winners of the flagged pixels join their near-ties in the queue; their result slots are initialised here (only
// flagged pixels have entries, so nothing else is ever read: no 24 B / pixel memset)
__global__ __launch_bounds__(256) void asw_exact_winners_kernel(const AswExactArgs A)
{
    const AswExactQueue &Q = A.q;
    const long long n = (long long)A.rows * A.W;
    for (long long q0 = (long long)blockIdx.x * blockDim.x; q0 < n; q0 += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + threadIdx.x;
        const bool inb = q < n;
        const bool fl = inb && Q.flagL[q];
        if (fl) {
            A.costL[q] = ~0ull;
            A.idxL[q] = 0xffffffffu;
        }
        const int dwin = fl ? (A.keyL ? (int)(uint32_t)A.keyL[q] : (int)A.disp[q]) : 0;
        const u64 ml = __builtin_amdgcn_ballot_w64(fl);
        if (ml && fl && (int)__builtin_ctzll(ml) == (int)(threadIdx.x & 63)) atomicAdd(Q.counter + 1, (unsigned)__builtin_popcountll(ml));
        asw_exact_push_wave(Q, fl, (uint32_t)q, dwin, EXACT_SIDE_L);
        if (A.keyR) {
            const bool fr = inb && Q.flagR[q];
            int xl = 0, xr = 0;
            if (fr) {
                A.costR[q] = ~0ull;
                A.idxR[q] = 0xffffffffu;
                xr = (int)(q % A.W);
                xl = (int)(uint32_t)A.keyR[q];
            }
            const u64 mr = __builtin_amdgcn_ballot_w64(fr);
            if (mr && fr && (int)__builtin_ctzll(mr) == (int)(threadIdx.x & 63)) atomicAdd(Q.counter + 2, (unsigned)__builtin_popcountll(mr));
            asw_exact_push_wave(Q, fr, (uint32_t)(q + (xl - xr)), xl - xr, EXACT_SIDE_R);
        }
    }
}

// 3. the reference's cost of one candidate, in its arithmetic (_passive.cpp:37-50, 57-88).  One WAVE per queue entry: the taps of
// the window that lie inside both images are numbered in the reference's order (window-row-major, left to right) and dealt to the
// lanes EXACT_CHUNK at a time: every lane evaluates the products w1*w2 and w1*w2*TAD of its taps (two exp, two sqrt, two divisions
// in fp64 each -- the expensive part, independent of each other, loads of all 64 lanes in flight together) into LDS; then lane 0
// adds the chunk's products to `cost` and lane 1 the weights to `tot`, one by one in that order: the sums round exactly as the
// reference's sequential loop does.  Round 6: flat tap numbering instead of one window row (35 of 64 lanes busy, a dependent
// load -> exp -> LDS -> serial-sum chain per row) and glibc's exp table in LDS instead of constant memory (two dependent global
// loads per exp): 784 -> see profiles/r06_exact_mode_cost.txt for the 24 141 candidates of the bench frame.
static constexpr int EXACT_WAVES = 4;          // entries in flight per workgroup
static constexpr int EXACT_CHUNK = 256;        // taps per chunk and wave (4 KB of LDS: up to six waves per SIMD at 82 VGPRs)
__global__ __launch_bounds__(64 * EXACT_WAVES) void asw_exact_eval_kernel(const AswExactArgs A)
{
#pragma clang fp contract(off)
    __shared__ double s_w[EXACT_WAVES][EXACT_CHUNK], s_c[EXACT_WAVES][EXACT_CHUNK];
    __shared__ uint64_t s_exp[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) s_exp[k] = gm_exp_tab[k];
    __syncthreads();
    if (exact_overflowed(A)) return;               // queue overflow: incomplete candidate sets, the fp32 map is kept (asw_exact_patch_kernel)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *const sw = s_w[wv], *const sc = s_c[wv];
    const unsigned n = min(A.q.counter[0], A.q.cap);
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    for (unsigned e = blockIdx.x * EXACT_WAVES + wv; e < n; e += gridDim.x * EXACT_WAVES) {
        const u64 ent = A.q.entries[e];
        const uint32_t pix = (uint32_t)ent;
        const int d = (int)((ent >> 32) & 0xffff);
        const unsigned sides = (unsigned)(ent >> 48) & 3u;
        const int yr = (int)(pix / (uint32_t)W), x = (int)(pix - (uint32_t)yr * (uint32_t)W);
        const int y = A.row0 + yr, xr = x - d;
        const double *const cl = A.labL + 3 * ((size_t)y * W + x), *const cr = A.labR + 3 * ((size_t)y * W + xr);
        const double cl0 = cl[0], cl1 = cl[1], cl2 = cl[2], cr0 = cr[0], cr1 = cr[1], cr2 = cr[2];
        // tap columns j with both the left column x - p + j and the right column xr - p + j inside the image (xr <= x):
        // `continue` below 0, `break` from the width on (_passive.cpp:65-68); window rows i likewise (:60-62)
        const int jlo = max(0, p - xr), jhi = min(win, W + p - x), ncol = jhi - jlo;
        const int ilo = max(0, p - y), ihi = min(win, H + p - y);
        const int ntap = (ihi - ilo) * ncol;
        double acc = 0.0;                           // lane 0: cost, lane 1: tot
        for (int t0 = 0; t0 < ntap; t0 += EXACT_CHUNK) {
            const int nt = min(EXACT_CHUNK, ntap - t0);
            for (int t = lane; t < nt; t += 64) {
                const int q = (t0 + t) / ncol, i = ilo + q, j = jlo + (t0 + t - q * ncol);
                const int ii = y - p + i, jj = xr - p + j, kk = x - p + j;
                const double *const tl = A.labL + 3 * ((size_t)ii * W + kk), *const tr = A.labR + 3 * ((size_t)ii * W + jj);
                const double pr = A.prox[i * win + j];
                const double a0 = tl[0] - cl0, a1 = tl[1] - cl1, a2 = tl[2] - cl2;
                const double b0 = tr[0] - cr0, b1 = tr[1] - cr1, b2 = tr[2] - cr2;
                const double w1 = pr * glibc_exp_t(-exact_sqrt(a0 * a0 + a1 * a1 + a2 * a2) / A.gammaC, s_exp);
                const double w2 = pr * glibc_exp_t(-exact_sqrt(b0 * b0 + b1 * b1 + b2 * b2) / A.gammaC, s_exp);
                const int tad = min(40, (int)__builtin_amdgcn_sad_u8(A.recL[(size_t)ii * W + kk].bgrx, A.recR[(size_t)ii * W + jj].bgrx, 0u));
                const double ww = w1 * w2;
                sw[t] = ww;
                sc[t] = ww * tad;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 2) {
                // the adds are a dependent chain in the reference's order; the LDS reads are not: sixteen in flight per batch
                const double *const src = lane ? sw : sc;
                int t = 0;
                for (; t + 16 <= nt; t += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = src[t + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += v[u];
                }
                for (; t < nt; ++t) acc += src[t];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const double tot = __shfl(acc, 1);
        if (lane == 0) {
            const double c = acc / tot;
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
    if (exact_overflowed(A)) return;
    const unsigned n = A.q.counter[0];
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const u64 ent = A.q.entries[e];
        const uint32_t pix = (uint32_t)ent;
        const int d = (int)((ent >> 32) & 0xffff);
        const unsigned sides = (unsigned)(ent >> 48) & 3u;
        const u64 bits = (u64)__double_as_longlong(A.ecost[e]);
        if ((sides & EXACT_SIDE_L) && A.costL[pix] == bits) atomicMin(A.idxL + pix, (uint32_t)d);
        if ((sides & EXACT_SIDE_R) && A.costR[pix - (uint32_t)d] == bits) atomicMin(A.idxR + (pix - (uint32_t)d), pix % (uint32_t)A.W);
    }
}

// 5. flagged pixels whose every candidate made it into the queue get the fp64 winner's index: in their key, or in the map itself
__global__ __launch_bounds__(256) void asw_exact_patch_kernel(const AswExactArgs A)
{
    if (exact_overflowed(A)) return;             // queue overflow (counted; the caller reports it): incomplete candidate sets, the map stays as it is
    const long long n = (long long)A.rows * A.W;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        if (A.q.flagL[q] && A.idxL[q] != 0xffffffffu) {
            if (A.keyL) A.keyL[q] = (A.keyL[q] & 0xffffffff00000000ull) | (u64)A.idxL[q];
            else A.disp[q] = (int16_t)A.idxL[q];
        }
        if (A.keyR && A.q.flagR[q] && A.idxR[q] != 0xffffffffu) A.keyR[q] = (A.keyR[q] & 0xffffffff00000000ull) | (u64)A.idxR[q];
    }
}

}  // namespace ssamd
