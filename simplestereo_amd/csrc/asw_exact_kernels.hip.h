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
    // weight tables of flagged pixels (asw_exact_wtab_kernel): the ~180 queue entries of a flagged left pixel share their LEFT support
    // weights (w1 depends on the left pixel only, _passive.cpp:47-50), those of a flagged right pixel their right ones
    uint32_t *wslotL, *wslotR;       // [rows][W] table slot of a flagged pixel (written for flagged pixels only; >= wcap: none)
    uint32_t *wpix;                  // [wcap] pixel | side << 31 of each slot
    double *wtab;                    // [wcap][win * win] prox * exp(-dLab / gammaC) towards the slot's centre pixel, 0 outside the image
    unsigned int wcap;               // slots available (counter[6] = slots asked for)
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
        uint32_t pix = 0, rkey = 0xffffffffu;
        int d = 0;
        if (e < n) {
            const u64 ent = A.raw.entries[e];
            const uint32_t key = A.raw.ekeys[e];
            rkey = key;
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
        asw_exact_push_wave(A.q, s2 != 0, pix, d, s2, rkey);
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

// 1d. Pixels with TWO OR MORE candidates whose fp32 cost is EXACTLY 0 (N = 0; marked by asw_exact_select / asw_exact_merge -- a lone zero,
// the true match of a synthetic pair, needs nothing).  Candidates with image 0 are not queued: deep inside the black margins of a
// rectified frame every candidate of a pixel is one -- 3e7 entries at 1080p.  The reference's cost of such a candidate is
// exactly 0.0 iff every in-image tap has TAD = 0 (weights are positive in fp64), and costs are >= 0: if the fp32 winner -- the smallest
// index among the candidates with image 0 -- passes that integer test, it IS the reference's first minimum (smaller indices have N > 0
// in fp32, hence a positive cost in fp64) and the pixel is settled without a single fp64 operation.  If it does not (fp32 weights
// that underflowed on taps with TAD > 0: small gammaC), the pixel gets ALL its candidates, as an escalated pixel does.
// Workgroup = 64 consecutive pixels of one output row (most groups hold no such pixel and leave after one ballot).  The anchor image's
// window rows of the whole segment are staged in LDS once, the other image's once per distinct disparity of the segment's zero-cost
// winners (one, in a margin): a pixel's 1 225 compares then read LDS (a first form read them from HBM per pixel: + 7.5 ms on a 1080p
// frame with margins).  side 0: anchors are left pixels x, the other image's column is x - d; side 1 (right-referenced winners): anchors
// are right pixels, the other column is x + d.
static constexpr int EXACT_ZSEG = 64, EXACT_ZWIN_MAX = 64;          // windows up to 63 x 63 go through LDS; larger ones take the escalation path
__global__ __launch_bounds__(256) void asw_exact_zero_kernel(const AswExactArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char zsmem[];
    const AswExactQueue &Q = A.q;
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    const int segs_per_row = (W + EXACT_ZSEG - 1) / EXACT_ZSEG;
    const int tid = threadIdx.x;
    const int TW = EXACT_ZSEG + 2 * p;                                  // tile columns
    uint32_t *const tA = reinterpret_cast<uint32_t *>(zsmem);           // [win][TW] anchor image, bit 31 = outside the image
    uint32_t *const tB = tA + win * TW;                                 // [win][TW] other image for the current disparity
    __shared__ int s_idx[EXACT_ZSEG], s_state[EXACT_ZSEG];              // winner index; 0 not a zero-cost winner, 1 pending, 2 all-zero, 3 not all-zero
    __shared__ int s_cur;
    for (long long seg = blockIdx.x; seg < (long long)A.rows * segs_per_row; seg += gridDim.x) {
        const int yr = (int)(seg / segs_per_row), x0 = (int)(seg - (long long)yr * segs_per_row) * EXACT_ZSEG, y = A.row0 + yr;
        if (!Q.zrow[yr]) continue;                                      // no tile-local winner of this row costs exactly 0 (the usual case)
        for (int side = 0; side < (A.keyR ? 2 : 1); ++side) {
            __syncthreads();
            if (tid < EXACT_ZSEG) {
                const int x = x0 + tid;
                bool z = false;
                int idx = 0;
                if (x < W) {
                    const long long q = (long long)yr * W + x;
                    if (side == 0) {                                    // marked by the aggregation kernels: two or more zero-cost candidates
                        z = Q.zeroL[q] != 0;
                        if (z) {
                            if (A.keyL) { const u64 g = A.keyL[q]; z = g != KEY_NONE && (uint32_t)(g >> 32) == 0u; idx = (int)(uint32_t)g; }
                            else idx = (int)A.disp[q];
                        }
                    } else {
                        z = Q.zeroR[q] != 0;
                        if (z) { const u64 g = A.keyR[q]; z = g != KEY_NONE && (uint32_t)(g >> 32) == 0u; idx = (int)(uint32_t)g; }
                    }
                }
                s_idx[tid] = idx;
                s_state[tid] = z ? 1 : 0;
            }
            if (__syncthreads_or(tid < EXACT_ZSEG && s_state[tid] == 1) == 0) continue;
            if (win >= EXACT_ZWIN_MAX) {                                // (LDS tiles sized for windows below 64: larger ones are re-evaluated in fp64)
                if (tid < EXACT_ZSEG && s_state[tid] == 1) s_state[tid] = 3;
            } else {
                const PixRec *const imgA = side ? A.recR : A.recL, *const imgB = side ? A.recL : A.recR;
                // (tiles are staged eight loads at a time: one load per loop trip serialises on the HBM latency -- a first form spent 1.2 ms here)
                auto stage = [&](uint32_t *tile, const PixRec *img, int shift) {
                    for (int k0 = 0; k0 < win * TW; k0 += 8 * 256) {
                        uint32_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int k = k0 + u * 256 + tid, i = k / TW, c = k - i * TW, ii = y - p + i, cc = x0 - p + c + shift;
                            v[u] = (k < win * TW && (unsigned)ii < (unsigned)H && (unsigned)cc < (unsigned)W) ? img[(size_t)ii * W + cc].bgrx : 0x80000000u;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int k = k0 + u * 256 + tid;
                            if (k < win * TW) tile[k] = v[u];
                        }
                    }
                };
                stage(tA, imgA, 0);                                     // anchor tile: rows y - p .., columns x0 - p ..
                for (;;) {
                    __syncthreads();
                    if (tid == 0) {
                        int cur = -1;
                        for (int k = 0; k < EXACT_ZSEG; ++k)
                            if (s_state[k] == 1) { cur = side ? s_idx[k] - (x0 + k) : s_idx[k]; break; }
                        s_cur = cur;                                    // disparity of the next pending pixel (-1: none left)
                    }
                    __syncthreads();
                    const int d = s_cur;
                    if (d < 0) break;
                    const int shift = side ? d : -d;                    // other image's column = anchor column + shift
                    stage(tB, imgB, shift);
                    __syncthreads();
                    // pixel = tid & 63, a quarter of the window rows per thread; taps outside either image are not taps (_passive.cpp:60-68)
                    const int px = tid & (EXACT_ZSEG - 1), part = tid >> 6;
                    const bool mine = s_state[px] == 1 && (side ? s_idx[px] - (x0 + px) : s_idx[px]) == d;
                    bool nz = false;
                    if (mine) {
                        uint32_t acc = 0;                               // branch-free: (a ^ b) where both taps are inside their images
                        for (int i = part; i < win; i += 4) {
                            const uint32_t *const ra = tA + i * TW + px, *const rb = tB + i * TW + px;
#pragma unroll 8
                            for (int j = 0; j < win; ++j) {
                                const uint32_t a = ra[j], b = rb[j];
                                acc |= ((a | b) >> 31) ? 0u : (a ^ b);
                            }
                        }
                        nz = acc != 0u;
                    }
                    __syncthreads();
                    if (mine && part == 0) s_state[px] = 2;
                    __syncthreads();
                    if (mine && nz) s_state[px] = 3;                    // (any of the four parts)
                }
            }
            __syncthreads();
            if (tid < EXACT_ZSEG && s_state[tid] == 3) {
                // not all-zero: every candidate of the pixel goes to the queue (rare: underflowed fp32 weights, or a window beyond 63)
                const int xq = x0 + tid, widx = s_idx[tid];
                const long long pq = (long long)yr * W + xq;
                if (side == 0) {
                    const int dhi = min(A.maxD, xq), cnt = dhi - A.minD;
                    if (cnt > 0) {
                        unsigned slot = atomicAdd(Q.counter, (unsigned)cnt);
                        for (int d = A.minD; d <= dhi; ++d)
                            if (d != widx) { if (slot < Q.cap) Q.entries[slot] = exact_entry((uint32_t)pq, d, EXACT_SIDE_L); ++slot; }
                        Q.flagL[pq] = 1;
                    }
                } else {
                    const int dhi = min(A.maxD, W - 1 - xq), cnt = dhi - A.minD;
                    if (cnt > 0) {
                        unsigned slot = atomicAdd(Q.counter, (unsigned)cnt);
                        for (int d = A.minD; d <= dhi; ++d)
                            if (xq + d != widx) { if (slot < Q.cap) Q.entries[slot] = exact_entry((uint32_t)(pq + d), d, EXACT_SIDE_R); ++slot; }
                        Q.flagR[pq] = 1;
                    }
                }
            }
        }
    }
}

// a weight-table slot for every flagged pixel of the wave (mask = ballot(flagged)): one atomic per wave
__device__ __forceinline__ void exact_assign_wslot(const AswExactArgs &A, bool flagged, u64 mask, uint32_t pix, uint32_t side)
{
    if (mask == 0 || A.wcap == 0) return;
    const int leader = (int)__builtin_ctzll(mask);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(A.q.counter + 6, (unsigned)__builtin_popcountll(mask));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    if (flagged) {
        const unsigned slot = base + (unsigned)__builtin_popcountll(mask & (((u64)1 << lane) - 1));
        (side ? A.wslotR : A.wslotL)[pix] = slot;
        if (slot < A.wcap) A.wpix[slot] = pix | (side << 31);
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
        exact_assign_wslot(A, fl, ml, (uint32_t)q, 0u);
        asw_exact_push_wave(Q, fl, (uint32_t)q, dwin, EXACT_SIDE_L, fl && A.keyL ? (uint32_t)(A.keyL[q] >> 32) : 0xffffffffu);
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
            exact_assign_wslot(A, fr, mr, (uint32_t)q, 1u);
            asw_exact_push_wave(Q, fr, (uint32_t)(q + (xl - xr)), xl - xr, EXACT_SIDE_R, fr ? (uint32_t)(A.keyR[q] >> 32) : 0xffffffffu);
        }
    }
}

// 2b. the support weights of every flagged pixel towards its window, once: prox * exp(-dLab / gammaC) in the reference's expression
// (_passive.cpp:47-50) -- the same instructions on the same operands as asw_exact_eval_kernel would execute per entry, so the same bits.
// One wave per table slot.
__global__ __launch_bounds__(256) void asw_exact_wtab_kernel(const AswExactArgs A)
{
#pragma clang fp contract(off)
    __shared__ uint64_t s_exp[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) s_exp[k] = gm_exp_tab[k];
    __syncthreads();
    if (exact_overflowed(A)) return;
    const unsigned n = min(A.q.counter[6], A.wcap);
    const int lane = threadIdx.x & 63, W = A.W, H = A.H, win = A.win, p = A.pad, win2 = win * win;
    for (unsigned slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n; slot += gridDim.x * 4) {
        const uint32_t pw = A.wpix[slot], pix = pw & 0x7fffffffu;
        const double *const lab = (pw >> 31) ? A.labR : A.labL;
        const int yr = (int)(pix / (uint32_t)W), x = (int)(pix - (uint32_t)yr * (uint32_t)W), y = A.row0 + yr;
        const double *const c = lab + 3 * ((size_t)y * W + x);
        const double c0 = c[0], c1 = c[1], c2 = c[2];
        double *const out = A.wtab + (size_t)slot * win2;
        for (int t = lane; t < win2; t += 64) {
            const int i = t / win, j = t - i * win, ii = y - p + i, cc = x - p + j;
            double w = 0.0;
            if ((unsigned)ii < (unsigned)H && (unsigned)cc < (unsigned)W) {
                const double *const tp = lab + 3 * ((size_t)ii * W + cc);
                const double a0 = tp[0] - c0, a1 = tp[1] - c1, a2 = tp[2] - c2;
                w = A.prox[t] * glibc_exp_t(-exact_sqrt(a0 * a0 + a1 * a1 + a2 * a2) / A.gammaC, s_exp);
            }
            out[t] = w;
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
    // static stride over a grid of exactly the waves the chip holds at once (the launcher sizes it: five 4-wave groups per CU): a larger
    // grid left its last round of waves to run after the others had finished (6 144 waves for 5 120 slots: ~1.7 x the ideal time).
    // (Handing entries out through an atomic counter + readfirstlane hung the kernel on this compiler: not pursued.)
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
        // weight tables (asw_exact_wtab_kernel): w1 of a flagged LEFT pixel, w2 of a flagged RIGHT pixel -- wave-uniform
        const uint32_t s1 = A.wcap && A.q.flagL && A.q.flagL[pix] ? A.wslotL[pix] : 0xffffffffu;
        const uint32_t s2 = A.wcap && A.q.flagR && A.q.flagR[pix - (uint32_t)d] ? A.wslotR[pix - (uint32_t)d] : 0xffffffffu;
        const double *const t1 = s1 < A.wcap ? A.wtab + (size_t)s1 * win * win : nullptr;
        const double *const t2 = s2 < A.wcap ? A.wtab + (size_t)s2 * win * win : nullptr;
        double acc = 0.0;                           // lane 0: cost, lane 1: tot
        // entries whose fp32 cost is 0 (EXACT_HINT_ZERO): if every in-image tap has TAD = 0 the reference's cost is exactly 0.0 --
        // `cost += w1*w2*0` for positive weights -- and nothing else has to be computed
        bool all_zero = false;
        if ((unsigned)(ent >> 48) & EXACT_HINT_ZERO) {
            bool nz = false;
            for (int t = lane; t < ntap; t += 64) {
                const int q = t / ncol, i = ilo + q, j = jlo + (t - q * ncol);
                const int ii = y - p + i, jj = xr - p + j, kk = x - p + j;
                nz = nz || __builtin_amdgcn_sad_u8(A.recL[(size_t)ii * W + kk].bgrx, A.recR[(size_t)ii * W + jj].bgrx, 0u) != 0u;
            }
            all_zero = __builtin_amdgcn_ballot_w64(nz) == 0;
        }
        for (int t0 = 0; t0 < (all_zero ? 0 : ntap); t0 += EXACT_CHUNK) {
            const int nt = min(EXACT_CHUNK, ntap - t0);
            for (int t = lane; t < nt; t += 64) {
                const int q = (t0 + t) / ncol, i = ilo + q, j = jlo + (t0 + t - q * ncol);
                const int ii = y - p + i, jj = xr - p + j, kk = x - p + j;
                const double pr = A.prox[i * win + j];
                double w1, w2;
                if (t1) w1 = t1[i * win + j];
                else {
                    const double *const tl = A.labL + 3 * ((size_t)ii * W + kk);
                    const double a0 = tl[0] - cl0, a1 = tl[1] - cl1, a2 = tl[2] - cl2;
                    w1 = pr * glibc_exp_t(-exact_sqrt(a0 * a0 + a1 * a1 + a2 * a2) / A.gammaC, s_exp);
                }
                if (t2) w2 = t2[i * win + j];
                else {
                    const double *const tr = A.labR + 3 * ((size_t)ii * W + jj);
                    const double b0 = tr[0] - cr0, b1 = tr[1] - cr1, b2 = tr[2] - cr2;
                    w2 = pr * glibc_exp_t(-exact_sqrt(b0 * b0 + b1 * b1 + b2 * b2) / A.gammaC, s_exp);
                }
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
            const double c = all_zero ? 0.0 : acc / tot;
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
