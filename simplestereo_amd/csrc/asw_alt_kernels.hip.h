// K1b: the "alternate pixel" ASW mode sketched in the reference's docstring (passive.py:43-46, a todo
// without code: "compute disparity map on every other pixel with the traditional algorithm, then fill
// the remaining pixels using left-right disparity boundaries"), as an opt-in flag.
//
// Our definition (there is no reference implementation to match; oracle/oracle.py restates it on the
// CPU with the reference's fp64 costs):
//   1. EVEN image rows are matched exactly (asw_aggregate_kernel launched with a row step of 2 --
//      rows are the independent jobs of this path, _passive.cpp:372-374).
//   2. A pixel (y, x) of an ODD row takes its candidates from its two exact neighbours
//      a = disp(y-1, x), b = disp(y+1, x) (b = a below the last row), clamped to the pixel's own
//      candidate range [minDisparity, min(maxDisparity, x)]:  lo = min(a,b), hi = max(a,b).
//      lo == hi: the value is copied.  Otherwise the exact ASW cost (_passive.cpp:56-95) is evaluated
//      for d = lo..hi and the first minimum wins, as in the exact mode.  An empty candidate range
//      gives x, as in the exact mode (_passive.cpp:54,98).
// On Tsukuba about 10 % of all pixels need an evaluation, over about 4 candidates each; bad-1.0 against
// the ground truth does not get worse.
#pragma once
#include "asw_kernels.hip.h"

namespace ssamd {

static constexpr int ASW_ALT_JC = 8;        // candidates per job

struct AswAltArgs {
    const PixRec *recL, *recR;   // [H][W] pixel records
    const float *prox;           // [win*win]
    int16_t *disp;               // [rows][W], row 0 = image row row0: rows 0, 2, .. hold exact disparities, rows 1, 3, .. are written here
    u64 *key;                    // [rows][W] WTA keys; the odd rows arrive as KEY_NONE
    u64 *queue;                  // [cap] jobs: pixel index | first candidate << 32 | candidate count << 48
    unsigned int *ctr;           // [0] jobs appended
    unsigned int cap;
    int H, W, win, pad, minD, maxD;
    int row0, rows;              // output rows [row0, row0 + rows) of the (sub-)image; row0 is an exactly matched row
    float kC;                    // -log2(e)/gammaC
};

// support weight of one tap, the arithmetic of asw_aggregate_kernel's weight build
__device__ __forceinline__ float asw_alt_weight(const PixRec tap, const PixRec cen, float prox, float kC)
{
    const float dL = tap.L - cen.L, da = tap.a - cen.a, db = tap.b - cen.b;
    const float dist = __builtin_amdgcn_sqrtf(fmaf(db, db, fmaf(da, da, dL * dL)));
    return asw_weight_finish(dist, kC, prox);
}

__device__ __forceinline__ float asw_alt_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Candidates d0 .. d0+cnt-1 (cnt <= ASW_ALT_JC) of pixel (y, x) by one wave: the lanes split the window taps, the
// left tap and its weight are shared by the candidates, (N, S') of every candidate are reduced with a
// butterfly.  Returns the best (cost, d) key on every lane.
__device__ __forceinline__ u64 asw_alt_eval(const AswAltArgs &A, int y, int x, int d0, int cnt, int lane)
{
    const int W = A.W, H = A.H, win = A.win, p = A.pad;
    const PixRec cl = A.recL[(size_t)y * W + x];
    PixRec cr[ASW_ALT_JC];
    float n[ASW_ALT_JC], s[ASW_ALT_JC];
#pragma unroll
    for (int c = 0; c < ASW_ALT_JC; ++c) {
        cr[c] = A.recR[(size_t)y * W + (x - d0 - min(c, cnt - 1))];     // past the end: repeat the last candidate
        n[c] = 0.f; s[c] = 0.f;
    }
    const int step_i = 64 / win, step_j = 64 - step_i * win;             // tap index advances by 64 per iteration
    int i = lane / win, j = lane - i * win;
    for (int t = lane; t < win * win; t += 64) {
        const int yy = y - p + i, xl = x - p + j;
        if ((unsigned)yy < (unsigned)H && (unsigned)xl < (unsigned)W) {
            const PixRec tl = A.recL[(size_t)yy * W + xl];
            const float pr = A.prox[t];
            const float wl = asw_alt_weight(tl, cl, pr, A.kC);
            const PixRec *const rrow = A.recR + (size_t)yy * W;
#pragma unroll
            for (int c = 0; c < ASW_ALT_JC; ++c) {
                const int xrj = xl - d0 - min(c, cnt - 1);
                if ((unsigned)xrj < (unsigned)W) {
                    const PixRec tr = rrow[xrj];
                    const float w = wl * asw_alt_weight(tr, cr[c], pr, A.kC);
                    const float e = (float)min(__builtin_amdgcn_sad_u8(tl.bgrx, tr.bgrx, 0u), 40u);
                    n[c] = fmaf(w, e, n[c]);
                    s[c] = fmaf(w, ASW_TAD_CAP - e, s[c]);
                }
            }
        }
        i += step_i; j += step_j;
        if (j >= win) { j -= win; ++i; }
    }
    u64 best = KEY_NONE;
#pragma unroll
    for (int c = 0; c < ASW_ALT_JC; ++c) {
        const float nn = asw_alt_wave_sum(n[c]), sv = asw_alt_wave_sum(s[c]);
        float cost;
        const u64 k = ((u64)asw_cost_key(nn, sv, cost) << 32) | (u64)(uint32_t)(d0 + c);
        if (c < cnt) best = min(best, k);
    }
    return best;
}

// Step 1, one thread per pixel of the odd rows: candidate interval from the exact rows above and below;
// copied values and empty ranges are written at once, the others are cut into jobs of ASW_ALT_JC candidates
// and appended to a global queue (one atomic per wave), so that step 2 spreads the evaluations -- which
// cluster along depth edges -- over the whole GPU.  If the queue is full the wave evaluates its pixels itself.
__global__ __launch_bounds__(256) void asw_alt_scan_kernel(const AswAltArgs A)
{
    const int lane = threadIdx.x & 63;
    const int yr = 2 * blockIdx.y + 1, y = A.row0 + yr;             // row of the output range / of the image
    const int W = A.W;
    if (yr >= A.rows) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    int16_t *const drow = A.disp + (size_t)yr * W;
    const int16_t *const up = drow - W;
    const int16_t *const down = (yr + 1 < A.rows) ? drow + W : up;    // (the range ends with an exact row or with the image)
    int lo = 0, cnt = 0;
    if (x < W) {
        const int dmaxv = min(A.maxD, x);
        if (A.minD > dmaxv) {
            drow[x] = (int16_t)x;                                    // empty candidate loop: dBest = 0 -> x
        } else {
            const int a = up[x], b = down[x];
            lo = min(max(min(a, b), A.minD), dmaxv);
            const int hi = min(max(max(a, b), A.minD), dmaxv);
            if (lo == hi) drow[x] = (int16_t)lo;
            else cnt = hi - lo + 1;
        }
    }
    const int njobs = (cnt + ASW_ALT_JC - 1) / ASW_ALT_JC;
    int incl = njobs;                                                // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);          // wave-uniform (SGPR) from here on
    if (total == 0) return;
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(&A.ctr[0], (unsigned int)total);
    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
    if ((u64)base + (u64)total <= (u64)A.cap) {
        const u64 pix = (u64)((size_t)yr * W + x);                    // index into disp / key
        for (int k = 0; k < njobs; ++k)
            A.queue[base + incl - njobs + k] =
                pix | ((u64)(uint32_t)(lo + ASW_ALT_JC * k) << 32) | ((u64)(uint32_t)min(ASW_ALT_JC, cnt - ASW_ALT_JC * k) << 48);
        return;
    }
    // queue full: slots of this wave below the capacity become empty jobs, and the wave does the work in place
    for (unsigned int k = base + lane; k < A.cap && k < base + (unsigned int)total; k += 64) A.queue[k] = 0;
    u64 todo = __ballot(cnt > 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int xs = __builtin_amdgcn_readfirstlane(__shfl(x, src, 64)), los = __builtin_amdgcn_readfirstlane(__shfl(lo, src, 64)),
                  cs = __builtin_amdgcn_readfirstlane(__shfl(cnt, src, 64));
        u64 best = KEY_NONE;
        for (int d0 = los; d0 < los + cs; d0 += ASW_ALT_JC)
            best = min(best, asw_alt_eval(A, y, xs, d0, min(ASW_ALT_JC, los + cs - d0), lane));
        if (lane == 0) drow[xs] = (int16_t)(uint32_t)best;
    }
}

// Step 2: the jobs are dealt round-robin to the waves of the launch (every job is at most ASW_ALT_JC candidates
// over the same window, so a static deal balances; it also keeps the loop free of lane-divergent branches --
// a per-wave "lane 0 takes the next index" atomic inside the loop gets unswitched by the compiler and then
// breaks the cross-lane read of the index).  Results merge per pixel with atomicMin.
__global__ __launch_bounds__(256) void asw_alt_jobs_kernel(const AswAltArgs A)
{
    const int lane = threadIdx.x & 63;
    const unsigned int njobs = min(A.ctr[0], A.cap);
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6);
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    for (unsigned int jb = wave; jb < njobs; jb += nwaves) {        // wave-uniform (scalar) loop
        const u64 job = A.queue[jb];
        const uint32_t pix = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)job);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(job >> 32));
        const int cnt = (int)(hi >> 16), d0 = (int)(hi & 0xffffu);
        if (cnt > 0) {
            const int yr = (int)(pix / (uint32_t)A.W), x = (int)(pix - (uint32_t)yr * (uint32_t)A.W);
            const u64 best = asw_alt_eval(A, A.row0 + yr, x, d0, cnt, lane);
            if (lane == 0) atomicMin(&A.key[pix], best);
        }
    }
}

// Step 3: odd-row pixels that went through the queue take the disparity of their best key.
__global__ __launch_bounds__(256) void asw_alt_decode_kernel(const AswAltArgs A)
{
    const int yr = 2 * blockIdx.y + 1;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (yr >= A.rows || x >= A.W) return;
    const u64 k = A.key[(size_t)yr * A.W + x];
    if (k != KEY_NONE) A.disp[(size_t)yr * A.W + x] = (int16_t)(uint32_t)k;
}

}  // namespace ssamd
