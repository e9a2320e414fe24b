// K5/K6: the two pointwise kernels either side of the matching path, so that
// a frame can go camera image -> rectified pair -> disparity -> 3-D points without leaving HBM.
//   K5 remap_bgr_kernel   : RectifiedStereoRig.rectifyImages = cv2.remap(img, mapx, mapy, INTER_LINEAR,
//                           BORDER_CONSTANT 0)  (reference _rigs.py:543-567)
//   K6 reproject_kernel   : RectifiedStereoRig.get3DPoints = cv2.reprojectImageTo3D(disparity, Q)
//                           (reference _rigs.py:569-628)
// Both are HBM-bound: K5 reads 8 B of map + a 2x2x3-byte neighbourhood (L2-friendly: neighbouring
// output pixels read neighbouring source pixels) and writes 3 B per pixel; K6 reads 2 B and writes 12 B.
#pragma once
#include "common.hip.h"
#include "lab_kernels.hip.h"

namespace ssamd {

// OpenCV's published 8-bit bilinear arithmetic (modules/imgproc/src/imgwarp.cpp, remap / remapBilinear and
// initInterTab2D): map coordinates are rounded to 1/32 pixel (INTER_BITS = 5, cvRound = half to even), the four
// weights are 15-bit integers (INTER_REMAP_COEF_BITS = 15; with 5-bit fractions a*b*32 exactly, summing to 32768)
// and the result is FixedPtCast: (sum + (1 << 14)) >> 15, i.e. ties round UP -- here with the common factor 32
// divided out: (S + 512) >> 10, S = sum a*b*pixel, a, b in 0..32.  Source pixels outside the image contribute the
// constant border value 0.
// One output pixel: the three channel values 0..255.  The four taps of the common case (all inside the image) come from
// two unaligned 8-byte loads -- the 6 bytes of two horizontally adjacent BGR pixels each -- instead of twelve byte loads.
__device__ __forceinline__ uint32_t remap_pixel(const uint8_t *__restrict__ src, int Hs, int Ws, size_t src_bytes, float mx, float my,
                                                int nearest)
{
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    if (nearest) {
        const int xi = (int)rintf(mx), yi = (int)rintf(my);
        if ((unsigned)xi < (unsigned)Ws && (unsigned)yi < (unsigned)Hs) {
            const uint8_t *s = src + ((size_t)yi * Ws + xi) * 3;
            return (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        return 0u;
    }
    const long long qx = llrint((double)mx * 32.0), qy = llrint((double)my * 32.0);
    const long long x0 = qx >> 5, y0 = qy >> 5;
    const int fx = (int)(qx & 31), fy = (int)(qy & 31);
    int acc[3] = {0, 0, 0};
    const size_t off0 = ((size_t)y0 * Ws + (size_t)x0) * 3, off1 = off0 + (size_t)Ws * 3;
    if (x0 >= 0 && x0 + 1 < Ws && y0 >= 0 && y0 + 1 < Hs && off1 + 8 <= src_bytes) {
        const uint64_t r0 = *reinterpret_cast<const u64_unaligned *>(src + off0), r1 = *reinterpret_cast<const u64_unaligned *>(src + off1);
        const int w00 = (32 - fy) * (32 - fx), w01 = (32 - fy) * fx, w10 = fy * (32 - fx), w11 = fy * fx;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            acc[ch] = w00 * (int)((r0 >> (8 * ch)) & 0xff) + w01 * (int)((r0 >> (8 * ch + 24)) & 0xff) +
                      w10 * (int)((r1 >> (8 * ch)) & 0xff) + w11 * (int)((r1 >> (8 * ch + 24)) & 0xff);
    } else {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long long xx = x0 + dx, yy = y0 + dy;
                if (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs) {
                    const int w = (dy ? fy : 32 - fy) * (dx ? fx : 32 - fx);
                    const uint8_t *s = src + ((size_t)yy * Ws + xx) * 3;
                    acc[0] += w * s[0]; acc[1] += w * s[1]; acc[2] += w * s[2];
                }
            }
    }
    // (S + 512) >> 10 of sums of at most 1024 * 255 is already inside 0..255
    return (uint32_t)((acc[0] + 512) >> 10) | ((uint32_t)((acc[1] + 512) >> 10) << 8) | ((uint32_t)((acc[2] + 512) >> 10) << 16);
}

// A thread owns FOUR consecutive output pixels: two 16-byte map reads, three 4-byte stores of the 12 output bytes
// (both fully coalesced across the wave); the pixels left over when npix is not a multiple of 4 go one per thread.
__global__ __launch_bounds__(256) void remap_bgr_kernel(const uint8_t *__restrict__ src, int Hs, int Ws,
                                                        const float *__restrict__ mapx, const float *__restrict__ mapy,
                                                        uint8_t *__restrict__ dst, long long npix, int nearest)
{
    const size_t src_bytes = (size_t)Hs * Ws * 3;
    const long long nquad = npix >> 2;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (long long q = tid; q < nquad; q += stride) {
        const float4 mx = reinterpret_cast<const float4 *>(mapx)[q], my = reinterpret_cast<const float4 *>(mapy)[q];
        const uint32_t p0 = remap_pixel(src, Hs, Ws, src_bytes, mx.x, my.x, nearest), p1 = remap_pixel(src, Hs, Ws, src_bytes, mx.y, my.y, nearest);
        const uint32_t p2 = remap_pixel(src, Hs, Ws, src_bytes, mx.z, my.z, nearest), p3 = remap_pixel(src, Hs, Ws, src_bytes, mx.w, my.w, nearest);
        uint32_t *const o = reinterpret_cast<uint32_t *>(dst) + 3 * q;
        o[0] = p0 | (p1 << 24);
        o[1] = (p1 >> 8) | (p2 << 16);
        o[2] = (p2 >> 16) | (p3 << 8);
    }
    for (long long p = 4 * nquad + tid; p < npix; p += stride) {
        const uint32_t v = remap_pixel(src, Hs, Ws, src_bytes, mapx[p], mapy[p], nearest);
        dst[3 * p] = (uint8_t)v; dst[3 * p + 1] = (uint8_t)(v >> 8); dst[3 * p + 2] = (uint8_t)(v >> 16);
    }
}

// K5+K0 (round 4): rectification remap and Lab records of BOTH images in one launch -- RectifiedStereoRig.rectifyImages
// (reference _rigs.py:543-567) feeding StereoASW.compute (passive.py:88), the reference's own pipeline (examples/009:20-39),
// without the rectified BGR frames ever existing in HBM: the remapped pixel goes straight into bgr_to_lab and the
// 16-byte record.  Same arithmetic as remap_bgr_kernel + bgr2lab_records_pair_kernel, so the records (and the maps
// matched from them) are bit-identical to the two-step path.  Rows [row_lo, row_hi) of the destination only (a matcher
// call touches its output rows +- winSize / 2).
struct RemapSrc {
    const uint8_t *src1, *src2;      // raw frames [Hs][Ws][3]
    const float *mapx1, *mapy1, *mapx2, *mapy2;      // [H][W] float32 maps of the rig
    int Hs, Ws, nearest;
};
__global__ __launch_bounds__(256) void remap_lab_records_pair_kernel(const RemapSrc S, PixRec *__restrict__ recL, PixRec *__restrict__ recR,
                                                                     long long pix0, long long npix)
{
    SSAMD_LAB_TABLES_IN_LDS(T)
    const size_t src_bytes = (size_t)S.Hs * S.Ws * 3;
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; q < 2 * npix; q += stride) {
        const bool right = q >= npix;
        const long long p = pix0 + (right ? q - npix : q);
        const uint32_t v = remap_pixel(right ? S.src2 : S.src1, S.Hs, S.Ws, src_bytes, (right ? S.mapx2 : S.mapx1)[p],
                                       (right ? S.mapy2 : S.mapy1)[p], S.nearest);
        PixRec o;
        bgr_to_lab(v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff, o.L, o.a, o.b, T);
        o.bgrx = v;
        (right ? recR : recL)[p] = o;
    }
}

// The same for GSW, which works on the raw colour bytes (_passive.cpp:740-741): remap + pack (B | G << 8 | R << 16) of both images.
__global__ __launch_bounds__(256) void remap_pack_pair_kernel(const RemapSrc S, uint32_t *__restrict__ outL, uint32_t *__restrict__ outR,
                                                              long long pix0, long long npix)
{
    const size_t src_bytes = (size_t)S.Hs * S.Ws * 3;
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; q < 2 * npix; q += stride) {
        const bool right = q >= npix;
        const long long p = pix0 + (right ? q - npix : q);
        (right ? outR : outL)[p] = remap_pixel(right ? S.src2 : S.src1, S.Hs, S.Ws, src_bytes, (right ? S.mapx2 : S.mapx1)[p],
                                              (right ? S.mapy2 : S.mapy1)[p], S.nearest);
    }
}

struct Mat4 {
    double m[16];
};

// [X Y Z W]^T = Q [x y d 1]^T ; point = (X/W, Y/W, Z/W) as float32.  Grid row = image row: no 64-bit division / modulo
// per pixel.  With W a multiple of 4 a thread owns four pixels: one 8-byte disparity read, three 16-byte stores.
__device__ __forceinline__ void reproject_pixel(const Mat4 &Q, double x, double y, double d, float &ox, float &oy, float &oz)
{
    const double X = Q.m[0] * x + Q.m[1] * y + Q.m[2] * d + Q.m[3];
    const double Y = Q.m[4] * x + Q.m[5] * y + Q.m[6] * d + Q.m[7];
    const double Z = Q.m[8] * x + Q.m[9] * y + Q.m[10] * d + Q.m[11];
    const double Wc = Q.m[12] * x + Q.m[13] * y + Q.m[14] * d + Q.m[15];
    // three quotients by the same (finite) divisor: one reciprocal, then q = X r corrected by one fma pair on the exact
    // remainder -- the correctly rounded quotient in all but double-rounding corner cases, which the float cast below
    // hides; W = 0 keeps q = +-inf / nan, exactly what the plain divisions give
    const double r = 1.0 / Wc;
    auto quot = [&](double n) {
        const double q = n * r;
        return (float)(__builtin_isfinite(q) ? fma(fma(-q, Wc, n), r, q) : q);
    };
    ox = quot(X);
    oy = quot(Y);
    oz = quot(Z);
}

__global__ __launch_bounds__(256) void reproject_kernel(const int16_t *__restrict__ disp, float *__restrict__ pts,
                                                        int H, int W, const Mat4 Q)
{
    const int y = blockIdx.y;
    const double yd = (double)y;
    const int16_t *const drow = disp + (size_t)y * W;
    float *const prow = pts + (size_t)y * W * 3;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    __shared__ float4 xchg[4][192];
    const int lane = threadIdx.x & 63;
    if ((W & 3) == 0) {
        // whole waves iterate (q0 = the wave's first quad): lanes past the row's last quad compute nothing but help write
        for (int q0 = t - lane; 4 * q0 < W; q0 += nt) {
            const int q = q0 + lane;
            float4 *const mine = xchg[threadIdx.x >> 6] + 3 * lane;
            if (4 * q < W) {
                const short4 dv = reinterpret_cast<const short4 *>(drow)[q];
                float o[12];
                reproject_pixel(Q, (double)(4 * q), yd, (double)dv.x, o[0], o[1], o[2]);
                reproject_pixel(Q, (double)(4 * q + 1), yd, (double)dv.y, o[3], o[4], o[5]);
                reproject_pixel(Q, (double)(4 * q + 2), yd, (double)dv.z, o[6], o[7], o[8]);
                reproject_pixel(Q, (double)(4 * q + 3), yd, (double)dv.w, o[9], o[10], o[11]);
                mine[0] = make_float4(o[0], o[1], o[2], o[3]);
                mine[1] = make_float4(o[4], o[5], o[6], o[7]);
                mine[2] = make_float4(o[8], o[9], o[10], o[11]);
            }
            // the wave's up to 64 x 48 bytes go out as three fully contiguous 1 KiB stores: the 16-byte chunks are
            // transposed through LDS (a wave's LDS accesses execute in order: no barrier), lane l writes chunks l, l + 64, l + 128
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int nchunk = 3 * min(64, W / 4 - q0);
            float4 *const op = reinterpret_cast<float4 *>(prow) + 3 * q0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (lane + 64 * k < nchunk) op[lane + 64 * k] = xchg[threadIdx.x >> 6][lane + 64 * k];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    for (int x = t; x < W; x += nt) {
        float ox, oy, oz;
        reproject_pixel(Q, (double)x, yd, (double)drow[x], ox, oy, oz);
        prow[3 * x] = ox; prow[3 * x + 1] = oy; prow[3 * x + 2] = oz;
    }
}

}  // namespace ssamd
