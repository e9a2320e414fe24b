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

namespace ssamd {

// OpenCV's published 8-bit bilinear arithmetic (modules/imgproc/src/imgwarp.cpp, remap / remapBilinear and
// initInterTab2D): map coordinates are rounded to 1/32 pixel (INTER_BITS = 5, cvRound = half to even), the four
// weights are 15-bit integers (INTER_REMAP_COEF_BITS = 15; with 5-bit fractions a*b*32 exactly, summing to 32768)
// and the result is FixedPtCast: (sum + (1 << 14)) >> 15, i.e. ties round UP -- here with the common factor 32
// divided out: (S + 512) >> 10, S = sum a*b*pixel, a, b in 0..32.  Source pixels outside the image contribute the
// constant border value 0.
__global__ __launch_bounds__(256) void remap_bgr_kernel(const uint8_t *__restrict__ src, int Hs, int Ws,
                                                        const float *__restrict__ mapx, const float *__restrict__ mapy,
                                                        uint8_t *__restrict__ dst, long long npix, int nearest)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; p < npix; p += stride) {
        float out[3] = {0.f, 0.f, 0.f};
        if (nearest) {
            const int xi = (int)rintf(mapx[p]), yi = (int)rintf(mapy[p]);
            if ((unsigned)xi < (unsigned)Ws && (unsigned)yi < (unsigned)Hs) {
                const uint8_t *s = src + ((size_t)yi * Ws + xi) * 3;
                out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
            }
        } else {
            const long long qx = llrint((double)mapx[p] * 32.0), qy = llrint((double)mapy[p] * 32.0);
            const long long x0 = qx >> 5, y0 = qy >> 5;
            const int fx = (int)(qx & 31), fy = (int)(qy & 31);
            int acc[3] = {0, 0, 0};
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
            out[0] = (float)((acc[0] + 512) >> 10); out[1] = (float)((acc[1] + 512) >> 10); out[2] = (float)((acc[2] + 512) >> 10);
        }
        dst[3 * p] = (uint8_t)fminf(fmaxf(out[0], 0.f), 255.f);
        dst[3 * p + 1] = (uint8_t)fminf(fmaxf(out[1], 0.f), 255.f);
        dst[3 * p + 2] = (uint8_t)fminf(fmaxf(out[2], 0.f), 255.f);
    }
}

struct Mat4 {
    double m[16];
};

// [X Y Z W]^T = Q [x y d 1]^T ; point = (X/W, Y/W, Z/W) as float32
__global__ __launch_bounds__(256) void reproject_kernel(const int16_t *__restrict__ disp, float *__restrict__ pts,
                                                        int H, int W, const Mat4 Q)
{
    const long long npix = (long long)H * W;
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; p < npix; p += stride) {
        const double x = (double)(p % W), y = (double)(p / W), d = (double)disp[p];
        const double X = Q.m[0] * x + Q.m[1] * y + Q.m[2] * d + Q.m[3];
        const double Y = Q.m[4] * x + Q.m[5] * y + Q.m[6] * d + Q.m[7];
        const double Z = Q.m[8] * x + Q.m[9] * y + Q.m[10] * d + Q.m[11];
        const double Wc = Q.m[12] * x + Q.m[13] * y + Q.m[14] * d + Q.m[15];
        pts[3 * p] = (float)(X / Wc);
        pts[3 * p + 1] = (float)(Y / Wc);
        pts[3 * p + 2] = (float)(Z / Wc);
    }
}

}  // namespace ssamd
