"""oracle/rig_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent CPU restatements (plain per-pixel Python/numpy loops, deliberately NOT sharing code
with simplestereo_amd/_rigs.py) of the two OpenCV calls around the matching path that the
reference makes and that this repo runs as HIP kernels:

  remap_bilinear  -- cv2.remap(img, mapx, mapy, INTER_LINEAR / INTER_NEAREST, BORDER_CONSTANT 0),
                     called by RectifiedStereoRig.rectifyImages (reference _rigs.py:564-565)
  reproject       -- cv2.reprojectImageTo3D(disparity, Q) called by get3DPoints (_rigs.py:628)

PARITY UNPINNED: OpenCV is not installed in this environment, so these restatements of the
published OpenCV semantics (imgwarp.cpp: map coordinates cvRound-ed to 1/32 pixel, 15-bit integer bilinear
weights summing to 32768, FixedPtCast (sum + 16384) >> 15 so that ties round up; homogeneous divide)
cannot be checked against cv2 itself.  They pin the HIP kernels and the numpy host path against
each other only.  Only tests/ may import this module.
"""
import math

import numpy as np


def remap_bilinear(img, mapx, mapy, nearest=False):
    img = np.asarray(img)
    Hs, Ws = img.shape[:2]
    H, W = mapx.shape
    out = np.zeros((H, W, img.shape[2]), np.uint8)
    for y in range(H):
        for x in range(W):
            mx, my = float(mapx[y, x]), float(mapy[y, x])
            if nearest:
                xi, yi = int(np.rint(mx)), int(np.rint(my))
                if 0 <= xi < Ws and 0 <= yi < Hs:
                    out[y, x] = img[yi, xi]
                continue
            qx, qy = int(np.rint(mx * 32.0)), int(np.rint(my * 32.0))
            x0, y0 = qx >> 5, qy >> 5                      # floor division, also for negatives
            fx, fy = qx & 31, qy & 31
            # OpenCV's BilinearTab_i entry for (fy, fx): shorts, round(weight * 32768), here exact
            tab = [[(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32], [fy * (32 - fx) * 32, fy * fx * 32]]
            for ch in range(img.shape[2]):
                total = 0
                for dy in (0, 1):
                    for dx in (0, 1):
                        xx, yy = x0 + dx, y0 + dy
                        if 0 <= xx < Ws and 0 <= yy < Hs:
                            total += tab[dy][dx] * int(img[yy, xx, ch])
                out[y, x, ch] = min(255, max(0, (total + (1 << 14)) >> 15))
    return out


def reproject(disparity, Q):
    d = np.asarray(disparity)
    H, W = d.shape
    Q = np.asarray(Q, dtype=np.float64)
    out = np.empty((H, W, 3), np.float32)
    for y in range(H):
        for x in range(W):
            v = Q.dot(np.array([x, y, float(d[y, x]), 1.0]))
            w = v[3]
            out[y, x] = [v[0] / w if w != 0 else math.copysign(math.inf, v[0]) if v[0] != 0 else math.nan,
                         v[1] / w if w != 0 else math.copysign(math.inf, v[1]) if v[1] != 0 else math.nan,
                         v[2] / w if w != 0 else math.copysign(math.inf, v[2]) if v[2] != 0 else math.nan]
    return out
