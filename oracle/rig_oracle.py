"""oracle/rig_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent CPU restatements (plain per-pixel Python/numpy loops, deliberately NOT sharing code
with simplestereo_amd/_rigs.py) of the two OpenCV calls around the matching path that the
reference makes and that this repo runs as HIP kernels:

  remap_bilinear  -- cv2.remap(img, mapx, mapy, INTER_LINEAR / INTER_NEAREST, BORDER_CONSTANT 0),
                     called by RectifiedStereoRig.rectifyImages (reference _rigs.py:564-565)
  reproject       -- cv2.reprojectImageTo3D(disparity, Q) called by get3DPoints (_rigs.py:628)

PARITY UNPINNED: OpenCV is not installed in this environment, so these restatements of the
published OpenCV semantics (fixed-point bilinear with 5 fractional bits; homogeneous divide)
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
            fx, fy = (qx & 31) / 32.0, (qy & 31) / 32.0
            acc = np.zeros(img.shape[2], np.float64)
            for dy, wy in ((0, 1.0 - fy), (1, fy)):
                for dx, wx in ((0, 1.0 - fx), (1, fx)):
                    xx, yy = x0 + dx, y0 + dy
                    if 0 <= xx < Ws and 0 <= yy < Hs:
                        acc += wy * wx * img[yy, xx].astype(np.float64)
            out[y, x] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
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
