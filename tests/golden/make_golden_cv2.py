#!/usr/bin/env python3
"""Golden vectors for the OpenCV-backed steps either side of the matchers, to be run WHEREVER
cv2 IS INSTALLED (it is absent from the build container and from the GPU boxes, so `cv2_pins.npz` may not exist):

    python tests/golden/make_golden_cv2.py          # writes tests/golden/cv2_pins.npz

What the reference calls (paths under the reference repo):
  * cv2.initUndistortRectifyMap(K, dist, R, newK, size, cv2.CV_32FC1)      simplestereo/_rigs.py:540-541
  * cv2.remap(img, mapx, mapy, interpolation)                              simplestereo/_rigs.py:564-565
  * cv2.reprojectImageTo3D(disparity, Q)                                   simplestereo/_rigs.py:628
Inputs: the reference's own example rig (tests/golden/rig_example2_rigRect.json = examples/res/2/rigRect.json), the
new camera matrices our own `computeRectificationMaps` derives for it (pure linear algebra, reference _rigs.py:
491-539), a seeded synthetic image and a seeded int16 disparity map.  tests/test_cv2_pins.py compares
simplestereo_amd._rigs with cv2 directly when cv2 can be imported, and with this file when it exists."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def inputs():
    import simplestereo_amd as ss
    rig = ss.RectifiedStereoRig.fromFile(os.path.join(HERE, "rig_example2_rigRect.json"))
    rig.computeRectificationMaps()
    w, h = rig.res1
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (np.stack([xx % 251, (xx + 2 * yy) % 241, (3 * yy) % 239], -1) + rng.integers(0, 8, (h, w, 3))).astype(np.uint8)
    img16 = (img[:, :, 0].astype(np.uint16) * 257)
    disp = rng.integers(1, 200, (h, w)).astype(np.int16)
    return rig, img, img16, disp


def with_cv2():
    import cv2
    rig, img, img16, disp = inputs()
    out = {}
    for k, (K, dist, R, newK) in {"1": (rig.intrinsic1, rig.distCoeffs1, rig.Rcommon, rig.K1),
                                  "2": (rig.intrinsic2, rig.distCoeffs2, rig.Rcommon.dot(rig.R.T), rig.K2)}.items():
        mx, my = cv2.initUndistortRectifyMap(K, dist, R, newK, tuple(rig.res1), cv2.CV_32FC1)
        out["mapx" + k], out["mapy" + k] = mx, my
        out["remap_linear" + k] = cv2.remap(img, mx, my, cv2.INTER_LINEAR)
        out["remap_nearest" + k] = cv2.remap(img, mx, my, cv2.INTER_NEAREST)
        out["remap_linear_u16_" + k] = cv2.remap(img16, mx, my, cv2.INTER_LINEAR)
    out["points"] = cv2.reprojectImageTo3D(disp, rig.getQ())
    return out


if __name__ == "__main__":
    res = with_cv2()
    np.savez_compressed(os.path.join(HERE, "cv2_pins.npz"), **res)
    print("wrote cv2_pins.npz:", {k: v.shape for k, v in res.items()})
