#!/usr/bin/env python3
"""The reference's `examples/009 StereoMatchingASW.py` (rig -> rectifyImages -> quarter size -> StereoASW) on the MI355X back-end.

The reference ships the pair `examples/res/2/lawn_{L,R}.png` with its rig; this repository stores the matcher INPUT of that
example -- the rectified pair at quarter size, 320 x 180 -- as arrays (tests/golden/photo_pairs.npz, made by
tests/golden/make_golden_photo.py; OpenCV is not needed) together with the map the unmodified reference computes from it
(tests/golden/photo_cases.npz, case P1).  This script runs the example's call and compares.

    python examples/lawn_asw.py [--out /tmp/lawn]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simplestereo_amd as ss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None, help="directory for disparity.npy")
    args = ap.parse_args()
    g = os.path.join(ROOT, "tests", "golden")
    pairs = np.load(os.path.join(g, "photo_pairs.npz"))
    img1_rect, img2_rect = np.ascontiguousarray(pairs["lawn_quarter_L"]), np.ascontiguousarray(pairs["lawn_quarter_R"])

    # same call as the reference example (examples/009:34)
    stereo = ss.passive.StereoASW(winSize=35, minDisparity=4, maxDisparity=25, gammaC=15, gammaP=17.5, consistent=False)
    stereo.compute(img1_rect, img2_rect)             # first call allocates device scratch
    t = time.perf_counter()
    disparityMap = stereo.compute(img1_rect, img2_rect)
    dt = time.perf_counter() - t

    ref = np.load(os.path.join(g, "photo_cases.npz"))["P1"]
    same = float(np.mean(disparityMap == ref))
    print("ASW %dx%d, D 4..25, win 35: %.3f ms per call; %.3f %% of the pixels identical to the reference's map "
          "(the reference needs ~3.5 s on 8 threads for this frame)" % (img1_rect.shape[1], img1_rect.shape[0], dt * 1e3, 100 * same))
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        np.save(os.path.join(args.out, "disparity.npy"), disparityMap)


if __name__ == "__main__":
    main()
