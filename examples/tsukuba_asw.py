#!/usr/bin/env python3
"""The reference's `examples/010 StereoMatchingTsukuba.py` on the MI355X back-end.

Runs ASW with the parameters of the reference example on the Middlebury Tsukuba pair
(stored as arrays in tests/golden/tsukuba_pair.npz; OpenCV is not needed), prints the
bad-1.0 error against the ground truth shipped with the reference, and writes the
disparity map and an adimensional point cloud.

    python examples/tsukuba_asw.py [--out /tmp/tsukuba]
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
    ap.add_argument("--out", default=None, help="directory for disparity.npy / cloud.ply")
    ap.add_argument("--alternate", action="store_true",
                    help="opt-in faster mode: every other row exact, rows in between by bounded search (not the reference's output)")
    args = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "tsukuba_pair.npz"))
    imgL, imgR = np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])

    # same call as the reference example (examples/010:30-31)
    stereo = ss.passive.StereoASW(winSize=35, minDisparity=4, maxDisparity=14, gammaC=15, gammaP=17.5, consistent=True,
                                  alternate=args.alternate)
    stereo.compute(imgL, imgR)                       # first call allocates device scratch
    t = time.perf_counter()
    disparityMap = stereo.compute(imgL, imgR)
    dt = time.perf_counter() - t

    gt = z["groundtruth"].astype(np.float64) / 16.0
    mask = (z["nonocc"] > 0) & (z["groundtruth"] > 0)
    bad1 = 100.0 * np.mean(np.abs(disparityMap - gt)[mask] > 1.0)
    print("ASW %dx%d, D 4..14, win 35, consistent%s: %.2f ms (host buffers), bad-1.0 = %.2f %% "
          "(reference C++: 6.04 s on 8 threads, 2.11 %%)" % (imgL.shape[1], imgL.shape[0], ", alternate rows" if args.alternate else "",
                                                             dt * 1e3, bad1))
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        np.save(os.path.join(args.out, "disparity.npy"), disparityMap)
        ss.points.exportPLY(ss.points.getAdimensional3DPoints(disparityMap), os.path.join(args.out, "cloud.ply"), imgL)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
