#!/usr/bin/env python3
"""Golden vectors at the HEADLINE launch geometries, from the UNMODIFIED reference (oracle/_ref).

    make -C oracle ref && python tests/golden/make_golden_wide.py [--only-new]       (build container only; ~6 min on 8 cores)

tests/golden/cases.npz stops at 640x480 / D 0..64.  The bench line and BASELINE configs 3, 4 and 5 run 1920- and
4096-wide frames with 193 / 257 disparities, i.e. other workgroup tiles, chunkings and key paths of the kernels.
Whole frames are too slow for the reference here (config 3: ~15 min, config 5: ~1.5 h on 8 cores), so each case is
a FULL-WIDTH strip of rows cropped from the centre of the config's synthetic frame, matched as a stand-alone
image by the reference (`_passive.cpp` treats whatever it is given as the whole image) -- and by the GPU tests.

Output: wide_cases.npz (int16 maps) + wide_cases.json (recipe, parameters, sha256).  Inputs are regenerated from
the recipe (simplestereo_amd.synth.make_pair is deterministic), not stored.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as _oracle            # noqa: E402
from simplestereo_amd.synth import make_pair     # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# id: (frame H, W, maxDisparity of the frame generator, seed, first row, rows, params)
A = lambda **k: dict(algo="asw", **k)   # noqa: E731
G = lambda **k: dict(algo="gsw", **k)   # noqa: E731
CASES = {
    # config 3 (bench line): 1920 wide, D 0..192, win 35
    "W3a": (1080, 1920, 192, 0, 504, 72, A(winSize=35, maxDisparity=192, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
    "W3b": (1080, 1920, 192, 1, 504, 72, A(winSize=35, maxDisparity=192, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
    # class-default range on the same frame (StereoASW(): win 35, D 0..16)
    "W3c": (1080, 1920, 192, 0, 520, 40, A(winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
    # config 5: 4096 wide, D 0..256, win 35
    "W5a": (2160, 4096, 256, 0, 1056, 48, A(winSize=35, maxDisparity=256, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
    # config 4: GSW class defaults, 1920 wide, D 0..192 (left-right check always on)
    "W4a": (1080, 1920, 192, 0, 520, 40, G(winSize=11, maxDisparity=192, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
    "W4b": (1080, 1920, 192, 1, 520, 40, G(winSize=11, maxDisparity=192, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
    # small disparity ranges on frames whose true disparities lie in the range (the wave kernel's 4- and 8-column tiles):
    # D 0..7, the class default D 0..16 with the left-right check, D 3..40 with a 21 x 21 window
    "W3d": (1080, 1920, 7, 2, 520, 40, A(winSize=35, maxDisparity=7, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
    "W3e": (1080, 1920, 16, 4, 520, 40, A(winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
    "W3f": (1080, 1920, 40, 3, 520, 44, A(winSize=21, maxDisparity=40, minDisparity=3, gammaC=5, gammaP=17.5, consistent=True)),
}


def strip_inputs(H, W, maxD, seed, r0, rows):
    L, R, _ = make_pair(H, W, maxD, seed)
    return np.ascontiguousarray(L[r0:r0 + rows]), np.ascontiguousarray(R[r0:r0 + rows])


def main():
    ref = _oracle.ref_module()
    if ref is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` first")
    maps, meta = {}, {}
    only_new = "--only-new" in sys.argv          # keep the cases already in wide_cases.npz, compute the ones added since
    if only_new:
        old = np.load(os.path.join(OUT, "wide_cases.npz"))
        maps = {k: old[k] for k in old.files}
        meta = json.load(open(os.path.join(OUT, "wide_cases.json")))
    for cid, (H, W, maxD, seed, r0, rows, p) in CASES.items():
        if only_new and cid in maps:
            continue
        a, b = strip_inputs(H, W, maxD, seed, r0, rows)
        t = time.time()
        if p["algo"] == "asw":
            d = ref.computeASW(a, b, p["winSize"], p["maxDisparity"], p["minDisparity"],
                               float(p["gammaC"]), float(p["gammaP"]), bool(p["consistent"]))
        else:
            d = ref.computeGSW(a, b, p["winSize"], p["maxDisparity"], p["minDisparity"],
                               p["gamma"], float(p["fMax"]), p["iterations"], p["bins"])
        dt = time.time() - t
        assert d.dtype == np.int16 and d.shape == a.shape[:2]
        maps[cid] = d
        meta[cid] = dict(recipe="simplestereo_amd.synth.make_pair(%d,%d,%d,seed=%d) rows [%d:%d]" % (H, W, maxD, seed, r0, r0 + rows),
                         frame=[H, W, maxD, seed], row0=r0, rows=rows, params=p, shape=list(d.shape),
                         sha256=hashlib.sha256(d.tobytes()).hexdigest(), checksum=int(d.astype(np.int64).sum()),
                         input_sha256=hashlib.sha256(a.tobytes() + b.tobytes()).hexdigest(), ref_seconds=round(dt, 2))
        print("%-4s %s sum=%d  %.1fs" % (cid, meta[cid]["sha256"][:16], meta[cid]["checksum"], dt), flush=True)
    np.savez_compressed(os.path.join(OUT, "wide_cases.npz"), **maps)
    with open(os.path.join(OUT, "wide_cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
