#!/usr/bin/env python3
"""Golden vectors on REAL PHOTOGRAPHS, from the UNMODIFIED reference (oracle/_ref).

    make -C oracle ref && python tests/golden/make_golden_photo.py        (build container only; ~2 min on 8 cores)

Every other reference-made golden is the Tsukuba pair (384 x 288) or the repository's synthetic generator.  Smooth natural
content at HD width -- lawn, brick, defocus, the black margins rectification leaves -- is where the fp32 kernels' argmin could part
from the reference's fp64 one, so this file pins the reference's OWN ASW example pipeline (reference examples/009
StereoMatchingASW.py:20-39: rig -> rectifyImages -> quarter size -> StereoASW(winSize=35, minDisparity=4, maxDisparity=25,
gammaC=15)) on the pair it ships (examples/res/2/lawn_{L,R}.png, 1280 x 720), full-width strips of the native-size rectified
pair, and a strip of one of the photographic pairs of examples/res/new (unrectified captures of the same rig: the matchers treat
whatever they get as a rectified pair).

cv2 is absent here: images are read with PIL (same decoded bytes as cv2.imread, channel order flipped to BGR), rectified by
simplestereo_amd's own numpy `rectifyImages`, and reduced 4x by the 2 x 2 mean of the two centre pixels of every 4 x 4 block
(what cv2.resize(fx=0.25, INTER_LINEAR) samples).  Those steps only PRODUCE the matcher inputs; the inputs themselves are stored
(photo_pairs.npz, like tsukuba_pair.npz: reference data as arrays), so what is pinned is matcher(input) -> map.

Output: photo_pairs.npz (uint8 inputs), photo_cases.npz (int16 maps), photo_cases.json (parameters, sha256, checksums).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as _oracle            # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
RES = "/root/reference/examples/res"

A = lambda **k: dict(algo="asw", **k)   # noqa: E731
G = lambda **k: dict(algo="gsw", **k)   # noqa: E731


def imread_bgr(path):
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def quarter(img):
    """cv2.resize(img, None, fx=0.25, fy=0.25) with the default INTER_LINEAR: destination pixel (y, x) samples the source at
    (4y + 1.5, 4x + 1.5), i.e. the mean of the four pixels (4y+1..4y+2, 4x+1..4x+2), rounded half up."""
    H, W = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    s = img[:H, :W].astype(np.uint32)
    q = s[1::4, 1::4] + s[1::4, 2::4] + s[2::4, 1::4] + s[2::4, 2::4]
    return np.ascontiguousarray(((q + 2) >> 2).astype(np.uint8))


def half(img):
    """cv2.resize(img, None, fx=0.5, fy=0.5) with INTER_LINEAR: the mean of every 2 x 2 block, rounded half up"""
    H, W = img.shape[0] // 2 * 2, img.shape[1] // 2 * 2
    s = img[:H, :W].astype(np.uint32)
    return np.ascontiguousarray(((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))


def inputs():
    import simplestereo_amd as ss
    rig = ss.RectifiedStereoRig.fromFile(os.path.join(RES, "2", "rigRect.json"))
    l, r = rig.rectifyImages(imread_bgr(os.path.join(RES, "2", "lawn_L.png")), imread_bgr(os.path.join(RES, "2", "lawn_R.png")))
    n3l, n3r = imread_bgr(os.path.join(RES, "new", "3_L.png")), imread_bgr(os.path.join(RES, "new", "3_R.png"))
    c = np.ascontiguousarray
    return {
        "lawn_quarter_L": quarter(l), "lawn_quarter_R": quarter(r),                # 180 x 320: the example's matcher input
        "lawn_brick_L": c(l[130:194]), "lawn_brick_R": c(r[130:194]),              # 64 x 1280: chair, brick wall, bicycle
        "lawn_grass_L": c(l[600:664]), "lawn_grass_R": c(r[600:664]),              # 64 x 1280: grass + the black margin below
        "new3_L": c(n3l[300:348]), "new3_R": c(n3r[300:348]),                      # 48 x 1280: unrectified capture
        "lawn_half_L": half(l), "lawn_half_R": half(r),                            # 360 x 640: the WHOLE rectified frame at half size
    }


EX009 = dict(winSize=35, maxDisparity=25, minDisparity=4, gammaC=15, gammaP=17.5)
CASES = {
    # reference examples/009:34, verbatim parameters, on the quarter-size pair
    "P1": ("lawn_quarter", A(consistent=False, **EX009)),
    "P1c": ("lawn_quarter", A(consistent=True, **EX009)),
    # native size, full width, D 4..100 (the quarter-size range x 4)
    "P2a": ("lawn_brick", A(winSize=35, maxDisparity=100, minDisparity=4, gammaC=15, gammaP=17.5, consistent=False)),
    "P2b": ("lawn_grass", A(winSize=35, maxDisparity=100, minDisparity=4, gammaC=15, gammaP=17.5, consistent=True)),
    # class-default gammas + the left-right check, and GSW (class defaults but the range) on an unrectified photograph
    "P3a": ("new3", A(winSize=35, maxDisparity=64, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
    "P3b": ("new3", G(winSize=11, maxDisparity=64, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
    # the class default StereoASW() / StereoGSW() on the quarter-size pair
    "P4a": ("lawn_quarter", A(winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
    "P4b": ("lawn_quarter", G(winSize=11, maxDisparity=16, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
    # the WHOLE frame -- sky, brick, chair, bicycle, mower, lawn and the black margins of the rectification -- at half size with
    # the example's window and twice its range, left-right check on; and GSW on the same frame
    "P5a": ("lawn_half", A(winSize=35, maxDisparity=50, minDisparity=8, gammaC=15, gammaP=17.5, consistent=True)),
    "P5b": ("lawn_half", G(winSize=11, maxDisparity=50, minDisparity=8, gamma=10, fMax=120, iterations=3, bins=20)),
}


def main():
    ref = _oracle.ref_module()
    if ref is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` first")
    pairs = inputs()
    maps, meta = {}, {}
    for cid, (pair, p) in CASES.items():
        a, b = pairs[pair + "_L"], pairs[pair + "_R"]
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
        meta[cid] = dict(pair=pair, params=p, shape=list(d.shape), sha256=hashlib.sha256(d.tobytes()).hexdigest(),
                         checksum=int(d.astype(np.int64).sum()),
                         input_sha256=hashlib.sha256(a.tobytes() + b.tobytes()).hexdigest(), ref_seconds=round(dt, 2))
        print("%-4s %-13s %s sum=%d  %.1fs" % (cid, pair, meta[cid]["sha256"][:16], meta[cid]["checksum"], dt), flush=True)
    np.savez_compressed(os.path.join(OUT, "photo_pairs.npz"), **pairs)
    np.savez_compressed(os.path.join(OUT, "photo_cases.npz"), **maps)
    with open(os.path.join(OUT, "photo_cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
