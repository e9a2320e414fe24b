#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

It imports the reference extension compiled by oracle/Makefile into oracle/_ref
(`simplestereo/_passive.cpp`, never copied into this repo), feeds it the inputs
below and stores inputs + expected outputs as small .npz fixtures:

  tsukuba_pair.npz     the Middlebury Tsukuba pair shipped as example DATA by the
                       reference (examples/res/tsukuba/*.png), as BGR uint8 arrays
                       exactly as cv2.imread would give them, + ground truth / mask
  cases.npz            one int16 disparity map per case (key = case id)
  cases.json           per-case parameters, input recipe, sha256 of the map bytes
  errors.json          exception type/message probes of the native boundary

The reference holds no tests for this path, so these vectors are the parity pins.
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
TSUKUBA = "/root/reference/examples/res/tsukuba"


def _imread_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def _imread_gray(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("L")))


def main():
    ref = _oracle.ref_module()
    if ref is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` first")

    L = _imread_bgr(os.path.join(TSUKUBA, "tsukuba_l.png"))
    R = _imread_bgr(os.path.join(TSUKUBA, "tsukuba_r.png"))
    gt = _imread_gray(os.path.join(TSUKUBA, "groundtruth.png"))
    nonocc = _imread_gray(os.path.join(TSUKUBA, "nonocc.png"))
    np.savez_compressed(os.path.join(OUT, "tsukuba_pair.npz"), left=L, right=R, groundtruth=gt, nonocc=nonocc)

    cropL = np.ascontiguousarray(L[100:118, 150:190])
    cropR = np.ascontiguousarray(R[100:118, 150:190])

    def synth(h, w, maxd, seed):
        a, b, _ = make_pair(h, w, maxd, seed)
        return a, b

    inputs = {
        "tsukuba": (L, R),
        "crop": (cropL, cropR),
        "tsukuba_top": (np.ascontiguousarray(L[:40]), np.ascontiguousarray(R[:40])),
        "synth_96x128": synth(96, 128, 32, 0),
        "synth_64x96": synth(64, 96, 24, 5),
        "synth_480x640": synth(480, 640, 64, 0),
    }
    recipes = {
        "tsukuba": "tsukuba_pair.npz left/right",
        "crop": "tsukuba_pair.npz left/right [100:118,150:190]",
        "tsukuba_top": "tsukuba_pair.npz left/right [:40]",
        "synth_96x128": "simplestereo_amd.synth.make_pair(96,128,32,seed=0)",
        "synth_64x96": "simplestereo_amd.synth.make_pair(64,96,24,seed=5)",
        "synth_480x640": "simplestereo_amd.synth.make_pair(480,640,64,seed=0)",
    }

    A = lambda **k: dict(algo="asw", **k)   # noqa: E731
    G = lambda **k: dict(algo="gsw", **k)   # noqa: E731
    cases = {
        # id: (input, params)            SURVEY.md section 8c ids in comments
        "G1": ("tsukuba", A(winSize=15, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
        "G2": ("tsukuba", A(winSize=15, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        "G3": ("tsukuba", A(winSize=35, maxDisparity=14, minDisparity=4, gammaC=15, gammaP=17.5, consistent=True)),
        "G4": ("tsukuba", G(winSize=11, maxDisparity=16, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G5a": ("crop", G(winSize=5, maxDisparity=5, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G5b": ("crop", G(winSize=7, maxDisparity=6, minDisparity=1, gamma=10, fMax=60, iterations=2, bins=20)),
        "G5c": ("crop", A(winSize=7, maxDisparity=6, minDisparity=1, gammaC=5, gammaP=17.5, consistent=True)),
        "G5d": ("crop", A(winSize=35, maxDisparity=8, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
        "G5e": ("crop", G(winSize=3, maxDisparity=4, minDisparity=0, gamma=7, fMax=120, iterations=1, bins=20)),
        "G5f": ("crop", G(winSize=9, maxDisparity=45, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G5g": ("crop", A(winSize=5, maxDisparity=60, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        "G6a": ("synth_96x128", A(winSize=35, maxDisparity=32, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
        "G6b": ("synth_96x128", A(winSize=35, maxDisparity=32, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        "G6c": ("synth_64x96", A(winSize=9, maxDisparity=24, minDisparity=2, gammaC=7.5, gammaP=36, consistent=True)),
        "G6d": ("synth_64x96", G(winSize=11, maxDisparity=24, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G6e": ("synth_64x96", G(winSize=7, maxDisparity=20, minDisparity=3, gamma=25, fMax=80.5, iterations=2, bins=20)),
        "G6f": ("synth_480x640", A(winSize=35, maxDisparity=64, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
        "G6g": ("synth_480x640", A(winSize=35, maxDisparity=64, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        "G6h": ("synth_480x640", G(winSize=11, maxDisparity=64, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G8a": ("tsukuba_top", G(winSize=11, maxDisparity=16, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
        "G8b": ("tsukuba_top", A(winSize=21, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        # parameter corners: 1x1 window, extreme gammas, centre-only GSW weights (iterations=0), tiny / zero fMax,
        # negative GSW gamma (growing weights), empty candidate range, class defaults on a real image
        "G9a": ("crop", A(winSize=1, maxDisparity=8, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True)),
        "G9b": ("crop", A(winSize=9, maxDisparity=12, minDisparity=0, gammaC=0.7, gammaP=3.0, consistent=False)),
        "G9c": ("crop", A(winSize=9, maxDisparity=12, minDisparity=0, gammaC=60, gammaP=200, consistent=True)),
        "G9d": ("crop", G(winSize=7, maxDisparity=8, minDisparity=0, gamma=10, fMax=120, iterations=0, bins=20)),
        "G9e": ("crop", G(winSize=7, maxDisparity=8, minDisparity=0, gamma=1, fMax=5.5, iterations=3, bins=20)),
        "G9f": ("crop", G(winSize=5, maxDisparity=9, minDisparity=2, gamma=-7, fMax=120, iterations=3, bins=20)),
        "G9g": ("crop", A(winSize=5, maxDisparity=3, minDisparity=5, gammaC=5, gammaP=17.5, consistent=False)),
        "G9h": ("tsukuba_top", A(winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False)),
        "G9i": ("crop", G(winSize=5, maxDisparity=6, minDisparity=0, gamma=10, fMax=0, iterations=3, bins=20)),
        "G9j": ("crop", G(winSize=1, maxDisparity=6, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20)),
    }

    maps, meta = {}, {}
    for cid, (inp, p) in cases.items():
        a, b = inputs[inp]
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
        meta[cid] = dict(input=inp, recipe=recipes[inp], params=p, shape=list(d.shape),
                         sha256=hashlib.sha256(d.tobytes()).hexdigest(), checksum=int(d.astype(np.int64).sum()),
                         input_sha256=hashlib.sha256(a.tobytes() + b.tobytes()).hexdigest(),
                         ref_seconds=round(dt, 3))
        print("%-4s %-14s %s sum=%d  %.2fs" % (cid, inp, meta[cid]["sha256"][:16], meta[cid]["checksum"], dt), flush=True)
    np.savez_compressed(os.path.join(OUT, "cases.npz"), **maps)
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)

    # ---- exception probes of the native boundary (_passive.cpp:301-325, 709-733)
    def probe(fn, *args):
        try:
            fn(*args)
            return ["ok", ""]
        except Exception as e:      # noqa: BLE001
            return [type(e).__name__, str(e)]

    a, b = cropL, cropR
    errors = {
        "asw_even_win": probe(ref.computeASW, a, b, 4, 5, 0, 5.0, 17.5, False),
        "asw_zero_win": probe(ref.computeASW, a, b, 0, 5, 0, 5.0, 17.5, False),
        "asw_float_img1": probe(ref.computeASW, a.astype(np.float32), b, 5, 5, 0, 5.0, 17.5, False),
        "asw_gray": probe(ref.computeASW, a[:, :, 0].copy(), b[:, :, 0].copy(), 5, 5, 0, 5.0, 17.5, False),
        "asw_shape_mismatch": probe(ref.computeASW, a, np.ascontiguousarray(b[:-1]), 5, 5, 0, 5.0, 17.5, False),
        "asw_four_channels": probe(ref.computeASW, np.zeros((8, 8, 4), np.uint8), np.zeros((8, 8, 4), np.uint8), 5, 5, 0, 5.0, 17.5, False),
        "asw_float_win": probe(ref.computeASW, a, b, 5.0, 5, 0, 5.0, 17.5, False),
        "asw_list_input": probe(ref.computeASW, a.tolist(), b, 5, 5, 0, 5.0, 17.5, False),
        "asw_int_gammas": probe(ref.computeASW, a, b, 5, 5, 0, 5, 17, False),
        "asw_no_consistent_arg": probe(ref.computeASW, a, b, 5, 5, 0, 5.0, 17.5),
        "gsw_float_gamma": probe(ref.computeGSW, a, b, 5, 5, 0, 10.5, 120.0, 3, 20),
        "gsw_even_win": probe(ref.computeGSW, a, b, 6, 5, 0, 10, 120.0, 3, 20),
        "gsw_int_fmax": probe(ref.computeGSW, a, b, 5, 5, 0, 10, 120, 3, 20),
        "gsw_float_img1": probe(ref.computeGSW, a.astype(np.float64), b, 5, 5, 0, 10, 120.0, 3, 20),
        "gsw_shape_mismatch": probe(ref.computeGSW, a, np.ascontiguousarray(b[:, :-2]), 5, 5, 0, 10, 120.0, 3, 20),
    }
    with open(os.path.join(OUT, "errors.json"), "w") as f:
        json.dump(errors, f, indent=1, sort_keys=True)
    for k, v in errors.items():
        print("%-24s %s" % (k, v))


if __name__ == "__main__":
    main()
