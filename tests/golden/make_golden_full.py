#!/usr/bin/env python3
"""FULL-FRAME golden maps of the bench frame, from the UNMODIFIED reference (oracle/_ref).

    make -C oracle ref && python tests/golden/make_golden_full.py [ids...]     (build container only; ~1 h on 8 cores)

The accuracy half of BASELINE.json's metric ("% bad-1.0 vs CPU ref") used to be computed on 72-row strips
(wide_cases.npz W3a/W3b = 6.7 % of a frame each).  This script runs the reference extension -- compiled from
/root/reference/simplestereo/_passive.cpp where it lies, `computeASW` :293-400 and `computeGSW` :703-774 -- on the
WHOLE frame that bench.py times, `simplestereo_amd.synth.make_pair(1080, 1920, 192, seed=1)`:

    F3p  config 3: ASW win 35, D 0..192, gammaC 5, gammaP 17.5, consistent=False      (~15 min on 8 threads)
    F3c  config 3 with consistent=True  (the reference does two aggregation passes)    (~30 min)
    F4   config 4: GSW class defaults win 11, D 0..192, gamma 10, fMax 120, it 3       (~10 min)
    F5p  config 5: ASW win 35, D 0..256 on make_pair(2160, 4096, 256, seed=1)         (~1.5 h; only when named)

Output: full_cases.npz (int16 maps, ~1 MB each compressed) + full_cases.json (recipe, parameters, sha256 of the
map and of the input bytes).  Inputs are regenerated from the recipe (make_pair is deterministic), not stored.
Each case is written as soon as it is done, so an interrupted run keeps what it finished.
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
FRAME = (1080, 1920, 192, 1)        # H, W, maxDisparity of the generator, seed  == bench.py's default workload

A = lambda **k: dict(algo="asw", **k)   # noqa: E731
G = lambda **k: dict(algo="gsw", **k)   # noqa: E731
CASES = {
    "F3p": A(winSize=35, maxDisparity=192, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False),
    "F4": G(winSize=11, maxDisparity=192, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20),
    "F3c": A(winSize=35, maxDisparity=192, minDisparity=0, gammaC=5, gammaP=17.5, consistent=True),
    # config 5's frame (bench.py --config c5_4k_d256_w35): make_pair(2160, 4096, 256, seed=1), ~1.5 h on 8 threads
    "F5p": A(winSize=35, maxDisparity=256, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False,
             frame=(2160, 4096, 256, 1)),
}


def _save(maps, meta):
    np.savez_compressed(os.path.join(OUT, "full_cases.npz"), **maps)
    with open(os.path.join(OUT, "full_cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    ref = _oracle.ref_module()
    if ref is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` first")
    want = [a for a in sys.argv[1:] if a in CASES] or [c for c in CASES if c != "F5p"]
    maps, meta = {}, {}
    if os.path.exists(os.path.join(OUT, "full_cases.npz")):
        old = np.load(os.path.join(OUT, "full_cases.npz"))
        maps = {k: old[k] for k in old.files}
        meta = json.load(open(os.path.join(OUT, "full_cases.json")))
    pairs = {}
    for cid in want:
        p = dict(CASES[cid])
        frame = tuple(p.pop("frame", FRAME))
        H, W, maxD, seed = frame
        if frame not in pairs:
            pairs[frame] = make_pair(H, W, maxD, seed)[:2]
        L, R = pairs[frame]
        t = time.time()
        if p["algo"] == "asw":
            d = ref.computeASW(L, R, p["winSize"], p["maxDisparity"], p["minDisparity"],
                               float(p["gammaC"]), float(p["gammaP"]), bool(p["consistent"]))
        else:
            d = ref.computeGSW(L, R, p["winSize"], p["maxDisparity"], p["minDisparity"],
                               p["gamma"], float(p["fMax"]), p["iterations"], p["bins"])
        dt = time.time() - t
        assert d.dtype == np.int16 and d.shape == (H, W)
        maps[cid] = d
        meta[cid] = dict(recipe="simplestereo_amd.synth.make_pair(%d,%d,%d,seed=%d), whole frame" % frame,
                         frame=list(frame), params=p, shape=[H, W],
                         sha256=hashlib.sha256(d.tobytes()).hexdigest(), checksum=int(d.astype(np.int64).sum()),
                         input_sha256=hashlib.sha256(L.tobytes() + R.tobytes()).hexdigest(),
                         ref_seconds=round(dt, 1), ref_threads=os.cpu_count())
        print("%-4s %s sum=%d  %.1fs" % (cid, meta[cid]["sha256"][:16], meta[cid]["checksum"], dt), flush=True)
        _save(maps, meta)


if __name__ == "__main__":
    main()
