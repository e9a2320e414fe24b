"""GPU box (round 5): where StereoASW(exact=True) still differs from the fp64 oracle on small cases -- the fp32 cost images (ulps
between the candidates) next to the oracle's fp64 costs: a tolerance miss (fp32 error > 128 ulps), or an exact tie?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
from oracle import oracle

G = os.path.join(ROOT, "tests", "golden")
z = np.load(os.path.join(G, "tsukuba_pair.npz"))
Lt, Rt = np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])
crop = (np.ascontiguousarray(Lt[100:118, 150:190]), np.ascontiguousarray(Rt[100:118, 150:190]))
cases = {"G6c": (make_pair(64, 96, 24, 5)[:2], dict(winSize=9, maxDisparity=24, minDisparity=2, gammaC=7.5, gammaP=36.0)),
         "G9b": (crop, dict(winSize=9, maxDisparity=12, minDisparity=0, gammaC=0.7, gammaP=3.0)),
         "T15": (make_pair(64, 200, 40, 21)[:2], dict(winSize=15, maxDisparity=40, minDisparity=2, gammaC=6.0, gammaP=12.0))}


def key_of(c):
    """asw_cost_key's image of a float32 cost (csrc/asw_kernels.hip.h) -- from the dumped cost itself, so the low side only"""
    return np.float32(c).view(np.uint32)


for name, ((a, b), p) in cases.items():
    H, W = a.shape[:2]
    nD = p["maxDisparity"] - p["minDisparity"] + 1
    ref, cref = oracle.asw(a, b, return_costs=True, **p)
    d32 = ss.passive.StereoASW(**p).compute(a, b)
    d64 = ss.passive.StereoASW(exact=True, **p).compute(a, b)
    c32 = np.empty((H, W, nD), np.float32)
    _native.check(_native.lib().ssamd_asw_costs(a.ctypes.data, b.ctypes.data, H, W, p["winSize"], p["maxDisparity"], p["minDisparity"],
                                                float(p["gammaC"]), float(p["gammaP"]), c32.ctypes.data, -1))
    print("== %s left-referenced: fp32 differs from the oracle on %d pixels, exact on %d; entries %d flagged %d" %
          (name, int(np.count_nonzero(d32 != ref)), int(np.count_nonzero(d64 != ref)), _native.counter("exact_entries"), _native.counter("exact_flagged_left")))
    for y, x in np.argwhere(d64 != ref)[:12]:
        kg, kr = int(d64[y, x]) - p["minDisparity"], int(ref[y, x]) - p["minDisparity"]
        f = c32[y, x]
        print("  (%d,%d) exact d=%d ref d=%d fp32 d=%d | fp64 cost at exact's %.17g at ref's %.17g (rel %.3g) | fp32 cost at exact's %.9g at ref's %.9g, gpu min %.9g; "
              "key ulps ref's - min = %d" % (y, x, d64[y, x], ref[y, x], d32[y, x], cref[y, x, kg], cref[y, x, kr], (cref[y, x, kg] - cref[y, x, kr]) / max(cref[y, x, kr], 1e-300),
                                            f[kg], f[kr], np.nanmin(f), int(key_of(f[kr])) - int(key_of(np.nanmin(f)))))
    pc = dict(p, consistent=True)
    refc = oracle.asw(a, b, **pc)
    e = ss.passive.StereoASW(exact=True, **pc).compute(a, b)
    print("   consistent: exact differs from the oracle on %d pixels (fp32 %d); flagged right %d" %
          (int(np.count_nonzero(e != refc)), int(np.count_nonzero(ss.passive.StereoASW(**pc).compute(a, b) != refc)), _native.counter("exact_flagged_right")))
    # the two raw argmins of the fp32 path against the oracle's (ssamd_asw_argmins): where does the right pass differ?
    left, right = np.empty((H, W), np.int16), np.empty((H, W), np.int16)
    _native.check(_native.lib().ssamd_asw_argmins(a.ctypes.data, b.ctypes.data, H, W, p["winSize"], p["maxDisparity"], p["minDisparity"],
                                                  float(p["gammaC"]), float(p["gammaP"]), left.ctypes.data, right.ctypes.data, -1))
    # oracle right argmin from its cost volume: right pixel xr picks the smallest xl = xr + d minimising cref[y, xl, d]
    bad = []
    for y in range(H):
        for xr in range(W):
            best, bx = np.inf, 0
            for d in range(p["minDisparity"], p["maxDisparity"] + 1):
                xl = xr + d
                if xl >= W:
                    break
                c = cref[y, xl, d - p["minDisparity"]]
                if c < best:
                    best, bx = c, xl
            if bx != right[y, xr]:
                bad.append((y, xr, int(right[y, xr]), bx))
    print("   fp32 right-referenced argmins differing from the oracle's: %d %s" % (len(bad), bad[:6]))
