"""GPU tier: REAL PHOTOGRAPHS against the unmodified reference.

tests/golden/photo_cases.npz holds the reference extension's (oracle/_ref, _passive.cpp) maps of the pair the reference ships for
its own ASW example (examples/009 StereoMatchingASW.py:20-39: examples/res/2/lawn_{L,R}.png, rectified with the shipped rig): the
quarter-size pair with the example's verbatim parameters (win 35, D 4..25, gammaC 15; plain, consistent, and the class defaults of
StereoASW() / StereoGSW()), two full-width 1280 x 64 strips of the native-size pair with D 4..100 (brick / chair / bicycle; grass +
the black margin rectification leaves), a 1280 x 48 strip of an unrectified capture from examples/res/new (ASW consistent and
GSW), and the WHOLE rectified frame at half size (640 x 360: sky, brick, chair, bicycle, mower, lawn, black margins; ASW win 35
D 8..50 consistent, GSW); inputs in photo_pairs.npz, generator tests/golden/make_golden_photo.py.  Smooth natural content -- lawn, defocus, the
constant margins -- is what the Tsukuba crops and the synthetic frames of the other goldens do not contain.

Bars, WITHOUT any tie exclusion: ASW (fp32 kernels vs the fp64 reference) >= 99.5 % of all pixels within 1 level (north_star) and
>= 99 % identical; GSW: 0 mismatching pixels.  When an ASW bar fails the test prints the tie audit (the GPU's raw cost at the
reference's disparity against its own minimum) so that a deviation can be stated and counted."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def photo():
    return (np.load(os.path.join(G, "photo_cases.npz")), json.load(open(os.path.join(G, "photo_cases.json"))),
            np.load(os.path.join(G, "photo_pairs.npz")))


def _pair(pairs, name):
    return np.ascontiguousarray(pairs[name + "_L"]), np.ascontiguousarray(pairs[name + "_R"])


def tie_audit(a, b, p, d, ref):
    """for every pixel whose disparity differs: the GPU's own cost at the reference's choice relative to the GPU's minimum"""
    from simplestereo_amd import _native
    H, W = a.shape[:2]
    costs = np.empty((H, W, p["maxDisparity"] - p["minDisparity"] + 1), np.float32)
    _native.check(_native.lib().ssamd_asw_costs(a.ctypes.data, b.ctypes.data, H, W, p["winSize"], p["maxDisparity"],
                                                p["minDisparity"], float(p["gammaC"]), float(p["gammaP"]), costs.ctypes.data, -1))
    ys, xs = np.nonzero(d != ref)
    out = []
    for y, x in zip(ys, xs):
        kd, kr = int(d[y, x]) - p["minDisparity"], int(ref[y, x]) - p["minDisparity"]
        if 0 <= kd < costs.shape[2] and 0 <= kr < costs.shape[2]:
            cg, cr = float(costs[y, x, kd]), float(costs[y, x, kr])
            out.append(abs(cr - cg) / max(1.0, abs(cg)))
    return np.array(out)


@pytest.mark.parametrize("cid", ["P1", "P1c", "P2a", "P2b", "P3a", "P4a", "P5a"])
def test_asw_photographs_vs_reference(cid, photo):
    import simplestereo_amd as ss
    maps, meta, pairs = photo
    m = meta[cid]
    a, b = _pair(pairs, m["pair"])
    p = {k: v for k, v in m["params"].items() if k != "algo"}
    d = ss.passive.StereoASW(**p).compute(a, b)
    ref = maps[cid]
    assert d.shape == ref.shape and d.dtype == np.int16
    diff = np.abs(d.astype(np.int32) - ref.astype(np.int32))
    within1, exact = float(np.mean(diff <= 1)), float(np.mean(diff == 0))
    print("%s %s %dx%d D %d..%d consistent=%s: exact %.4f %%, within-1 %.4f %% (%d / %d pixels differ)" %
          (cid, m["pair"], a.shape[1], a.shape[0], p["minDisparity"], p["maxDisparity"], p["consistent"], 100 * exact,
           100 * within1, int(np.count_nonzero(diff)), diff.size))
    if within1 < 0.995 or exact < 0.99:
        if not p["consistent"]:
            t = tie_audit(a, b, p, d, ref)
            print("tie audit: %d differing pixels, relative cost gap at the reference's choice: median %.3g, max %.3g, "
                  "%d above 1e-6" % (t.size, np.median(t) if t.size else 0, t.max() if t.size else 0, int(np.sum(t > 1e-6))))
    assert within1 >= 0.995, (cid, within1)
    assert exact >= 0.99, (cid, exact)
    import torch
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert np.array_equal(ss.passive.StereoASW(**p).compute(ta, tb).cpu().numpy(), d)


@pytest.mark.parametrize("cid", ["P3b", "P4b", "P5b"])
def test_gsw_photographs_vs_reference_bit_exact(cid, photo):
    import simplestereo_amd as ss
    maps, meta, pairs = photo
    m = meta[cid]
    a, b = _pair(pairs, m["pair"])
    p = {k: v for k, v in m["params"].items() if k != "algo"}
    d = ss.passive.StereoGSW(**p).compute(a, b)
    assert np.array_equal(d, maps[cid]), (cid, int(np.count_nonzero(d != maps[cid])))


def test_example_009_pipeline_on_device(photo):
    """the reference's example end to end on device tensors: rectified quarter-size pair -> StereoASW(example parameters) ->
    the reference's map, and the strip path (two row strips with their halo) reproduces the one-launch map"""
    import simplestereo_amd as ss
    maps, meta, pairs = photo
    a, b = _pair(pairs, "lawn_quarter")
    p = {k: v for k, v in meta["P1"]["params"].items() if k != "algo"}
    d = ss.passive.StereoASW(**p).compute(a, b)
    assert float(np.mean(d == maps["P1"])) >= 0.99
    pad, H = p["winSize"] // 2, a.shape[0]
    cut = H // 2
    top = ss.passive.StereoASW(**p).compute(np.ascontiguousarray(a[:cut + pad]), np.ascontiguousarray(b[:cut + pad]))[:cut]
    bot = ss.passive.StereoASW(**p).compute(np.ascontiguousarray(a[cut - pad:]), np.ascontiguousarray(b[cut - pad:]))[pad:]
    assert np.array_equal(np.concatenate([top, bot]), d)
