"""GPU tier: the fp64 tie-break pass of StereoASW (`exact=True`, the default since round 6; ssamd_asw_exact*; csrc/asw_exact_kernels.hip.h).

The fp32 kernels may pick the other one of two candidates whose costs agree to ~1e-6 relative; the reference decides
those in double (reference _passive.cpp:23, 56-95).  With exact=True every candidate within 128 ulps of its pixel's winning
cost image is re-evaluated in fp64 with the reference's expression and summation order and the argmin redone.  Bars here:
the goldens on which the default path is NOT 100 % identical to the reference become (near) 100 %; on seeded fuzz cases
every remaining difference from the fp64 oracle is a tie of the oracle's own costs at 1e-12 relative; every kernel family,
the strip path, the consistent mode and the queue-overflow path are covered."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ss():
    import simplestereo_amd
    return simplestereo_amd


def _photo(cid):
    maps = np.load(os.path.join(G, "photo_cases.npz"))
    meta = json.load(open(os.path.join(G, "photo_cases.json")))
    pairs = np.load(os.path.join(G, "photo_pairs.npz"))
    m = meta[cid]
    p = {k: v for k, v in m["params"].items() if k != "algo"}
    return np.ascontiguousarray(pairs[m["pair"] + "_L"]), np.ascontiguousarray(pairs[m["pair"] + "_R"]), p, maps[cid]


@pytest.mark.parametrize("cid,bar", [("P4a", 1.0), ("P2a", 1.0), ("P5a", 1.0), ("P1", 1.0), ("P1c", 1.0), ("P3a", 1.0), ("P2b", 1.0)])
def test_exact_mode_on_photographs(cid, bar, ss):
    """P4a (class default on the quarter-size lawn pair) is 99.90 % identical on the fp32 path, P2a / P5a miss 2 / 4 pixels"""
    a, b, p, ref = _photo(cid)
    d32 = ss.passive.StereoASW(exact=False, **p).compute(a, b)
    d64 = ss.passive.StereoASW(exact=True, **p).compute(a, b)
    e32, e64 = float(np.mean(d32 == ref)), float(np.mean(d64 == ref))
    from simplestereo_amd import _native
    print("%s: identical to the reference fp32 %.5f %% -> exact %.5f %% (%d -> %d pixels differ); %d candidates re-evaluated, %d + %d pixels flagged" %
          (cid, 100 * e32, 100 * e64, int(np.count_nonzero(d32 != ref)), int(np.count_nonzero(d64 != ref)),
           _native.counter("exact_entries"), _native.counter("exact_flagged_left"), _native.counter("exact_flagged_right")))
    assert _native.counter("exact_overflow") == 0
    assert e64 >= bar, (cid, e64)
    assert e64 >= e32


@pytest.mark.parametrize("cid", ["G1", "G2", "G3", "G5c", "G5d", "G5g", "G6a", "G6b", "G6c", "G6f", "G6g", "G8b", "G9a", "G9b", "G9c", "G9g", "G9h"])
def test_exact_mode_on_the_small_goldens(cid, ss, golden_cases, golden_inputs):
    """every ASW golden of cases.npz, all kernel families (wave / wave6 / phase-shifted / plain) and both passes: 100 %
    identical to the reference's map, or -- where not -- at least as close as the fp32 path"""
    maps, meta = golden_cases
    m = meta[cid]
    a, b = golden_inputs(m["input"])
    p = {k: v for k, v in m["params"].items() if k != "algo"}
    d32 = ss.passive.StereoASW(exact=False, **p).compute(a, b)
    d64 = ss.passive.StereoASW(exact=True, **p).compute(a, b)
    n32, n64 = int(np.count_nonzero(d32 != maps[cid])), int(np.count_nonzero(d64 != maps[cid]))
    print("%s: pixels differing from the reference fp32 %d -> exact %d of %d" % (cid, n32, n64, d64.size))
    assert n64 == 0, (cid, n64, n32)         # bit-identical weights (glibc's exp / powf restated): the reference's map, ties included


def test_exact_mode_wide_strip_is_identical(ss):
    """W3a / W3b: 1920-wide strips of the config-3 frames (the bench geometry) -- 99.999 % on the fp32 path"""
    from simplestereo_amd.synth import make_pair
    maps = np.load(os.path.join(G, "wide_cases.npz"))
    meta = json.load(open(os.path.join(G, "wide_cases.json")))
    for cid in ("W3a", "W3b"):
        m = meta[cid]
        H, W, maxD, seed = m["frame"]
        L, R, _ = make_pair(H, W, maxD, seed)
        a = np.ascontiguousarray(L[m["row0"]:m["row0"] + m["rows"]])
        b = np.ascontiguousarray(R[m["row0"]:m["row0"] + m["rows"]])
        p = {k: v for k, v in m["params"].items() if k != "algo"}
        d64 = ss.passive.StereoASW(exact=True, **p).compute(a, b)
        n = int(np.count_nonzero(d64 != maps[cid]))
        print("%s exact: %d of %d pixels differ from the reference" % (cid, n, d64.size))
        assert n == 0, (cid, n)


def _oracle_ties(a, b, p, d):
    """pixels of d that differ from the fp64 oracle's map and are NOT ties of the oracle's own costs at 1e-12 relative"""
    from oracle import oracle
    ref, cref = oracle.asw(a, b, return_costs=True, **p)
    bad, ties = 0, 0
    for y, x in np.argwhere(d != ref):
        row = cref[y, x]
        cg, cr = row[int(d[y, x]) - p["minDisparity"]], row[int(ref[y, x]) - p["minDisparity"]]
        if np.isfinite(cg) and abs(cg - cr) <= 1e-12 * max(1.0, abs(cr)):
            ties += 1
        else:
            bad += 1
    return bad, ties


@pytest.mark.parametrize("seed", range(8))
def test_exact_mode_fuzz_vs_fp64_oracle(seed, ss):
    """seeded random shapes / windows / ranges on frames with saturated regions (make_pair's occlusion noise): whatever
    differs from the oracle's left-referenced map must be an exact tie of the oracle's fp64 costs"""
    from simplestereo_amd.synth import make_pair
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(20, 70)), int(rng.integers(60, 260))
    win = int(rng.choice([5, 9, 15, 21, 35]))
    minD = int(rng.integers(0, 4))
    maxD = minD + int(rng.choice([3, 7, 16, 17, 31, 64, 70, 100]))
    L, R, _ = make_pair(H, W, max(8, maxD), seed)
    p = dict(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=float(3 + seed), gammaP=float(6 + 3 * seed))
    d64 = ss.passive.StereoASW(exact=True, **p).compute(L, R)
    bad, ties = _oracle_ties(L, R, p, d64)
    d32 = ss.passive.StereoASW(exact=False, **p).compute(L, R)
    bad32, ties32 = _oracle_ties(L, R, p, d32)
    print("seed %d %dx%d win %d D %d..%d: exact mode %d non-tie + %d tie differences (fp32 path: %d + %d)" %
          (seed, W, H, win, minD, maxD, bad, ties, bad32, ties32))
    assert bad == 0 and ties == 0, (seed, bad, ties)      # the oracle's map itself


def test_exact_mode_consistent_vs_oracle_and_strips(ss):
    """consistent=True (both argmins are tie-broken) against the oracle, and two row strips with their halo through the
    device entry point reproduce the whole-frame rows"""
    import torch
    from oracle import oracle
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(64, 200, 40, 21)
    p = dict(winSize=15, maxDisparity=40, minDisparity=2, gammaC=6.0, gammaP=12.0, consistent=True)
    m = ss.passive.StereoASW(exact=True, **p)
    d = m.compute(L, R)
    ref = oracle.asw(L, R, **p)
    n = int(np.count_nonzero(d != ref))
    n32 = int(np.count_nonzero(ss.passive.StereoASW(exact=False, **p).compute(L, R) != ref))
    print("consistent exact: %d pixels differ from the oracle (fp32 path %d)" % (n, n32))
    assert n == 0, (n, n32)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    whole = m.compute(tL, tR).cpu().numpy()
    assert np.array_equal(whole, d)
    pad, cut = 7, 30
    top = m._compute_device(tL[:cut + pad].contiguous(), tR[:cut + pad].contiguous(), 0, cut).cpu().numpy()
    bot = m._compute_device(tL[cut - pad:].contiguous(), tR[cut - pad:].contiguous(), pad, 64 - cut).cpu().numpy()
    assert np.array_equal(np.concatenate([top, bot]), d)


def test_exact_mode_queue_overflow_keeps_the_fp32_map(ss):
    """a queue too small for the flagged candidates (test hook SSAMD_EXACT_CAP): the keys stay untouched, the overflow is counted"""
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(48, 160, 32, 5)
    p = dict(winSize=9, maxDisparity=32)
    d32 = ss.passive.StereoASW(exact=False, **p).compute(L, R)
    with _native.options(SSAMD_EXACT_CAP="4"):
        d = ss.passive.StereoASW(exact=True, **p).compute(L, R)
        assert _native.counter("exact_overflow") == 1 and _native.counter("exact_entries") > 4
    assert np.array_equal(d, d32)
    d64 = ss.passive.StereoASW(exact=True, **p).compute(L, R)
    assert _native.counter("exact_overflow") == 0
    assert d64.shape == d32.shape


def test_exact_mode_raw_queue_overflow_of_a_merging_call_keeps_the_fp32_map(ss):
    """consistent=True goes through the RAW queue (tile-local near-ties + cost images) before the filter: a raw queue too small for
    them (test hook SSAMD_EXACT_RAWCAP) must leave the fp32 map, count the overflow and warn -- every later kernel of the pass returns"""
    import warnings
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(40, 300, 64, 9)
    L = (L // 64 * 64).astype(np.uint8)                       # quantised colours: near-ties in every tile
    p = dict(winSize=9, maxDisparity=64, consistent=True)
    d32 = ss.passive.StereoASW(exact=False, **p).compute(L, R)
    with _native.options(SSAMD_EXACT_RAWCAP="3"):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            d = ss.passive.StereoASW(exact=True, **p).compute(L, R)
        assert _native.counter("exact_overflow") == 1 and _native.counter("exact_raw_entries") > 3
        assert any(issubclass(x.category, RuntimeWarning) for x in w)
    assert np.array_equal(d, d32)
    from oracle import oracle
    d64 = ss.passive.StereoASW(exact=True, **p).compute(L, R)
    assert _native.counter("exact_overflow") == 0 and _native.counter("exact_raw_entries") > 3
    assert np.array_equal(d64, oracle.asw(L, R, **p))


@pytest.mark.parametrize("cons", [False, True])
def test_black_margins_of_rectified_frames(cons, ss):
    """rectified frames carry black margins: deep inside them every candidate of a pixel costs exactly 0 and ties every other -- a first
    form of the in-kernel selection queued 3e7 of them on a 1080p frame (overflow: the whole frame fell back to the fp32 map).  Such
    candidates are no longer queued: a pixel whose fp32 winner costs exactly 0 is settled by asw_exact_zero_kernel with integer compares
    (all taps TAD = 0 => the reference's cost is exactly 0.0 and the smallest such index is its first minimum).  Same map as the oracle,
    a short queue, no overflow."""
    from oracle import oracle
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(48, 420, 64, 31)
    L[:, :150] = 0; R[:, :150] = 0                       # a black band on the left of both views
    L[:6] = 0; R[:6] = 0                                 # ... and along the top
    L[:, 400:] = 0                                       # the left view also on the right (the right view keeps its content there)
    L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
    p = dict(winSize=15, maxDisparity=64, minDisparity=0, consistent=cons)
    d = ss.passive.StereoASW(**p).compute(L, R)
    n = _native.counter("exact_entries")
    assert _native.counter("exact_overflow") == 0
    ref = oracle.asw(L, R, **p)
    zero_cost_pixels = int(np.count_nonzero((L[:, :, 0] == 0) & (d >= 0)))
    print("black margins, consistent=%s: %d candidates in the queue, %d pixels differ from the oracle" % (cons, n, int(np.count_nonzero(d != ref))))
    assert n < 40000 and zero_cost_pixels > 5000         # (the first form queued 367 072 here)
    assert np.array_equal(d, ref)


@pytest.mark.parametrize("cons", [False, True])
def test_zero_cost_winners_whose_weights_underflow(cons, ss):
    """the other branch of asw_exact_zero_kernel: with a tiny gammaC the fp32 weights of taps with TAD > 0 underflow to 0, so a candidate's
    fp32 cost is exactly 0 although its window is NOT all-zero -- the reference's fp64 cost is positive.  Such pixels get all their
    candidates re-evaluated; the map is the oracle's."""
    from oracle import oracle
    from simplestereo_amd import _native
    rng = np.random.default_rng(77)
    H, W = 30, 160
    L = np.zeros((H, W, 3), np.uint8); R = np.zeros((H, W, 3), np.uint8)
    L[:, :, :] = rng.integers(0, 2, (H, W, 1)) * 255          # black / white speckle: Lab distances of 100 between neighbours
    R[:, :-3] = L[:, 3:]                                      # shifted by 3; elsewhere black
    L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
    p = dict(winSize=9, maxDisparity=12, minDisparity=0, consistent=cons, gammaC=0.7, gammaP=17.5)
    d = ss.passive.StereoASW(**p).compute(L, R)
    assert _native.counter("exact_overflow") == 0
    ref = oracle.asw(L, R, **p)
    d32 = ss.passive.StereoASW(exact=False, **p).compute(L, R)
    print("underflowing weights, consistent=%s: exact %d, fp32 %d pixels differ from the oracle; %d candidates re-evaluated" %
          (cons, int(np.count_nonzero(d != ref)), int(np.count_nonzero(d32 != ref)), _native.counter("exact_entries")))
    assert np.array_equal(d, ref)


def test_exact_mode_argument_errors(ss):
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(24, 64, 8, 1)
    with pytest.raises(ValueError):
        ss.passive.StereoASW(exact=True, alternate=True, maxDisparity=8).compute(L, R)
    # devices=[...]: one strip per listed GPU, each tie-broken on its own device (the 1-GPU box lists GPU 0 twice: test hook)
    from simplestereo_amd import _native
    with _native.options(SSAMD_MULTI_ALLOW_REPEAT="1"):
        two = ss.passive.StereoASW(exact=True, maxDisparity=8, winSize=9).compute(L, R, devices=[0, 0])
    assert np.array_equal(two, ss.passive.StereoASW(exact=True, maxDisparity=8, winSize=9).compute(L, R))
    # empty candidate loops (maxDisparity < minDisparity): nothing to break ties between, same output as the plain call
    a = ss.passive.StereoASW(exact=True, winSize=5, maxDisparity=3, minDisparity=5).compute(L, R)
    assert np.array_equal(a, ss.passive.StereoASW(exact=False, winSize=5, maxDisparity=3, minDisparity=5).compute(L, R))


def test_device_exp_and_powf_are_the_hosts_libm_bit_for_bit():
    """csrc/glibc_math.hip.h on the device (ssamd_debug_libm) against the libm of this process: exp on the arguments the support
    weights use and on the rescaled / subnormal / special branches, powf(x, (float)(1/3.0)) on every 8th float of the range the
    Lab conversion can produce (all of them on the host: oracle/libm_check.c)"""
    import math
    from simplestereo_amd import _native
    rng = np.random.default_rng(5)
    x = np.concatenate([-60.0 * rng.random(400000) * rng.random(400000), -760.0 * rng.random(300000), -1100.0 * rng.random(100000),
                        2.0 * rng.random(50000), 1e-17 * rng.random(1000) - 5e-18,
                        np.array([0.0, -0.0, -np.inf, np.inf, -745.2, -746.0, -1023.9, -1024.0, -5000.0, 709.7, 710.0, 1e-300, -1e-300, -512.0])])
    out = np.empty_like(x)
    _native.check(_native.lib().ssamd_debug_libm(0, x.size, x.ctypes.data, out.ctypes.data))
    want = np.array([math.exp(v) if v < 709.78 else np.inf for v in x])
    assert np.array_equal(out.view(np.uint64), want.view(np.uint64)), int(np.count_nonzero(out.view(np.uint64) != want.view(np.uint64)))
    lo, hi = np.float32(0.008856).view(np.uint32), np.float32(1.3).view(np.uint32)
    xf = np.arange(int(lo), int(hi), 8, dtype=np.uint32).view(np.float32)
    of = np.empty_like(xf)
    _native.check(_native.lib().ssamd_debug_libm(1, xf.size, xf.ctypes.data, of.ctypes.data))
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    third = float(np.float32(1 / 3.0))
    sample = rng.integers(0, xf.size, 200000)
    wantf = np.array([libm.powf(float(xf[i]), third) for i in sample], dtype=np.float32)
    assert np.array_equal(of[sample].view(np.uint32), wantf.view(np.uint32))


def test_lab_records_are_the_references_doubles_rounded():
    """with glibc's powf restated, ssamd_bgr2lab returns EXACTLY float32(reference Lab) -- the C restatement of
    colorconversion.hpp (pinned bit-exact against the reference's header) on a colour-cube sample and on random pixels"""
    from oracle import oracle
    from simplestereo_amd import _native
    rng = np.random.default_rng(9)
    g = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 1, 3)
    img = np.ascontiguousarray(np.concatenate([cube, rng.integers(0, 256, (60000, 1, 3), dtype=np.uint8)]))
    H, W = img.shape[:2]
    lab = np.empty((H, W, 3), np.float32)
    _native.check(_native.lib().ssamd_bgr2lab(img.ctypes.data, H, W, lab.ctypes.data, -1))
    want = oracle.bgr2lab(img).astype(np.float32)
    assert np.array_equal(lab.view(np.uint32), want.view(np.uint32)), int(np.count_nonzero(lab != want))


def test_exact_mode_full_bench_frame_is_the_references_map():
    """the WHOLE bench frame (tests/golden/full_cases.npz): plain and consistent, every pixel -- saturated patches included"""
    from simplestereo_amd.synth import make_pair
    path = os.path.join(G, "full_cases.npz")
    maps = np.load(path)
    L, R, _ = make_pair(1080, 1920, 192, 1)
    for cid, cons in (("F3p", False), ("F3c", True)):
        if cid not in maps.files:
            continue
        d = __import__("simplestereo_amd").passive.StereoASW(winSize=35, maxDisparity=192, consistent=cons, exact=True).compute(L, R)
        n = int(np.count_nonzero(d != maps[cid]))
        print("%s exact: %d of %d pixels differ from the reference" % (cid, n, d.size))
        assert n == 0, (cid, n)


def test_exact_pass_costs_and_lab_are_the_oracles_bits():
    """ssamd_debug_exact_costs: the fp64 cost the tie-break pass computes for random candidates, and the fp64 Lab images it
    reads, against the C restatement's (bit-exact with the reference on every golden) -- every bit, flat / saturated content and
    extreme gammas included; and the device's corrected sqrt and its division against the host's"""
    from oracle import oracle
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    rng = np.random.default_rng(11)
    for (H, W, win, maxd, gc, gp, flat) in [(10, 221, 35, 56, 5.0, 17.5, True), (53, 66, 21, 39, 0.7, 17.5, True), (30, 120, 9, 30, 5.0, 17.5, False),
                                             (5, 199, 21, 50, 0.7, 3.0, False)]:
        L, R, _ = make_pair(H, W, maxd, int(rng.integers(0, 1 << 30)))
        if flat:
            L = (L // 64 * 64).astype(np.uint8)
            R = np.ascontiguousarray(R[:, ::-1])
        L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
        _, cref = oracle.asw(L, R, winSize=win, maxDisparity=maxd, gammaC=gc, gammaP=gp, return_costs=True)
        n = 2000
        ys, xs = rng.integers(0, H, n), rng.integers(0, W, n)
        ds = np.minimum(rng.integers(0, maxd + 1, n), xs)
        yxd = np.ascontiguousarray(np.stack([ys, xs, ds], 1).astype(np.int32))
        costs, l1, l2 = np.empty(n), np.empty((H, W, 3)), np.empty((H, W, 3))
        _native.check(_native.lib().ssamd_debug_exact_costs(L.ctypes.data, R.ctypes.data, H, W, win, gc, gp, n, yxd.ctypes.data, costs.ctypes.data,
                                                            l1.ctypes.data, l2.ctypes.data))
        assert np.array_equal(costs.view(np.uint64), cref[ys, xs, ds].view(np.uint64)), (H, W, win)
        assert np.array_equal(l1.view(np.uint64), oracle.bgr2lab(L).view(np.uint64)) and np.array_equal(l2.view(np.uint64), oracle.bgr2lab(R).view(np.uint64))
    x = np.concatenate([rng.random(400000) * 3.0e4, rng.random(200000) * 50.0, (rng.integers(0, 400, 100000) ** 2).astype(np.float64), np.array([0.0, 1.0, 2.0])])
    out = np.empty_like(x)
    _native.check(_native.lib().ssamd_debug_libm(2, x.size, x.ctypes.data, out.ctypes.data))
    assert np.array_equal(out.view(np.uint64), np.sqrt(x).view(np.uint64))
    _native.check(_native.lib().ssamd_debug_libm(3, x.size, x.ctypes.data, out.ctypes.data))
    assert np.array_equal(out.view(np.uint64), (x / 0.7 + x / 5.0).view(np.uint64))


def test_exact_mode_mini_soak_against_the_oracle(ss):
    """60 seeded random cases of tools/soak_r05.py's generator (flat / saturated content, windows 3..35, ranges 1..120, plain and
    consistent, gammaC down to 0.7): every exact-mode map is the fp64 oracle's.  The cases that taught the pass its near-tie
    rules live here: candidates whose fp32 images differ although the reference's fp64 costs are within its own rounding noise of
    each other (saturated side: absolute bound ~win^2 ulps of 40), and pairs either side of cost = 20 where the image changes form."""
    from oracle import oracle
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    rng = np.random.default_rng(5)
    done = 0
    for _ in range(60):
        H, W = int(rng.integers(4, 60)), int(rng.integers(8, 400))
        win = int(rng.choice([3, 5, 9, 11, 15, 21, 27, 35]))
        nD, mind = int(rng.integers(1, 121)), int(rng.choice([0, 0, 1, 5]))
        maxd = mind + nD - 1
        p = dict(winSize=win, maxDisparity=maxd, minDisparity=mind, consistent=bool(rng.random() < 0.5),
                 gammaC=float(rng.choice([5.0, 7.0, 0.7, 50.0])), gammaP=float(rng.choice([17.5, 3.0, 100.0])))
        L, R, _ = make_pair(H, W, max(1, maxd), int(rng.integers(0, 1 << 30)))
        if rng.random() < 0.4:
            L = (L // 64 * 64).astype(np.uint8)
            R = np.ascontiguousarray(R[:, ::-1])
        L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
        if H * W * nD * win * win > 3e8:
            continue
        d = ss.passive.StereoASW(exact=True, **p).compute(L, R)
        assert _native.counter("exact_overflow") == 0
        ref = oracle.asw(L, R, **p)
        assert np.array_equal(d, ref), (H, W, p, int(np.count_nonzero(d != ref)))
        done += 1
    assert done >= 40


@pytest.mark.parametrize("win,H,W,maxd,cons", [(61, 40, 180, 24, False), (101, 36, 160, 12, True), (151, 30, 200, 10, False), (255, 20, 300, 6, False)])
def test_large_windows_against_the_oracle(win, H, W, maxd, cons, ss):
    """the near-tie band was a fixed 128 key ulps whatever the window -- at winSize 35 the (N, S') sums have 1 225
    fp32 terms, at the allowed 255 they have 65 025.  The band now grows with the window (ssamd_api.hip asw_exact_prepare); here
    windows of 61 .. 255 on textured, quantised and mirrored content against the fp64 oracle: the maps are equal, and the fp32
    argmin alone stays inside north_star's tolerance."""
    from oracle import oracle
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    for variant in range(3):
        L, R, _ = make_pair(H, W, max(8, maxd), 700 + win + variant)
        if variant == 1:                                   # quantised colours + a mirrored right image: wide saturated regions
            L = (L // 64 * 64).astype(np.uint8)
            R = np.ascontiguousarray(R[:, ::-1])
        if variant == 2:                                   # low contrast: many candidates at nearly equal, unsaturated costs
            L = (L // 8 + 100).astype(np.uint8)
            R = (R // 8 + 100).astype(np.uint8)
        L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
        p = dict(winSize=win, maxDisparity=maxd, minDisparity=0, consistent=cons, gammaC=5.0, gammaP=17.5)
        ref = oracle.asw(L, R, **p)
        d = ss.passive.StereoASW(exact=True, **p).compute(L, R)
        assert _native.counter("exact_overflow") == 0
        n = int(np.count_nonzero(d != ref))
        d32 = ss.passive.StereoASW(exact=False, **p).compute(L, R)
        n32 = int(np.count_nonzero(d32 != ref))
        print("win %d variant %d: exact %d, fp32 %d of %d pixels differ from the oracle; %d candidates re-evaluated" %
              (win, variant, n, n32, d.size, _native.counter("exact_entries")))
        assert n == 0, (win, variant, n, n32)
