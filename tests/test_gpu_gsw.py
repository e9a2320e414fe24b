"""Tier T2 (GPU): HIP GSW path.  The reference computes GSW in fp32 with a fixed operation
order; the kernels reproduce weights (host libm table), truncated colour distances (correctly
rounded sqrt) and the unfused fp32 running sum operation by operation, so the int16 maps are
required to be BIT-EXACT against the reference's golden vectors and against the oracle."""
import json
import os

import numpy as np
import pytest
from simplestereo_amd import _native      # noqa: E402  (tuning options: _native.set_option)

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "cases.json")) as _f:
    _META = json.load(_f)
_GSW = sorted(k for k, m in _META.items() if m["params"]["algo"] == "gsw")


@pytest.fixture(scope="module")
def ss():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import simplestereo_amd
    return simplestereo_amd


@pytest.mark.parametrize("cid", _GSW)
def test_gsw_vs_reference_golden_bit_exact(cid, ss, golden_cases, golden_inputs):
    maps, meta = golden_cases
    a, b = golden_inputs(meta[cid]["input"])
    p = dict(meta[cid]["params"])
    p.pop("algo")
    d = ss.passive.StereoGSW(**p).compute(a, b)
    assert d.dtype == np.int16 and d.shape == maps[cid].shape
    nm = int((d != maps[cid]).sum())
    print(cid, "mismatches", nm)
    assert nm == 0


@pytest.mark.parametrize("shape,seed,params", [
    ((37, 90), 1, dict(winSize=11, maxDisparity=70, minDisparity=0, gamma=10, fMax=120, iterations=3)),
    ((20, 64), 2, dict(winSize=3, maxDisparity=9, minDisparity=2, gamma=4, fMax=33.25, iterations=1)),
    ((25, 140), 3, dict(winSize=15, maxDisparity=130, minDisparity=5, gamma=30, fMax=500, iterations=2)),
    ((12, 33), 4, dict(winSize=21, maxDisparity=12, minDisparity=0, gamma=10, fMax=120, iterations=3)),   # window > image
    ((16, 48), 5, dict(winSize=5, maxDisparity=8, minDisparity=0, gamma=10, fMax=120, iterations=0)),      # centre-only weights
    ((9, 30), 6, dict(winSize=7, maxDisparity=3, minDisparity=5, gamma=10, fMax=120, iterations=3)),       # empty range
])
def test_gsw_vs_oracle_bit_exact(shape, seed, params, ss):
    from oracle import oracle
    from simplestereo_amd.synth import make_pair
    a, b, _ = make_pair(shape[0], shape[1], max(params["maxDisparity"], 4), seed)
    d = ss.passive.StereoGSW(**params).compute(a, b)
    ref = oracle.gsw(a, b, closed=False, **params)            # literal relaxation loops
    assert np.array_equal(d, ref)


def test_gsw_saturated_and_flat_images(ss):
    """exact ties everywhere (flat image) and maximal colour distances"""
    from oracle import oracle
    flat = np.full((14, 40, 3), 200, np.uint8)
    p = dict(winSize=5, maxDisparity=6, minDisparity=1)
    assert np.array_equal(ss.passive.StereoGSW(**p).compute(flat, flat), oracle.gsw(flat, flat, **p))
    rng = np.random.default_rng(7)
    a = (rng.integers(0, 2, (15, 44, 3)) * 255).astype(np.uint8)
    b = (rng.integers(0, 2, (15, 44, 3)) * 255).astype(np.uint8)
    p = dict(winSize=7, maxDisparity=10, minDisparity=0, gamma=3, fMax=1000)
    assert np.array_equal(ss.passive.StereoGSW(**p).compute(a, b), oracle.gsw(a, b, **p))


def test_gsw_device_tensor_path_and_strips_bit_exact(ss, golden_inputs):
    import torch
    from simplestereo_amd import strips
    a, b = golden_inputs("synth_64x96")
    m = ss.passive.StereoGSW(winSize=9, maxDisparity=24)
    full = m.compute(a, b)
    tL, tR = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert np.array_equal(m.compute(tL, tR).cpu().numpy(), full)
    H, pad = a.shape[0], m.winSize // 2
    for world in (2, 4):
        rows = []
        for rank in range(world):
            r0, r1 = strips.strip_bounds(H, world, rank)
            h0, h1 = strips.halo_bounds(H, r0, r1, pad)
            rows.append(m._compute_device(tL[h0:h1], tR[h0:h1], out_row0=r0 - h0, out_rows=r1 - r0).cpu().numpy())
        got = np.concatenate(rows, 0)
        # the left-pass border quirk of the reference depends on the ABSOLUTE image row 0
        # (_passive.cpp:445-446): strips that do not start at row 0 see y != 0, like the whole image does
        assert np.array_equal(got, full), world


def test_gsw_1080p_config4_properties(ss):
    """BASELINE config 4 (1920x1080, D 0..192, GSW defaults): determinism + LR-filled output is dense"""
    from simplestereo_amd.synth import make_pair
    a, b, gt = make_pair(1080, 1920, 192, 2)
    m = ss.passive.StereoGSW(maxDisparity=192)
    d1, d2 = m.compute(a, b), m.compute(a, b)
    assert np.array_equal(d1, d2)
    assert d1.min() >= 0 and d1.max() <= 1919          # every invalidated run was filled
    good = np.abs(d1.astype(np.int32) - gt)[:, 192:] <= 1
    print("config4 quality vs synthetic GT: %.4f" % good.mean())
    assert good.mean() > 0.6


def test_gsw_integer_sqrt_is_exact_over_whole_domain(ss):
    """the kernels' short sqrt sequence == the reference's (float)sqrt((double)s) for EVERY s it can see"""
    from simplestereo_amd import _native
    n = 3 * 255 * 255 + 1
    out = np.empty(n, np.float32)
    _native.check(_native.lib().ssamd_debug_gsw_sqrt(n, out.ctypes.data))
    want = np.sqrt(np.arange(n, dtype=np.float64)).astype(np.float32)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("geom", ["8,4,1", "8,4,2", "5,3,2", "16,2,1", "3,7,2", "8,4,2,2", "5,3,2,4", "3,7,2,2", "6,7,2,8"])
def test_gsw_forced_geometries_and_strip_heights_agree(geom, ss, golden_cases, golden_inputs):
    """one- and two-row strips, every tile shape, and 2 / 4 / 8 thread groups sharing the e tile of an image row (strips of
    4 / 8 / 16 output rows, "XG,DG,Ty,Hy") accumulate each output row's taps in the reference's raster order: forced
    shapes via the SSAMD_GSW_GEOM tuning hook reproduce the reference bit for bit (odd image heights leave a
    part-filled last strip)"""
    maps, _ = golden_cases
    a, b = golden_inputs("synth_64x96")
    m = ss.passive.StereoGSW(winSize=11, maxDisparity=24, minDisparity=0)
    base = m.compute(a, b)
    odd = m.compute(np.ascontiguousarray(a[:37]), np.ascontiguousarray(b[:37]))
    _native.set_option("SSAMD_GSW_GEOM", geom)
    try:
        got = m.compute(a, b)
        got_odd = m.compute(np.ascontiguousarray(a[:37]), np.ascontiguousarray(b[:37]))
    finally:
        _native.set_option("SSAMD_GSW_GEOM", None)
    assert np.array_equal(got, base)
    assert np.array_equal(got_odd, odd)
    assert np.array_equal(got, maps["G6d"])     # G6d = these parameters on this pair, from the reference


def test_gsw_1080p_config4_known_shift_and_strip_invariance(ss):
    """BASELINE config 4 at full size through size-independent properties: a right image that is the left image
    moved by a constant shift costs exactly 0 at d = shift in both passes wherever the window and its shifted
    copy are inside the image, so the (left-right checked) map equals the shift there; and rows matched as a
    strip carrying its halo equal the rows of the whole frame bit for bit."""
    import torch
    from simplestereo_amd.synth import make_pair
    H, W, maxd, shift, pad = 1080, 1920, 192, 141, 5
    L = make_pair(H, W, maxd, 7)[0]
    R = np.zeros_like(L)
    R[:, :W - shift] = L[:, shift:]
    R[:, W - shift:] = L[:, :shift]
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoGSW(maxDisparity=maxd)
    d = m.compute(tL, tR)
    inner = d[:, shift + pad:W - pad].cpu().numpy()
    frac = float((inner == shift).mean())
    print("GSW 1920x1080 D0..192: %.5f of the interior pixels at the known shift %d" % (frac, shift))
    assert frac >= 0.9999
    r0, rows = 500, 37
    h0, h1 = r0 - pad, r0 + rows + pad
    strip = m._compute_device(tL[h0:h1].contiguous(), tR[h0:h1].contiguous(), out_row0=r0 - h0, out_rows=rows)
    assert torch.equal(strip, d[r0:r0 + rows])


def test_gsw_autotune_changes_the_geometry_never_the_map(ss, golden_cases, golden_inputs):
    """ssamd_autotune applies to GSW as well (round 4): the first call of a shape times tiles of 2 / 4 / 8-row strips whose thread
    groups fill whole waves and caches the fastest; every geometry accumulates each output row's taps in the reference's raster
    order, so the map is the reference's whatever wins"""
    import torch
    from simplestereo_amd.synth import make_pair
    maps, _ = golden_cases
    lib = _native.lib()
    a, b = golden_inputs("synth_64x96")
    L, R, _ = make_pair(200, 700, 40, 4)
    prev = lib.ssamd_autotune(0)
    try:
        m = ss.passive.StereoGSW(winSize=11, maxDisparity=24, minDisparity=0)
        big = ss.passive.StereoGSW(winSize=11, maxDisparity=16)
        want, want_big = m.compute(a, b), big.compute(L, R)
        g0 = _native.gsw_geometry(700, 200, 11, 16, 0)
        lib.ssamd_autotune(1)
        got, got_big = m.compute(a, b), big.compute(L, R)
        again = big.compute(L, R)
        g1 = _native.gsw_geometry(700, 200, 11, 16, 0)
    finally:
        lib.ssamd_autotune(prev)
    print("model", g0, "tuned", g1)
    assert np.array_equal(got, want) and np.array_equal(got, maps["G6d"])
    assert np.array_equal(got_big, want_big) and np.array_equal(again, want_big)
