"""GPU tier: the opt-in "alternate pixel" ASW mode (a docstring todo in the reference,
passive.py:43-46, without code -- so there is no reference output and this mode is checked against our
own CPU restatement built from the reference-exact pieces, oracle.asw_alternate, and by its properties)."""
import numpy as np
import pytest
from simplestereo_amd import _native      # noqa: E402  (tuning options: _native.set_option)

pytestmark = pytest.mark.gpu

WITHIN1, EXACT = 0.995, 0.99


@pytest.fixture(scope="module")
def ss():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import simplestereo_amd
    return simplestereo_amd


def _pairs(golden_inputs):
    a, b = golden_inputs("tsukuba")
    yield "tsukuba_rows", np.ascontiguousarray(a[60:131]), np.ascontiguousarray(b[60:131]), dict(winSize=15, maxDisparity=16)
    yield "tsukuba_min4", np.ascontiguousarray(a[100:140]), np.ascontiguousarray(b[100:140]), dict(winSize=21, maxDisparity=14, minDisparity=4, gammaC=15.0)
    a, b = golden_inputs("synth_64x96")
    yield "synth", a, b, dict(winSize=9, maxDisparity=24, minDisparity=2, gammaC=7.5, gammaP=36.0)
    a, b = golden_inputs("crop")
    yield "crop_bigwin", a, b, dict(winSize=35, maxDisparity=8)


def test_even_rows_are_the_exact_mode_and_odd_rows_match_the_restatement(ss, golden_inputs):
    from oracle import oracle
    for name, a, b, p in _pairs(golden_inputs):
        exact = ss.passive.StereoASW(exact=False, **p).compute(a, b)      # (the rows the alternate mode matches fully are fp32 argmins)
        alt = ss.passive.StereoASW(alternate=True, **p).compute(a, b)
        assert alt.dtype == np.int16 and alt.shape == exact.shape
        assert np.array_equal(alt[::2], exact[::2]), name            # same kernel, same rows
        # fill stage in isolation: the restatement starts from the GPU's own exact rows
        want, evaluated = oracle.asw_alternate(a, b, exact_rows=exact, **p)
        diff = np.abs(alt[1::2].astype(np.int32) - want[1::2])
        print(name, "evaluated pixels", evaluated, "odd rows exact %.5f within1 %.5f" % ((diff == 0).mean(), (diff <= 1).mean()))
        assert evaluated > 0
        assert (diff <= 1).mean() >= WITHIN1 and (diff == 0).mean() >= EXACT, name
        # end to end against the fp64 restatement
        want2, _ = oracle.asw_alternate(a, b, **p)
        d2 = np.abs(alt.astype(np.int32) - want2)
        assert (d2 <= 1).mean() >= WITHIN1, name
        # the bounded search never leaves the interval spanned by its two exact neighbours
        lo = np.minimum(alt[0:-2:2], alt[2::2]); hi = np.maximum(alt[0:-2:2], alt[2::2])
        mid = alt[1:-1:2][:lo.shape[0]]
        x = np.arange(alt.shape[1])[None, :]
        valid = x >= p.get("minDisparity", 0)
        lo_c = np.clip(lo, p.get("minDisparity", 0), np.minimum(p["maxDisparity"], x))
        hi_c = np.clip(hi, p.get("minDisparity", 0), np.minimum(p["maxDisparity"], x))
        assert np.all(((mid >= lo_c) & (mid <= hi_c))[np.broadcast_to(valid, mid.shape)]), name


def test_alternate_device_tensors_edges_and_errors(ss, golden_inputs):
    import torch
    a, b = golden_inputs("crop")
    p = dict(winSize=7, maxDisparity=6, minDisparity=1)
    m = ss.passive.StereoASW(alternate=True, **p)
    host = m.compute(a, b)
    dev = m.compute(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert dev.dtype == torch.int16 and np.array_equal(dev.cpu().numpy(), host)
    exact = ss.passive.StereoASW(exact=False, **p)
    for rows in (1, 2, 3):                            # no odd row / last odd row without a row below
        aa, bb = np.ascontiguousarray(a[:rows]), np.ascontiguousarray(b[:rows])
        got, ex = m.compute(aa, bb), exact.compute(aa, bb)
        assert np.array_equal(got[::2], ex[::2])
        if rows == 2:                                 # row 1 has only the neighbour above: copied
            lo = np.clip(ex[0], 1, np.minimum(6, np.arange(a.shape[1])))
            assert np.array_equal(got[1][1:], lo[1:])
    # empty candidate ranges give x, as in the exact mode (_passive.cpp:54,98)
    e = ss.passive.StereoASW(winSize=5, maxDisparity=3, minDisparity=7, alternate=True).compute(a, b)
    assert np.array_equal(e, np.tile(np.arange(a.shape[1], dtype=np.int16), (a.shape[0], 1)))
    # row ranges of the whole image and of sub-images with a halo of winSize/2 + 1 rows (round 2): the rows of the whole map
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    whole = m._compute_device(ta, tb)
    H = a.shape[0]
    pad = int(m.winSize) // 2 + 1
    for r0, rows in ((2, 6), (3, 6), (3, 7), (0, 5), (H - 5, 5), (H - 4, 4), (1, 1), (7, 2)):
        assert torch.equal(m._compute_device(ta, tb, out_row0=r0, out_rows=rows), whole[r0:r0 + rows]), (r0, rows)
        h0, h1 = max(0, r0 - pad), min(H, r0 + rows + pad)
        sub = m._compute_device(ta[h0:h1].contiguous(), tb[h0:h1].contiguous(), out_row0=r0 - h0, out_rows=rows, row_parity=h0 & 1)
        assert torch.equal(sub, whole[r0:r0 + rows]), (r0, rows, h0)
    with pytest.raises(ValueError):              # an odd first row needs the exact row above it inside the sub-image
        m._compute_device(ta[1:].contiguous(), tb[1:].contiguous(), out_row0=0, out_rows=4, row_parity=1)


def test_alternate_quality_on_tsukuba_is_not_worse(ss, golden_inputs):
    """the reference's note: 'no significant decrease in quality' -- bad-1.0 over the non-occluded pixels"""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tsukuba_pair.npz"))
    a, b, gt, nonocc = z["left"], z["right"], z["groundtruth"], z["nonocc"]
    g = gt.astype(np.float64) / 16.0 if gt.max() > 64 else gt.astype(np.float64)
    mask = nonocc > 0
    bad = {}
    for alt in (False, True):
        d = ss.passive.StereoASW(winSize=35, maxDisparity=16, alternate=alt).compute(a, b)
        bad[alt] = 100.0 * float(np.mean(np.abs(d[mask] - g[mask]) > 1.0))
    print("Tsukuba win 35 bad-1.0: exact %.3f %%, alternate %.3f %%" % (bad[False], bad[True]))
    assert bad[True] <= bad[False] + 0.5


def test_alternate_full_queue_falls_back_to_in_place_evaluation(ss, golden_inputs):
    """a job queue that is too small (forced through the SSAMD_ALT_QUEUE_CAP test hook) must not lose work:
    waves that find it full evaluate their pixels themselves, with the same result"""
    import os
    a, b = golden_inputs("tsukuba")
    a, b = np.ascontiguousarray(a[60:131]), np.ascontiguousarray(b[60:131])
    m = ss.passive.StereoASW(winSize=15, maxDisparity=16, alternate=True)
    want = m.compute(a, b)
    for cap in ("1", "37", "300"):
        _native.set_option("SSAMD_ALT_QUEUE_CAP", cap)
        try:
            got = m.compute(a, b)
        finally:
            _native.set_option("SSAMD_ALT_QUEUE_CAP", None)
        assert np.array_equal(got, want), cap


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = [(1, 9, 3, 0, 4, 1), (2, 1, 5, 0, 3, 2), (3, 70, 35, 2, 40, 3), (9, 300, 3, 0, 280, 4)]
    for k in range(n):
        out.append((int(rng.integers(2, 34)), int(rng.integers(1, 100)), int(rng.choice([1, 3, 5, 9, 15, 21])),
                    int(rng.integers(0, 5)), int(rng.integers(0, 60)), 50 + k))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(14, 99))
def test_alternate_fuzz_vs_restatement(case, ss):
    """random small shapes: even rows == exact mode; odd rows == the CPU rule applied to the GPU's own exact rows,
    up to numerical ties of the fp64 costs (1e-6 relative)"""
    from oracle import oracle
    from simplestereo_amd.synth import make_pair
    H, W, win, minD, span, seed = case
    maxD = minD + span
    a = make_pair(H, W, 8, seed)[0]
    a = (100 + (a.astype(np.int32) - 128) // 5).astype(np.uint8)
    rng = np.random.default_rng(seed)
    shift = np.repeat(rng.integers(minD, max(minD, min(maxD, W // 2)) + 1, size=(H + 7) // 8), 8)[:H]   # depth steps between rows
    b = np.stack([np.roll(a[y], -int(shift[y]), axis=0) for y in range(H)]).astype(np.int32) + rng.integers(-2, 3, size=a.shape)
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(np.clip(b, 0, 255).astype(np.uint8))
    p = dict(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=6.0, gammaP=15.0)
    exact = ss.passive.StereoASW(exact=False, **p).compute(a, b)      # (the rows the alternate mode matches fully are fp32 argmins)
    alt = ss.passive.StereoASW(alternate=True, **p).compute(a, b)
    assert np.array_equal(alt[::2], exact[::2])
    if H < 2:
        return
    want, _ = oracle.asw_alternate(a, b, exact_rows=exact, **p)
    _, costs = oracle.asw(a, b, return_costs=True, **p) if maxD >= minD else (None, None)
    bad = 0
    for y, x in np.argwhere(alt != want):
        cg, cw = costs[y, x, alt[y, x] - minD], costs[y, x, want[y, x] - minD]
        if not (np.isfinite(cg) and abs(cg - cw) <= 1e-6 * max(1.0, abs(cw))):
            bad += 1
    assert bad == 0, (case, bad)


def test_alternate_with_several_disparity_chunks_goes_through_the_key_path(ss, golden_inputs):
    """a forced small tile splits the disparity range over several workgroups: the even rows are then merged
    through the WTA keys and decoded, instead of being written by the aggregation kernel -- same map"""
    import os
    a, b = golden_inputs("synth_96x128")
    m = ss.passive.StereoASW(winSize=21, maxDisparity=39, alternate=True)
    want = m.compute(a, b)
    _native.set_option("SSAMD_ASW_GEOM", "6,5,8")          # Dc = 20: two chunks for 40 disparities, chunked tap staging
    try:
        got = m.compute(a, b)
    finally:
        _native.set_option("SSAMD_ASW_GEOM", None)
    assert np.array_equal(got, want)


def test_alternate_with_consistency_check_on_the_exact_rows(ss, golden_inputs):
    """consistent=True + alternate=True: the even rows are the consistent mode's rows (right-referenced pass,
    left-right check, occlusion filling), the odd rows are derived from them by the bounded search"""
    from oracle import oracle
    a, b = golden_inputs("tsukuba")
    a, b = np.ascontiguousarray(a[40:111]), np.ascontiguousarray(b[40:111])
    p = dict(winSize=15, maxDisparity=16, consistent=True)
    exact = ss.passive.StereoASW(exact=False, **p).compute(a, b)      # (the rows the alternate mode matches fully are fp32 argmins)
    alt = ss.passive.StereoASW(alternate=True, **p).compute(a, b)
    assert np.array_equal(alt[::2], exact[::2])
    want, evaluated = oracle.asw_alternate(a, b, exact_rows=exact, **p)
    diff = np.abs(alt[1::2].astype(np.int32) - want[1::2])
    print("consistent+alternate: evaluated", evaluated, "odd rows exact %.5f" % (diff == 0).mean())
    assert evaluated > 0 and (diff <= 1).mean() >= WITHIN1 and (diff == 0).mean() >= EXACT
    want2, _ = oracle.asw_alternate(a, b, **p)
    assert (np.abs(alt.astype(np.int32) - want2) <= 1).mean() >= 0.99     # LR fill amplifies rare argmin ties into runs
