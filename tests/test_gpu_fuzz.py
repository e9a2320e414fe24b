"""GPU fuzz: random small shapes / windows / disparity ranges against the oracle, to catch
indexing errors at image borders, ragged tiles, windows larger than the image, nD not a
multiple of the register tile, single-row / single-column images, empty candidate sets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ss():
    import torch
    assert torch.cuda.is_available()
    import simplestereo_amd
    return simplestereo_amd


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        H = int(rng.integers(1, 41))
        W = int(rng.integers(1, 90))
        win = int(rng.choice([1, 3, 5, 7, 9, 11, 15, 21, 35]))
        minD = int(rng.integers(0, 6))
        maxD = int(minD + rng.integers(0, 70))
        out.append((H, W, win, minD, maxD, bool(rng.integers(0, 2)), 100 + k))
    out += [(1, 1, 1, 0, 0, True, 7), (1, 50, 9, 0, 20, True, 8), (30, 1, 9, 0, 5, True, 9), (2, 9, 35, 1, 40, False, 10),
            (17, 130, 5, 0, 129, True, 11), (8, 300, 3, 2, 290, True, 12),
            # disparity range split over several chunks (nD > 512), very large windows
            (6, 760, 5, 0, 700, True, 13), (5, 1200, 3, 10, 1100, False, 14), (40, 150, 101, 0, 30, True, 15),
            (70, 90, 69, 3, 40, False, 16)]
    return out


def _shifted_pair(H, W, minD, maxD, seed):
    """low-contrast smooth texture and a shifted, slightly noisy copy: candidates are rarely fully
    saturated (all taps at the cap 40), where the fp64 reference's own argmin is rounding noise"""
    from simplestereo_amd.synth import make_pair
    rng = np.random.default_rng(seed)
    L = make_pair(H, W, 8, seed)[0]
    L = (110 + (L.astype(np.int32) - 128) // 6).astype(np.uint8)
    d0 = int(rng.integers(minD, max(minD, min(maxD, W // 2)) + 1))
    R = np.roll(L, -d0, axis=1).astype(np.int32) + rng.integers(-2, 3, size=L.shape)
    return np.ascontiguousarray(L), np.ascontiguousarray(np.clip(R, 0, 255).astype(np.uint8))


@pytest.mark.parametrize("case", _cases(40, 2024))
def test_asw_fuzz_vs_oracle(case, ss):
    """left-referenced maps: every pixel must be within 1 level of the oracle, or be a numerical tie
    of the oracle itself (its fp64 costs at the two choices agree to 1e-6 relative)"""
    from oracle import oracle
    H, W, win, minD, maxD, _, seed = case
    a, b = _shifted_pair(H, W, minD, maxD, seed)
    p = dict(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=float(3 + seed % 9), gammaP=float(5 + seed % 20))
    d = ss.passive.StereoASW(**p).compute(a, b)
    ref, cref = oracle.asw(a, b, return_costs=True, **p)
    bad = 0
    for y, x in np.argwhere(np.abs(d.astype(np.int32) - ref) > 1):
        row = cref[y, x]
        cg, cr = row[d[y, x] - minD], row[ref[y, x] - minD]
        if not (np.isfinite(cg) and abs(cg - cr) <= 1e-6 * max(1.0, abs(cr))):
            bad += 1
    assert bad <= max(1, 0.005 * H * W), (case, bad)


@pytest.mark.parametrize("case", [c for c in _cases(40, 4048) if c[0] >= 16 and c[1] >= 32][:14])
def test_asw_consistent_fuzz_vs_oracle(case, ss):
    """LR check + fill amplify a flipped argmin into a run of pixels, so the bar is on the fraction"""
    from oracle import oracle
    H, W, win, minD, maxD, _, seed = case
    a, b = _shifted_pair(H, W, minD, maxD, seed)
    p = dict(winSize=win, maxDisparity=maxD, minDisparity=minD, gammaC=float(3 + seed % 9), gammaP=float(5 + seed % 20), consistent=True)
    d = ss.passive.StereoASW(**p).compute(a, b)
    ref = oracle.asw(a, b, **p)
    within1 = float(np.mean(np.abs(d.astype(np.int32) - ref) <= 1))
    assert within1 >= 0.98, (case, within1)


@pytest.mark.parametrize("case", _cases(30, 77))
def test_gsw_fuzz_vs_oracle_bit_exact(case, ss):
    from oracle import oracle
    from simplestereo_amd.synth import make_pair
    H, W, win, minD, maxD, _, seed = case
    if win > 21:
        win = 21
    a, b, _ = make_pair(H, W, max(maxD, 4), seed)
    p = dict(winSize=win, maxDisparity=maxD, minDisparity=minD, gamma=int(2 + seed % 30), fMax=float(20 + 7 * (seed % 40)),
             iterations=int(seed % 4))
    d = ss.passive.StereoGSW(**p).compute(a, b)
    ref = oracle.gsw(a, b, closed=True, **p)
    assert np.array_equal(d, ref), case
