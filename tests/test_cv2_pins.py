"""Pins of the OpenCV-backed plumbing that can only be checked where cv2 exists.

The reference computes its rectification maps, the remap and the 3-D reprojection with OpenCV
(reference simplestereo/_rigs.py:540-541, 564-565, 628).  cv2 is not installed in the build container nor on the GPU
boxes, so there these tests SKIP and f1 / f2 stay "parity unpinned" (DESIGN.md section 7); on any machine with cv2
they compare `simplestereo_amd._rigs` -- the numpy restatement the GPU kernels are tested against -- with cv2 itself,
and with tests/golden/cv2_pins.npz when tests/golden/make_golden_cv2.py has been run there."""
import os
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def _pins():
    path = os.path.join(GOLDEN, "cv2_pins.npz")
    if os.path.exists(path):
        return dict(np.load(path))
    pytest.importorskip("cv2", reason="cv2 absent and tests/golden/cv2_pins.npz not generated: f1/f2 stay unpinned here")
    import make_golden_cv2
    return make_golden_cv2.with_cv2()


def test_rectification_maps_match_initUndistortRectifyMap():
    pins = _pins()
    import make_golden_cv2
    rig, _, _, _ = make_golden_cv2.inputs()
    for k, (mx, my) in {"1": (rig.mapx1, rig.mapy1), "2": (rig.mapx2, rig.mapy2)}.items():
        # float32 maps of a double pipeline: agreement to a few ulps of a pixel coordinate ~1e3
        assert np.abs(mx - pins["mapx" + k]).max() < 2e-3 and np.abs(my - pins["mapy" + k]).max() < 2e-3


def test_remap_matches_cv2_remap_on_cv2s_own_maps():
    pins = _pins()
    import make_golden_cv2
    from simplestereo_amd import _rigs
    _, img, img16, _ = make_golden_cv2.inputs()
    for k in ("1", "2"):
        mx, my = pins["mapx" + k], pins["mapy" + k]
        assert np.array_equal(_rigs._remap(img, mx, my, _rigs.INTER_LINEAR), pins["remap_linear" + k])       # fixed point: exact
        assert np.array_equal(_rigs._remap(img, mx, my, _rigs.INTER_NEAREST), pins["remap_nearest" + k])
        d16 = np.abs(_rigs._remap(img16, mx, my, _rigs.INTER_LINEAR).astype(np.int64) - pins["remap_linear_u16_" + k].astype(np.int64))
        assert d16.max() <= 1                                                                                  # float path: last-bit


def test_get3DPoints_matches_reprojectImageTo3D():
    pins = _pins()
    import make_golden_cv2
    rig, _, _, disp = make_golden_cv2.inputs()
    got = rig.get3DPoints(disp)
    want = pins["points"]
    ok = np.isfinite(want).all(-1)
    assert np.allclose(got[ok], want[ok], rtol=2e-6, atol=1e-6)
