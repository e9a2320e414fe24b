"""GPU: the kernels either side of the matchers -- rectification remap and
disparity -> 3-D reprojection -- against the per-pixel oracle restatement (oracle/rig_oracle.py)
and against the numpy host path of the rig class; then the whole chain on device tensors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RIGRECT = os.path.join(G, "rig_example2_rigRect.json")


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import simplestereo_amd as ss
    rig = ss.RectifiedStereoRig.fromFile(RIGRECT)
    return ss, torch, rig


def test_remap_kernel_vs_oracle_and_host_path(env):
    ss, torch, rig = env
    from oracle import rig_oracle
    from simplestereo_amd import _rigs
    rig.computeRectificationMaps(destDims=(96, 54))
    w, h = rig.res1
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    b = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for interp in (1, 0):
        r1, r2 = rig.rectifyImages(ta, tb, interpolation=interp)
        assert r1.is_cuda and r1.dtype == torch.uint8 and tuple(r1.shape) == (54, 96, 3)
        h1, h2 = rig.rectifyImages(a, b, interpolation=interp)
        assert np.array_equal(r1.cpu().numpy(), h1) and np.array_equal(r2.cpu().numpy(), h2)
        o1 = rig_oracle.remap_bilinear(a, rig.mapx1, rig.mapy1, nearest=(interp == 0))
        assert np.array_equal(r1.cpu().numpy(), o1)
    # maps pointing far outside the image: constant border 0, also for negative coordinates
    mx = (np.arange(12, dtype=np.float32)[None, :] * 3.7 - 20).repeat(5, 0)
    my = (np.arange(5, dtype=np.float32)[:, None] * 2.3 - 4).repeat(12, 1)
    small = rng.integers(0, 256, (6, 9, 3)).astype(np.uint8)
    out = torch.empty((5, 12, 3), dtype=torch.uint8, device="cuda")
    import ctypes
    from simplestereo_amd import _native
    ts, tmx, tmy = torch.from_numpy(small).cuda(), torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda()
    _native.check(_native.lib().ssamd_remap_bgr_device(ts.data_ptr(), 6, 9, tmx.data_ptr(), tmy.data_ptr(), 5, 12, 1,
                                                       out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), rig_oracle.remap_bilinear(small, mx, my))
    assert np.array_equal(out.cpu().numpy(), _rigs._remap(small, mx, my))


def test_remap_kernel_rounds_ties_up_like_opencv_fixed_point(env):
    """(sum + 16384) >> 15 of cv2.remap's 8-bit bilinear path: fx = 16/32 between pixels 2 and 3 gives 3"""
    ss, torch, rig = env
    import ctypes
    from oracle import rig_oracle
    from simplestereo_amd import _native, _rigs
    img = np.zeros((4, 6, 3), np.uint8)
    img[:, :, 0] = [0, 2, 3, 7, 8, 255]
    img[:, :, 1] = np.arange(6)[None, :] * 40
    img[:, :, 2] = 255 - img[:, :, 0]
    img[2:] //= 2
    fy, fx = np.mgrid[0:32, 0:32]
    mx = np.concatenate([(1 + fx / 32.0), np.full((1, 32), 1.5)], 0).astype(np.float32)
    my = np.concatenate([(1 + fy / 32.0), np.zeros((1, 32))], 0).astype(np.float32)
    out = torch.empty((33, 32, 3), dtype=torch.uint8, device="cuda")
    ts, tmx, tmy = torch.from_numpy(img).cuda(), torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda()
    _native.check(_native.lib().ssamd_remap_bgr_device(ts.data_ptr(), 4, 6, tmx.data_ptr(), tmy.data_ptr(), 33, 32, 1,
                                                       out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got[32, 0, 0] == 3                        # (2 + 3) / 2 = 2.5 -> 3, not 2
    assert np.array_equal(got, rig_oracle.remap_bilinear(img, mx, my))
    assert np.array_equal(got, _rigs._remap(img, mx, my))


@pytest.mark.parametrize("shape", [(7, 9, 5, 11), (33, 21, 40, 37), (1, 1, 3, 2), (16, 16, 16, 16)])
def test_remap_kernel_odd_sizes_borders_and_last_pixels(shape, env):
    """outputs whose pixel count is not a multiple of the four pixels a thread owns, maps that point outside the source
    on every side and exactly at its last pixels (the 8-byte source reads must not leave the buffer): equal to the oracle"""
    ss, torch, rig = env
    import ctypes
    from oracle import rig_oracle
    from simplestereo_amd import _native
    hs, ws, hd, wd = shape
    rng = np.random.default_rng(hs * 100 + wd)
    img = rng.integers(0, 256, (hs, ws, 3)).astype(np.uint8)
    mx = rng.uniform(-2.5, ws + 1.5, (hd, wd)).astype(np.float32)
    my = rng.uniform(-2.5, hs + 1.5, (hd, wd)).astype(np.float32)
    mx.flat[0], my.flat[0] = ws - 1, hs - 1                      # the very last source pixel
    mx.flat[-1], my.flat[-1] = ws - 1.5, hs - 1.5                # the 2 x 2 block that ends the buffer
    if mx.size > 2:
        mx.flat[1], my.flat[1] = ws - 2.0, hs - 1.0              # last row, last pixel pair
    t = torch.from_numpy(img).cuda()
    dmx, dmy = torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda()
    for interp in (1, 0):
        out = torch.empty((hd, wd, 3), dtype=torch.uint8, device="cuda")
        _native.check(_native.lib().ssamd_remap_bgr_device(t.data_ptr(), hs, ws, dmx.data_ptr(), dmy.data_ptr(), hd, wd, interp,
                                                           out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        want = rig_oracle.remap_bilinear(img, mx, my, nearest=(interp == 0))
        assert np.array_equal(out.cpu().numpy(), want), (shape, interp)


@pytest.mark.parametrize("wh", [(64, 36), (37, 21), (1, 3), (260, 2)])
def test_reproject_kernel_vs_oracle_and_host_path(wh, env):
    """widths that are and are not a multiple of the four pixels a thread owns"""
    ss, torch, rig = env
    from oracle import rig_oracle
    w, h = wh
    rig.computeRectificationMaps(destDims=(w, h))
    rng = np.random.default_rng(5)
    d = rng.integers(1, 60, (h, w)).astype(np.int16)
    pts = rig.get3DPoints(torch.from_numpy(d).cuda())
    assert pts.is_cuda and pts.dtype == torch.float32 and tuple(pts.shape) == (h, w, 3)
    host = rig.get3DPoints(d)
    ref = rig_oracle.reproject(d, rig.getQ())
    g = pts.cpu().numpy()
    assert np.allclose(g, ref, rtol=2e-6, atol=1e-6)
    assert np.allclose(g, host, rtol=2e-6, atol=1e-6)


def test_device_chain_rectify_match_reproject(env):
    """camera frames -> rectified pair -> disparity -> 3-D points, all resident in HBM"""
    ss, torch, rig = env
    rig.computeRectificationMaps(destDims=(160, 90))
    w, h = rig.res1
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(h, w, 40, 11)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    rL, rR = rig.rectifyImages(tL, tR)
    m = ss.passive.StereoASW(winSize=9, maxDisparity=24, consistent=True)
    disp = m.compute(rL, rR)
    pts = rig.get3DPoints(disp)
    assert disp.is_cuda and pts.is_cuda and tuple(pts.shape) == (90, 160, 3)
    # same chain through host arrays
    hL, hR = rig.rectifyImages(L, R)
    hd = m.compute(hL, hR)
    assert np.array_equal(disp.cpu().numpy(), hd)
    assert np.allclose(pts.cpu().numpy(), rig.get3DPoints(hd), rtol=2e-6, atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("dest,params,interp", [((320, 180), dict(winSize=35, maxDisparity=25, minDisparity=4, gammaC=15, consistent=True), 1),
                                                ((640, 360), dict(winSize=35, maxDisparity=100, minDisparity=4, gammaC=15), 1),
                                                ((161, 91), dict(winSize=11, maxDisparity=30), 0)])
def test_rectify_and_match_in_one_call(dest, params, interp, env):
    """compute(raw1, raw2, rectify=rig): rectification + Lab records as ONE launch (remap_lab_records_pair_kernel), the rectified
    frames never in HBM -- the same map, bit for bit, as rectifyImages followed by compute (the reference's pipeline of
    examples/009: quarter-size destination with the example's parameters; a destination large enough for the phase-shifted
    kernel; odd sizes with nearest-neighbour maps)"""
    ss, torch, rig = env
    rig.computeRectificationMaps(destDims=dest)
    w, h = rig.res1
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(h, w, 120, 5)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(**params)
    two_step = m.compute(*rig.rectifyImages(tL, tR, interpolation=interp))
    fused = m.compute(tL, tR, rectify=rig, interpolation=interp)
    assert fused.is_cuda and fused.dtype == torch.int16 and tuple(fused.shape) == (dest[1], dest[0])
    assert torch.equal(fused, two_step)
    g = ss.passive.StereoGSW(winSize=min(params["winSize"], 11), maxDisparity=params["maxDisparity"], minDisparity=params.get("minDisparity", 0))
    assert torch.equal(g.compute(tL, tR, rectify=rig, interpolation=interp), g.compute(*rig.rectifyImages(tL, tR, interpolation=interp)))
    with pytest.raises(ValueError):
        m.compute(L, R, rectify=rig)                      # host arrays: the two-step path is the one to use
    with pytest.raises(ValueError):
        ss.passive.StereoASW(alternate=True, **params).compute(tL, tR, rectify=rig)
