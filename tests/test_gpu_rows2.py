"""GPU tier: ssamd_asw_device_rows2 -- two row ranges of a sub-image in ONE launch (the border bands of a row strip whose
interior rows were matched while the halo was in flight, simplestereo_amd/strips.py).  Rows are the reference's independent jobs
(_passive.cpp:372-374): whatever the cut, the rows written must equal those of one launch over the whole range, and the rows
in between must stay untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,params", [
    (80, 300, dict(winSize=35, maxDisparity=16)),                          # wave kernel, direct write
    (80, 300, dict(winSize=35, maxDisparity=17, consistent=True)),         # six-per-lane wave kernel, keys + left-right check per band
    (60, 400, dict(winSize=35, maxDisparity=100, minDisparity=2)),         # phase-shifted kernel
    (60, 400, dict(winSize=21, maxDisparity=150, consistent=True)),        # several tiles, keys
    (50, 120, dict(winSize=5, maxDisparity=60)),                           # round-1 kernel (window of one chunk)
    (52, 1920, dict(winSize=35, maxDisparity=192)),                        # headline tile, 16 workgroups per row: the two-range launch
                                                                           # has a full round + a part-filled one (half-width tail tiles)
])
def test_two_row_ranges_equal_one_launch(H, W, params):
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(H, W, params["maxDisparity"], 13)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(**params)
    row0, rows = 3, H - 7
    want = m._compute_device(tL, tR, out_row0=row0, out_rows=rows)
    for skip0, nskip in ((row0 + 9, rows - 20), (row0, 11), (row0 + rows - 5, 5), (row0 + 4, 0), (row0, rows)):
        out = torch.full((rows, W), -7, dtype=torch.int16, device="cuda")
        got = m._compute_device(tL, tR, out_row0=row0, out_rows=rows, out=out, skip=(skip0, nskip))
        assert got.data_ptr() == out.data_ptr()
        a, b = skip0 - row0, skip0 - row0 + nskip
        if nskip == 0:
            assert torch.equal(out, want)
            continue
        assert torch.equal(out[:a], want[:a]) and torch.equal(out[b:], want[b:]), (params, skip0, nskip)
        assert bool((out[a:b] == -7).all()), "rows between the two ranges were written"
        # ... and the interior rows written by an ordinary call into the same buffer complete the strip
        m._compute_device(tL, tR, out_row0=skip0, out_rows=nskip, out=out[a:b])
        assert torch.equal(out, want)


def test_two_row_ranges_argument_errors():
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(40, 100, 16, 2)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((30, 100), dtype=torch.int16, device="cuda")
    with pytest.raises(ValueError):
        ss.passive.StereoASW(maxDisparity=16, winSize=9)._compute_device(tL, tR, out_row0=5, out_rows=30, out=out, skip=(2, 5))
    with pytest.raises(ValueError):
        ss.passive.StereoASW(maxDisparity=16, winSize=9, alternate=True)._compute_device(tL, tR, out_row0=5, out_rows=30, out=out, skip=(8, 5))
    # (round 6: exact=True has a two-range form, ssamd_asw_exact_device_rows2 -- covered by the parametrised test above with the default exact="auto")
    with pytest.raises(ValueError):
        ss.passive.StereoASW(maxDisparity=16, winSize=9)._compute_device(tL, tR, out_row0=5, out_rows=30, out=out[:, :50])


@pytest.mark.parametrize("H,W,params", [
    (288, 384, dict(winSize=15, maxDisparity=16)),                         # Tsukuba-size: the call the shared launch is for
    (70, 333, dict(winSize=35, maxDisparity=17, consistent=True)),         # six per lane (8-byte e slots)
    (50, 401, dict(winSize=35, maxDisparity=100, minDisparity=3)),         # phase-shifted kernel, a width that is no multiple of 4
    (40, 260, dict(winSize=21, maxDisparity=300)),                         # a range the frame is too narrow for (round-1 kernel or chunks)
    (48, 900, dict(winSize=35, maxDisparity=400)),                         # two disparity chunks of the volume
])
def test_lab_records_and_tad_volume_in_one_launch_equal_two(H, W, params):
    """round 5: K0 (Lab records) and K0e (TAD volume, now read from the image bytes) share a launch when a call goes straight to
    its cached geometry; SSAMD_ASW_PREPASS_FUSE=0 restores the two dependent launches of rounds 2-4 -- same maps, same raw costs"""
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd import _native
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(H, W, min(params["maxDisparity"], W // 2), 17)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(**params)
    m.compute(tL, tR)                                   # (a first call of a shape may run the tuner's trial launches)
    lib = _native.lib()
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    fused = m.compute(tL, tR)
    torch.cuda.synchronize()
    _, n_fused = _native.profile_read()
    with _native.options(SSAMD_ASW_PREPASS_FUSE="0"):
        lib.ssamd_profile_reset()
        plain = m.compute(tL, tR)
        torch.cuda.synchronize()
        _, n_plain = _native.profile_read()
        sub_plain = m._compute_device(tL, tR, out_row0=5, out_rows=H - 11)
    lib.ssamd_profile_enable(0)
    assert torch.equal(fused, plain)
    # (a call without the TAD volume -- e.g. the round-1 kernel of the last case -- has the records as its only pre-pass launch either way)
    uses_volume = bool(_native.asw_kernel_form(W, H, params["winSize"], params["maxDisparity"], params.get("minDisparity", 0))["phase_shifted"] or
                       _native.asw_kernel_form(W, H, params["winSize"], params["maxDisparity"], params.get("minDisparity", 0))["wave_kernel"])
    assert n_fused[_native.K_LAB] == 1 and n_plain[_native.K_LAB] == (2 if uses_volume else 1), (n_fused, n_plain, uses_volume)
    # a row range of the sub-image (strips): records and volume of rows [row0 - pad, row0 + rows + pad) only
    assert torch.equal(m._compute_device(tL, tR, out_row0=5, out_rows=H - 11), sub_plain)
    assert torch.equal(sub_plain, fused[5:H - 6])
    # host arrays go through the same path
    assert np.array_equal(m.compute(L, R), fused.cpu().numpy())


@pytest.mark.parametrize("H,W,params", [
    (60, 300, dict(winSize=11, maxDisparity=16)),                          # class default StereoGSW()
    (48, 500, dict(winSize=11, maxDisparity=100, minDisparity=2)),         # two disparity chunks
    (40, 200, dict(winSize=21, maxDisparity=40, gamma=7)),
])
def test_gsw_two_row_ranges_equal_one_launch(H, W, params):
    """ssamd_gsw_device_rows2 (round 6: the overlapped strip step for StereoGSW): the two bands equal the rows of one launch over
    the whole range -- GSW is bit-exact, so equal means equal to the reference's rows -- and the rows in between stay untouched"""
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd.synth import make_pair
    L, R, _ = make_pair(H, W, params["maxDisparity"], 17)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoGSW(**params)
    row0, rows = 2, H - 5
    want = m._compute_device(tL, tR, out_row0=row0, out_rows=rows)
    for skip0, nskip in ((row0 + 6, rows - 13), (row0, 9), (row0 + rows - 4, 4), (row0 + 3, 0), (row0, rows)):
        out = torch.full((rows, W), -7, dtype=torch.int16, device="cuda")
        got = m._compute_device(tL, tR, out_row0=row0, out_rows=rows, out=out, skip=(skip0, nskip))
        assert got.data_ptr() == out.data_ptr()
        a, b = skip0 - row0, skip0 - row0 + nskip
        if nskip == 0:
            assert torch.equal(out, want)
            continue
        assert torch.equal(out[:a], want[:a]) and torch.equal(out[b:], want[b:]), (params, skip0, nskip)
        assert bool((out[a:b] == -7).all()), "rows between the two ranges were written"
    with pytest.raises(ValueError):
        m._compute_device(tL, tR, out_row0=row0, out_rows=rows, out=torch.empty((rows, W), dtype=torch.int16, device="cuda"), skip=(0, 3))
