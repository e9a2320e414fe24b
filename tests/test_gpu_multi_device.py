"""GPU tier: the `devices=[...]` extension of StereoASW / StereoGSW.compute (ssamd_asw_multi / ssamd_gsw_multi,
include/ssamd.h): the frame is cut into one row strip per listed GPU, matched concurrently from one process and
reassembled.  Rows are the reference's independent jobs (_passive.cpp:372-374), so the map must equal the one-GPU
map bit for bit.  A 1-GPU box exercises the strip cut through the test hook SSAMD_MULTI_ALLOW_REPEAT (the same device
listed several times: the strips queue on that device's mutex); with >= 2 GPUs the real thing runs."""
import os
import threading

import numpy as np
import pytest
from simplestereo_amd import _native      # noqa: E402  (tuning options: _native.set_option)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def allow_repeat():
    _native.set_option("SSAMD_MULTI_ALLOW_REPEAT", "1")
    yield
    _native.set_option("SSAMD_MULTI_ALLOW_REPEAT", None)


def _devices(n):
    import torch
    have = torch.cuda.device_count()
    return [k % have for k in range(n)]


@pytest.mark.parametrize("n", [1, 2, 3, 5])
@pytest.mark.parametrize("consistent", [False, True])
def test_asw_devices_equal_one_gpu(n, consistent, allow_repeat, golden_inputs):
    import simplestereo_amd as ss
    a, b = golden_inputs("synth_96x128")
    m = ss.passive.StereoASW(winSize=21, maxDisparity=32, consistent=consistent)
    want = m.compute(a, b)
    got = m.compute(a, b, devices=_devices(n))
    assert got.dtype == np.int16 and np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 4])
def test_gsw_devices_equal_one_gpu(n, allow_repeat, golden_inputs):
    import simplestereo_amd as ss
    a, b = golden_inputs("synth_96x128")
    m = ss.passive.StereoGSW(winSize=9, maxDisparity=24)
    assert np.array_equal(m.compute(a, b, devices=_devices(n)), m.compute(a, b))


def test_more_devices_than_rows_and_strips_thinner_than_the_halo(allow_repeat, golden_inputs):
    import simplestereo_amd as ss
    a, b = golden_inputs("crop")                       # 18 rows
    m = ss.passive.StereoASW(winSize=35, maxDisparity=8)
    want = m.compute(a, b)
    assert np.array_equal(m.compute(a, b, devices=_devices(7)), want)       # strips of 3 rows under a 17-row halo
    a3, b3 = np.ascontiguousarray(a[:3]), np.ascontiguousarray(b[:3])
    assert np.array_equal(m.compute(a3, b3, devices=_devices(5)), m.compute(a3, b3))   # 5 devices, 3 rows


def test_devices_argument_errors(golden_inputs):
    import simplestereo_amd as ss
    import torch
    a, b = golden_inputs("crop")
    m = ss.passive.StereoASW(winSize=7, maxDisparity=6)
    with pytest.raises(ValueError, match="listed twice"):
        m.compute(a, b, devices=[0, 0])
    with pytest.raises(ValueError, match="out of range"):
        m.compute(a, b, devices=[0, torch.cuda.device_count() + 3])
    with pytest.raises(ValueError):
        m.compute(a, b, devices=[])
    with pytest.raises(ValueError):
        m.compute(a, b, devices=[-1])
    alt = ss.passive.StereoASW(winSize=7, maxDisparity=6, alternate=True)
    assert np.array_equal(alt.compute(a, b, devices=[0]), alt.compute(a, b))      # (round 2: the alternate-rows mode takes strips)
    with pytest.raises(ValueError, match="host arrays"):
        m.compute(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), devices=[0])


def test_entry_points_leave_the_callers_device_alone(golden_inputs):
    """an operator asked to run on device k must not leave the calling thread on device k"""
    import torch
    import simplestereo_amd as ss
    from simplestereo_amd import _native
    a, b = golden_inputs("crop")
    n = torch.cuda.device_count()
    before = torch.cuda.current_device()
    for dev in range(n):
        ss.passive.StereoASW(winSize=7, maxDisparity=6, device=dev).compute(a, b)
        ss.passive.StereoGSW(winSize=5, maxDisparity=6, device=dev).compute(a, b)
        assert torch.cuda.current_device() == before
    _native.lib().ssamd_profile_reset()
    _native.profile_read()
    assert torch.cuda.current_device() == before
    if n >= 2:      # the raw HIP current device of this thread, not torch's cached idea of it
        torch.cuda.set_device(1)
        ss.passive.StereoASW(winSize=7, maxDisparity=6, device=0).compute(a, b)
        x = torch.zeros(4, device="cuda")
        assert x.device.index == 1
        torch.cuda.set_device(before)


def test_parameter_tables_are_cached_per_parameter_set(golden_inputs):
    """alternating matchers (different winSize / gammaP / gamma) on one device: results stay those of each
    matcher alone -- the proximity and GSW tables are keyed by their parameters (more sets than cache slots)"""
    import simplestereo_amd as ss
    a, b = golden_inputs("crop")
    asw = [ss.passive.StereoASW(winSize=w, maxDisparity=6, gammaP=gp) for w in (5, 7, 9) for gp in (10.0, 17.5, 30.0, 44.0)]
    gsw = [ss.passive.StereoGSW(winSize=5, maxDisparity=6, gamma=g) for g in (3, 5, 8, 10, 14, 20)]
    want_a = [m.compute(a, b) for m in asw]
    want_g = [m.compute(a, b) for m in gsw]
    for _ in range(3):
        for m, w in zip(asw, want_a):
            assert np.array_equal(m.compute(a, b), w)
        for m, w in zip(gsw, want_g):
            assert np.array_equal(m.compute(a, b), w)


def test_threads_on_one_device_still_serialise_correctly(golden_inputs):
    import simplestereo_amd as ss
    a, b = golden_inputs("synth_64x96")
    m = ss.passive.StereoASW(winSize=9, maxDisparity=24)
    want = m.compute(a, b)
    errs = []

    def work():
        for _ in range(5):
            if not np.array_equal(ss.passive.StereoASW(winSize=9, maxDisparity=24).compute(a, b, devices=[0]), want):
                errs.append("mismatch")
    ts = [threading.Thread(target=work) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs



@pytest.mark.parametrize("n", [2, 3, 5])
def test_alternate_rows_mode_across_strips(n, allow_repeat, golden_inputs):
    """StereoASW(alternate=True) cut into row strips (halo of winSize/2 + 1 rows, even rows of the WHOLE image exact
    whatever row a strip starts at) gives the whole-image map bit for bit"""
    import simplestereo_amd as ss
    a, b = golden_inputs("synth_96x128")
    for rows in (96, 95, 7):
        aa, bb = np.ascontiguousarray(a[:rows]), np.ascontiguousarray(b[:rows])
        m = ss.passive.StereoASW(winSize=9, maxDisparity=20, minDisparity=1, alternate=True)
        want = m.compute(aa, bb)
        assert np.array_equal(m.compute(aa, bb, devices=_devices(n)), want), (n, rows)
