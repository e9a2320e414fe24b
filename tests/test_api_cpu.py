"""CPU tier: the Python boundary (signatures, defaults, exceptions of passive.py /
_passive.cpp), the C-ABI library loads and exports every symbol of include/ssamd.h,
and the product refuses to run without a GPU (no CPU fallback)."""
import inspect
import os
import subprocess
import sys
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ss():
    from simplestereo_amd.build import build_native
    build_native()
    import simplestereo_amd
    return simplestereo_amd


def test_signatures_match_reference(ss):
    """passive.py:59 and passive.py:133-134 of the reference; our only additions are trailing keywords
    (`device=None`, and `alternate=False`, `exact="auto"` for ASW), so every positional / keyword call of the reference
    binds identically"""
    sig = inspect.signature(ss.passive.StereoASW.__init__)
    assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
        ("winSize", 35), ("maxDisparity", 16), ("minDisparity", 0), ("gammaC", 5), ("gammaP", 17.5),
        ("consistent", False), ("device", None), ("alternate", False), ("exact", "auto")]
    sig = inspect.signature(ss.passive.StereoGSW.__init__)
    assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
        ("winSize", 11), ("maxDisparity", 16), ("minDisparity", 0), ("gamma", 10), ("fMax", 120),
        ("iterations", 3), ("bins", 20), ("device", None)]
    assert ss.passive.StereoASW()._alternate(0) is False and ss.passive.StereoASW(alternate=True)._alternate(0) is True
    # exact="auto" (round 6): the reference's fp64 argmin by default, off only where it has no form (alternate rows)
    assert ss.passive.StereoASW()._exact() is True and ss.passive.StereoASW(exact=True)._exact() is True
    assert ss.passive.StereoASW(exact=False)._exact() is False and ss.passive.StereoASW(alternate=True)._exact() is False
    with pytest.raises(ValueError):
        ss.passive.StereoASW(exact="always")._exact()
    with pytest.raises(ValueError):
        ss.passive.StereoASW(exact=True, alternate=True)._exact()
    with pytest.raises(ValueError):
        ss.passive._device_index(-2)
    assert ss.passive._device_index(None) == -1 and ss.passive._device_index(3) == 3
    m = ss.passive.StereoASW(winSize=7, maxDisparity=3)
    assert (m.winSize, m.maxDisparity, m.minDisparity, m.gammaC, m.gammaP, m.consistent) == (7, 3, 0, 5, 17.5, False)
    g = ss.passive.StereoGSW()
    assert (g.winSize, g.gamma, g.maxDisparity, g.minDisparity, g.fMax, g.iterations, g.bins) == (11, 10, 16, 0, 120, 3, 20)
    # compute(img1, img2) of the reference (passive.py:72, 147) + optional trailing keywords: devices=None,
    # rectify=None / interpolation=1 (rectification + matching in one call)
    for cls, extra in ((ss.passive.StereoASW, ["rectify", "interpolation"]), (ss.passive.StereoGSW, ["rectify", "interpolation"])):
        params = inspect.signature(cls.compute).parameters
        assert list(params) == ["self", "img1", "img2", "devices"] + extra and params["devices"].default is None
    assert inspect.signature(ss.passive.StereoASW.compute).parameters["rectify"].default is None
    with pytest.raises(ValueError):
        ss.passive._device_list([])
    with pytest.raises(ValueError):
        ss.passive._device_list([0, -1])
    with pytest.raises(ValueError):
        ss.passive._device_list(3)
    arr, n = ss.passive._device_list([2, 0, 1])
    assert n == 3 and list(arr) == [2, 0, 1]


@pytest.mark.parametrize("cls", ["StereoASW", "StereoGSW"])
def test_constructor_rejects_even_window(ss, cls):
    for bad in (4, 0, -3):
        with pytest.raises(ValueError, match="winSize must be a positive odd number!"):
            getattr(ss.passive, cls)(winSize=bad)


def test_compute_exceptions_match_reference_probes(ss, golden_errors, golden_inputs):
    """every failing probe recorded from the reference extension (tests/golden/errors.json)
    raises the same exception type and message here, before any device work"""
    a, b = golden_inputs("crop")
    E = {"ValueError": ValueError, "TypeError": TypeError}

    def asw(i1, i2, win=5, cons=False):
        m = ss.passive.StereoASW(winSize=5, maxDisparity=5, minDisparity=0, gammaC=5.0, gammaP=17.5, consistent=cons)
        m.winSize = win
        return m.compute(i1, i2)

    def gsw(i1, i2, win=5, gamma=10):
        m = ss.passive.StereoGSW(winSize=5, maxDisparity=5, minDisparity=0, gamma=gamma, fMax=120.0, iterations=3)
        m.winSize = win
        return m.compute(i1, i2)

    probes = {
        "asw_even_win": lambda: asw(a, b, win=4),
        "asw_zero_win": lambda: asw(a, b, win=0),
        "asw_float_img1": lambda: asw(a.astype(np.float32), b),
        "asw_gray": lambda: asw(a[:, :, 0].copy(), b[:, :, 0].copy()),
        "asw_shape_mismatch": lambda: asw(a, np.ascontiguousarray(b[:-1])),
        "asw_four_channels": lambda: asw(np.zeros((8, 8, 4), np.uint8), np.zeros((8, 8, 4), np.uint8)),
        "asw_float_win": lambda: asw(a, b, win=5.0),
        "asw_list_input": lambda: asw(a.tolist(), b),
        "gsw_float_gamma": lambda: gsw(a, b, gamma=10.5),
        "gsw_even_win": lambda: gsw(a, b, win=6),
        "gsw_float_img1": lambda: gsw(a.astype(np.float64), b),
        "gsw_shape_mismatch": lambda: gsw(a, np.ascontiguousarray(b[:, :-2])),
    }
    for name, call in probes.items():
        etype, msg = golden_errors[name]
        assert etype in E, name
        with pytest.raises(E[etype]) as ei:
            call()
        assert str(ei.value) == msg, name
    # documented deviations: the reference reads these as UB / silently; we refuse
    with pytest.raises(TypeError, match="Wrong type input!"):
        asw(a, b.astype(np.float32))                      # reference checks img1 twice (_passive.cpp:309)


@pytest.mark.parametrize("gc,gp", [(0, 17.5), (-5, 17.5), (5, 0), (5, -1.0), (float("nan"), 17.5)])
def test_asw_rejects_non_positive_gammas(gc, gp, ss, golden_inputs):
    """documented deviation: the reference computes with any gamma (_passive.cpp:47-50)"""
    a, b = golden_inputs("crop")
    with pytest.raises(ValueError, match="gammaC and gammaP must be positive"):
        ss.passive.StereoASW(winSize=5, maxDisparity=4, gammaC=gc, gammaP=gp).compute(a, b)


def test_c_abi_exports_every_declared_symbol(ss):
    from simplestereo_amd import _native
    lib = _native.lib()
    header = open(os.path.join(ROOT, "include", "ssamd.h")).read()
    declared = set(re.findall(r"\b(ssamd_[a-z0-9_]+)\s*\(", header))
    assert {"ssamd_asw", "ssamd_gsw", "ssamd_asw_device", "ssamd_gsw_device", "ssamd_asw_costs",
            "ssamd_bgr2lab", "ssamd_last_error", "ssamd_device_count"} <= declared
    for sym in declared:
        assert hasattr(lib, sym), sym
    m = re.search(r"#define SSAMD_ABI_VERSION (\d+)", header)
    assert lib.ssamd_abi_version() == int(m.group(1)) == _native.ABI_VERSION
    assert "asw_aggregate" in lib.ssamd_kernel_name(1).decode()


def test_experiment_library_override_needs_two_switches():
    """SSAMD_LIB (ablation builds: wrong maps by construction) is refused unless SSAMD_EXPERIMENT=1 is set as well"""
    import subprocess
    import sys
    env = dict(os.environ, SSAMD_LIB="/nonexistent/libssamd_ablation.so")
    env.pop("SSAMD_EXPERIMENT", None)
    r = subprocess.run([sys.executable, "-c", "import simplestereo_amd._native"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "SSAMD_EXPERIMENT=1" in r.stderr


def test_tuning_options_are_a_table_not_the_environment(ss):
    """the SSAMD_* hooks are read from the environment once, at load; afterwards only ssamd_set_option changes them
    (no getenv on the per-call host path) and unknown names are refused"""
    from simplestereo_amd import _native
    src = open(os.path.join(ROOT, "simplestereo_amd", "csrc", "ssamd_api.hip")).read()
    assert src.count("getenv(") == 1            # tuning_from_env, run once by the static initialiser
    f0 = _native.asw_kernel_form(1920, 10, 35, 16, 0)
    assert f0["wave_kernel"] in (4, 8)
    with _native.options(SSAMD_ASW_WAVE="0"):
        assert _native.asw_kernel_form(1920, 10, 35, 16, 0)["wave_kernel"] == 0
    assert _native.asw_kernel_form(1920, 10, 35, 16, 0) == f0
    with pytest.raises(_native.NativeError):
        _native.set_option("SSAMD_NO_SUCH_OPTION", "1")


def test_options_go_back_to_the_environment_baseline():
    """`with options(...)` / set_option(name, None) restore what the process was STARTED with, not "unset": a process
    launched with SSAMD_ASW_WAVE=0 keeps that after a block that overrides it; SSAMD_AUTOTUNE=NULL restores the mode too;
    a block whose second option is refused rolls the first one back"""
    code = r"""
import sys
sys.path.insert(0, %r)
from simplestereo_amd import _native
q = lambda: _native.asw_kernel_form(1920, 10, 35, 16, 0)["wave_kernel"]
assert q() == 0                                   # from the environment
with _native.options(SSAMD_ASW_WAVE="1"):
    assert q() in (4, 8)
assert q() == 0, "the load-time value must come back"
lib = _native.lib()
assert lib.ssamd_autotune(1) == 0                 # SSAMD_AUTOTUNE=0 from the environment; now forced on
_native.set_option("SSAMD_AUTOTUNE", "1")
_native.set_option("SSAMD_AUTOTUNE", None)
assert lib.ssamd_autotune(-1) == 0, "NULL restores the environment's autotune mode"
try:
    with _native.options(SSAMD_ASW_WAVE="1", SSAMD_NO_SUCH_OPTION="1"):
        raise SystemExit("the unknown option must be refused")
except _native.NativeError:
    pass
assert q() == 0, "the option set before the refused one must be rolled back"
print("ok")
""" % ROOT
    env = dict(os.environ, SSAMD_ASW_WAVE="0", SSAMD_AUTOTUNE="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def test_geometry_query_is_sane(ss):
    from simplestereo_amd import _native
    for (W, win, maxd, mind) in [(1920, 35, 192, 0), (640, 35, 64, 0), (384, 15, 16, 0), (4096, 35, 256, 0), (40, 7, 6, 1), (64, 255, 3, 0)]:
        g = _native.asw_geometry(W, 10, win, maxd, mind)
        nD = maxd - mind + 1
        assert g["tile_x"] % 4 == 0 and (g["chunk_d"] % 4 == 0 or g["chunk_d"] % 6 == 0)        # 8- or 4-column register tiles x 4 (or 6: wave6) disparities
        assert g["chunk_d"] * g["n_chunks"] >= nD and g["chunk_d"] * (g["n_chunks"] - 1) < nD
        assert g["threads"] % 64 == 0 and 64 <= g["threads"] <= 768
        assert g["lds_bytes"] <= 160 * 1024
        assert g["grid_x"] * g["tile_x"] >= W


def test_no_cpu_fallback_without_device(ss, golden_inputs):
    """without a GPU the operators fail loudly instead of computing on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from simplestereo_amd import _native
    a, b = golden_inputs("crop")
    with pytest.raises(_native.NativeError) as ei:
        ss.passive.StereoASW(winSize=5, maxDisparity=4).compute(a, b)
    assert ei.value.code == -2 and "no CPU fallback" in ei.value.message
    with pytest.raises(_native.NativeError):
        ss.passive.StereoGSW(winSize=5, maxDisparity=4).compute(a, b)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under simplestereo_amd/ may reference it"""
    pkg = os.path.join(ROOT, "simplestereo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "liboracle" not in src and "oracle_passive" not in src, f


def test_autotune_mode_round_trip(ss):
    """mode switch of the launch-geometry autotuner (pure host state: works without a GPU);
    default = None = 'small problems only'"""
    first = ss.passive.set_autotune(True)
    try:
        assert first in (None, True, False)
        assert ss.passive.set_autotune(False) is True
        assert ss.passive.set_autotune(None) is False
        assert ss.passive.set_autotune(True) is None
    finally:
        ss.passive.set_autotune(first)
