"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of the plain-C restatement (``oracle_passive.c``) and loader of
the real reference extension built into ``oracle/_ref`` by ``oracle/Makefile``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; nothing in ``simplestereo_amd/`` does.

Parity pinning: the C restatement is checked bit-exactly against golden maps
produced by the unmodified reference (``tests/golden/make_golden.py``); see
``tests/test_oracle_golden.py``.
"""
import ctypes
import importlib.util
import os
import subprocess
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_passive.so")
_lib = None


def build(force=False):
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "oracle_passive.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_passive.so", "libm_check"])
    if os.path.exists("/root/reference/simplestereo/_passive.cpp") and (
            force or ref_module() is None or not os.path.exists(os.path.join(_HERE, "_ref", "libref_lab.so"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = ctypes.CDLL(_LIB)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.oracle_asw.restype = ctypes.c_int
        lib.oracle_asw.argtypes = [u8p, u8p] + [ctypes.c_int] * 5 + [ctypes.c_double] * 2 + \
            [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_double)]
        lib.oracle_gsw.restype = ctypes.c_int
        lib.oracle_gsw.argtypes = [u8p, u8p] + [ctypes.c_int] * 6 + [ctypes.c_float] + \
            [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int16)]
        lib.oracle_bgr2lab.restype = None
        lib.oracle_bgr2lab.argtypes = [u8p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int]
        _lib = lib
    return _lib


def _img(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("oracle expects uint8 [H,W,3]")
    return a


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def asw(img1, img2, winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5,
        consistent=False, hoist=True, nthreads=0, return_costs=False):
    """C restatement of ``_passive.computeASW`` (reference _passive.cpp:293-400)."""
    lib = _load()
    a, b = _img(img1), _img(img2)
    H, W = a.shape[:2]
    out = np.empty((H, W), np.int16)
    costs = None
    cptr = ctypes.POINTER(ctypes.c_double)()
    if return_costs:
        costs = np.empty((H, W, maxDisparity - minDisparity + 1), np.float64)
        cptr = costs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = lib.oracle_asw(_u8(a), _u8(b), H, W, winSize, maxDisparity, minDisparity,
                        float(gammaC), float(gammaP), int(bool(consistent)), int(bool(hoist)),
                        int(nthreads), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), cptr)
    if rc != 0:
        raise RuntimeError("oracle_asw failed rc=%d" % rc)
    return (out, costs) if return_costs else out


def asw_alternate(img1, img2, winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False,
                  exact_rows=None):
    """CPU restatement of the alternate-rows ASW mode (simplestereo_amd/csrc/asw_alt_kernels.hip.h).

    The reference only sketches the idea (docstring todo, reference passive.py:43-46) -- there is no
    reference output to pin this against ("parity unpinned" for this mode).  The restatement uses the
    reference-exact pieces: even rows = the exact map of ``asw`` (restated _passive.cpp:34-100), odd
    rows = argmin of the reference's fp64 costs over the disparities between the results above and
    below (first minimum wins, as in _passive.cpp:90-93).  ``exact_rows``: optional [H,W] map whose even
    rows replace the oracle's own (to test the fill stage in isolation from fp32-vs-fp64 argmin ties
    on the exact rows).  ``consistent``: the even rows come from the consistent mode (left-right check and
    occlusion filling of _passive.cpp:191-285); the odd rows are derived from them in the same way.
    Returns (map, number of evaluated pixels)."""
    exact, costs = asw(img1, img2, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent=False,
                       hoist=True, return_costs=True) if maxDisparity >= minDisparity else (asw(
                           img1, img2, winSize, maxDisparity, minDisparity, gammaC, gammaP), None)
    if consistent:
        exact = asw(img1, img2, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent=True)
    out = np.array(exact if exact_rows is None else exact_rows, dtype=np.int32)
    H, W = out.shape
    evaluated = 0
    for y in range(1, H, 2):
        up = out[y - 1]
        down = out[y + 1] if y + 1 < H else up
        for x in range(W):
            dmaxv = min(maxDisparity, x)
            if minDisparity > dmaxv:                 # empty candidate loop: the exact mode writes x
                out[y, x] = x
                continue
            lo = min(max(min(up[x], down[x]), minDisparity), dmaxv)
            hi = min(max(max(up[x], down[x]), minDisparity), dmaxv)
            if lo == hi:
                out[y, x] = lo
                continue
            c = costs[y, x, lo - minDisparity:hi - minDisparity + 1]
            out[y, x] = lo + int(np.argmin(c))       # np.argmin: first minimum = smallest disparity
            evaluated += 1
    return out.astype(np.int16), evaluated


def gsw(img1, img2, winSize=11, maxDisparity=16, minDisparity=0, gamma=10, fMax=120,
        iterations=3, bins=20, closed=False, nthreads=0):
    """C restatement of ``_passive.computeGSW`` (reference _passive.cpp:703-774)."""
    lib = _load()
    a, b = _img(img1), _img(img2)
    H, W = a.shape[:2]
    out = np.empty((H, W), np.int16)
    rc = lib.oracle_gsw(_u8(a), _u8(b), H, W, winSize, maxDisparity, minDisparity, int(gamma),
                        float(fMax), int(iterations), int(bins), int(bool(closed)), int(nthreads),
                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)))
    if rc != 0:
        raise RuntimeError("oracle_gsw failed rc=%d" % rc)
    return out


def bgr2lab(img):
    """C restatement of ColorConversion::ImageFromBGR2Lab (colorconversion.hpp:81-86)."""
    lib = _load()
    a = _img(img)
    H, W = a.shape[:2]
    lab = np.empty((H, W, 3), np.float64)
    lib.oracle_bgr2lab(_u8(a), lab.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), W, H)
    return lab


def ref_bgr2lab(img):
    """The reference's OWN ColorConversion::ImageFromBGR2Lab (headers/colorconversion.hpp:81-86), compiled where it
    lies behind oracle/ref_lab_harness.cpp into oracle/_ref/libref_lab.so; None when that file is absent."""
    path = os.path.join(_HERE, "_ref", "libref_lab.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.ref_bgr2lab.restype = None
    lib.ref_bgr2lab.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int]
    a = _img(img)
    H, W = a.shape[:2]
    lab = np.empty((H, W, 3), np.float64)
    lib.ref_bgr2lab(_u8(a), lab.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), W, H)
    return lab


_ref = False


def ref_module():
    """The real reference extension (``_passive``) from oracle/_ref, or None."""
    global _ref
    if _ref is False:
        path = os.path.join(_HERE, "_ref", "_passive" + sysconfig.get_config_var("EXT_SUFFIX"))
        _ref = None
        if os.path.exists(path):
            try:
                spec = importlib.util.spec_from_file_location("_passive", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _ref = mod
            except Exception:
                _ref = None
    return _ref
