"""ctypes binding of libssamd.so (C ABI in include/ssamd.h).

This is the only place Python touches the native library.  There is no Python or
CPU fallback for the operators: if the library is missing, or no HIP device is
visible, the calls fail loudly.
"""
import ctypes
import os

from .build import LIB_PATH as _DEFAULT_LIB_PATH

# experiments only (tools/build_variants.sh): another build of the same library is loaded only when the caller says
# twice that this is an experiment -- ablation builds compute wrong maps by construction
LIB_PATH = _DEFAULT_LIB_PATH
if os.environ.get("SSAMD_LIB"):
    if os.environ.get("SSAMD_EXPERIMENT") != "1":
        raise ImportError("SSAMD_LIB is an experiment hook: set SSAMD_EXPERIMENT=1 as well to load %s instead of the product "
                          "library" % os.environ["SSAMD_LIB"])
    LIB_PATH = os.environ["SSAMD_LIB"]
ABI_VERSION = 5

K_LAB, K_ASW_AGG, K_ASW_FIN, K_GSW_AGG, K_GSW_FIN, K_REMAP, K_REPROJECT, K_ASW_ALT, K_ASW_EXACT, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

_lib = None
_u8p = ctypes.c_void_p
_i16p = ctypes.c_void_p


class NativeError(RuntimeError):
    """A libssamd call returned an error code."""

    def __init__(self, code, msg):
        super().__init__("libssamd error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


def lib():
    """Load libssamd.so (once).  Raises ImportError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "simplestereo_amd: native library %s is missing. Build it with "
            "`python -m simplestereo_amd.build` (needs hipcc, targets gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  When torch is importable it goes
    # first, so libssamd's DT_NEEDED libamdhip64 resolves to the copy torch already loaded (same SONAME) and device
    # tensors, streams and this library share a runtime; loaded the other way round, whichever runtime initialises
    # second sees no device.
    try:
        import torch  # noqa: F401
    except Exception:      # noqa: BLE001 -- absent or broken torch: host-array calls do not need it
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.ssamd_abi_version.restype = ctypes.c_int
    if L.ssamd_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this package needs %d: rebuild it with `python -m simplestereo_amd.build`"
                          % (LIB_PATH, L.ssamd_abi_version(), ABI_VERSION))
    I, D, F, P = ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_void_p
    L.ssamd_abi_version.restype = I
    L.ssamd_last_error.restype = ctypes.c_char_p
    L.ssamd_device_count.restype = I
    L.ssamd_asw.restype = I
    L.ssamd_asw.argtypes = [P, P, I, I, I, I, I, D, D, I, P, I]
    L.ssamd_gsw.restype = I
    L.ssamd_gsw.argtypes = [P, P, I, I, I, I, I, I, F, I, I, P, I]
    L.ssamd_asw_multi.restype = I
    L.ssamd_asw_multi.argtypes = [P, P, I, I, I, I, I, D, D, I, P, ctypes.POINTER(I), I]
    L.ssamd_gsw_multi.restype = I
    L.ssamd_gsw_multi.argtypes = [P, P, I, I, I, I, I, I, F, I, I, P, ctypes.POINTER(I), I]
    L.ssamd_asw_device.restype = I
    L.ssamd_asw_device.argtypes = [P, P, I, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_gsw_device.restype = I
    L.ssamd_gsw_device.argtypes = [P, P, I, I, I, I, I, I, I, I, F, I, I, P, P]
    L.ssamd_asw_device_rows2.restype = I
    L.ssamd_asw_device_rows2.argtypes = [P, P, I, I, I, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_asw_exact.restype = I
    L.ssamd_asw_exact.argtypes = [P, P, I, I, I, I, I, D, D, I, P, I]
    L.ssamd_asw_exact_multi.restype = I
    L.ssamd_asw_exact_multi.argtypes = [P, P, I, I, I, I, I, D, D, I, P, ctypes.POINTER(I), I]
    L.ssamd_asw_exact_device.restype = I
    L.ssamd_asw_exact_device.argtypes = [P, P, I, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_gsw_device_rows2.restype = I
    L.ssamd_gsw_device_rows2.argtypes = [P, P, I, I, I, I, I, I, I, I, I, I, F, I, I, P, P]
    L.ssamd_asw_exact_device_rows2.restype = I
    L.ssamd_asw_exact_device_rows2.argtypes = [P, P, I, I, I, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_asw_exact_rectified_device.restype = I
    L.ssamd_asw_exact_rectified_device.argtypes = [P, P, I, I, P, P, P, P, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_asw_alternate.restype = I
    L.ssamd_asw_alternate.argtypes = [P, P, I, I, I, I, I, D, D, I, P, I]
    L.ssamd_asw_alternate_device.restype = I
    L.ssamd_asw_alternate_device.argtypes = [P, P, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_asw_alternate_rows_device.restype = I
    L.ssamd_asw_alternate_rows_device.argtypes = [P, P, I, I, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_asw_alternate_multi.restype = I
    L.ssamd_asw_alternate_multi.argtypes = [P, P, I, I, I, I, I, D, D, I, P, ctypes.POINTER(I), I]
    L.ssamd_asw_costs.restype = I
    L.ssamd_asw_costs.argtypes = [P, P, I, I, I, I, I, D, D, P, I]
    L.ssamd_asw_argmins.restype = I
    L.ssamd_asw_argmins.argtypes = [P, P, I, I, I, I, I, D, D, P, P, I]
    L.ssamd_bgr2lab.restype = I
    L.ssamd_bgr2lab.argtypes = [P, I, I, P, I]
    L.ssamd_remap_bgr_device.restype = I
    L.ssamd_remap_bgr_device.argtypes = [P, I, I, P, P, I, I, I, P, P]
    L.ssamd_reproject_device.restype = I
    L.ssamd_reproject_device.argtypes = [P, I, I, ctypes.POINTER(D), P, P]
    L.ssamd_debug_exact_queue.restype = I
    L.ssamd_debug_exact_queue.argtypes = [I, ctypes.c_longlong, P, P, P]
    L.ssamd_debug_libm.restype = I
    L.ssamd_debug_libm.argtypes = [I, I, P, P]
    L.ssamd_debug_exact_costs.restype = I
    L.ssamd_debug_exact_costs.argtypes = [P, P, I, I, I, D, D, I, P, P, P, P]
    L.ssamd_debug_gsw_sqrt.restype = I
    L.ssamd_debug_gsw_sqrt.argtypes = [I, P]
    L.ssamd_profile_enable.restype = I
    L.ssamd_profile_enable.argtypes = [I]
    L.ssamd_profile_reset.restype = I
    L.ssamd_profile_read.restype = I
    L.ssamd_profile_read.argtypes = [ctypes.POINTER(D), ctypes.POINTER(ctypes.c_longlong)]
    L.ssamd_kernel_name.restype = ctypes.c_char_p
    L.ssamd_kernel_name.argtypes = [I]
    L.ssamd_autotune.restype = I
    L.ssamd_autotune.argtypes = [I]
    L.ssamd_asw_geometry.restype = I
    L.ssamd_asw_geometry.argtypes = [I, I, I, I, I, ctypes.POINTER(I)]
    L.ssamd_asw_kernel_form.restype = I
    L.ssamd_asw_kernel_form.argtypes = [I, I, I, I, I, ctypes.POINTER(I)]
    L.ssamd_gsw_geometry.restype = I
    L.ssamd_gsw_geometry.argtypes = [I, I, I, I, I, ctypes.POINTER(I)]
    L.ssamd_set_option.restype = I
    L.ssamd_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.ssamd_asw_rectified_device.restype = I
    L.ssamd_asw_rectified_device.argtypes = [P, P, I, I, P, P, P, P, I, I, I, I, I, I, D, D, I, P, P]
    L.ssamd_gsw_rectified_device.restype = I
    L.ssamd_gsw_rectified_device.argtypes = [P, P, I, I, P, P, P, P, I, I, I, I, I, I, I, F, I, I, P, P]
    L.ssamd_counter.restype = I
    L.ssamd_counter.argtypes = [I, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]
    _lib = L
    return L


def set_option(name, value):
    """Experiment / test hook: set one of the SSAMD_* tuning options of the loaded library; ``None`` puts it back to what
    the environment had set when the library was loaded (unset if it had not).  The environment variables of the same
    names are only read at load time."""
    check(lib().ssamd_set_option(name.encode(), None if value is None else str(value).encode()))


def counter(name, device=-1):
    """Diagnostic counter of a device context (``evol_fallbacks``, ``evol_bytes``: include/ssamd.h)."""
    v = ctypes.c_longlong(0)
    check(lib().ssamd_counter(device, name.encode(), ctypes.byref(v)))
    return int(v.value)


class options:
    """``with _native.options(SSAMD_ASW_GEOM="3,5,8"): ...`` -- set tuning options for a block; afterwards every one of
    them is back at its load-time value.  If setting one fails, the ones already set are rolled back."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        done = []
        try:
            for k, v in self.kw.items():
                set_option(k, v)
                done.append(k)
        except Exception:
            for k in done:
                set_option(k, None)
            raise
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, None)
        return False


def check(rc):
    if rc != 0:
        raise NativeError(rc, lib().ssamd_last_error().decode("utf-8", "replace"))


def profile_read():
    ms = (ctypes.c_double * K_COUNT)()
    n = (ctypes.c_longlong * K_COUNT)()
    check(lib().ssamd_profile_read(ms, n))
    return list(ms), list(n)


def asw_geometry(width, rows, winSize, maxDisparity, minDisparity):
    out = (ctypes.c_int * 8)()
    check(lib().ssamd_asw_geometry(width, rows, winSize, maxDisparity, minDisparity, out))
    keys = ("tile_x", "chunk_d", "n_chunks", "threads", "lds_bytes", "grid_x", "grid_y", "grid_z")
    return dict(zip(keys, list(out)))


def asw_kernel_form(width, rows, winSize, maxDisparity, minDisparity):
    out = (ctypes.c_int * 5)()
    check(lib().ssamd_asw_kernel_form(width, rows, winSize, maxDisparity, minDisparity, out))
    return dict(zip(("phase_shifted", "tile_columns", "chunk_columns", "build_first_waves", "wave_kernel"), list(out)))


def gsw_geometry(width, rows, winSize, maxDisparity, minDisparity):
    out = (ctypes.c_int * 9)()
    check(lib().ssamd_gsw_geometry(width, rows, winSize, maxDisparity, minDisparity, out))
    keys = ("tile_x", "chunk_d", "n_chunks", "threads", "lds_bytes", "grid_x", "grid_y", "grid_z", "strip_rows")
    return dict(zip(keys, list(out)))
