"""
passive
=======
Passive stereo matchers with the API of ``simplestereo.passive`` (reference
``simplestereo/passive.py``), executed by hand-written HIP kernels on an AMD
MI355X through the C ABI of ``libssamd.so`` (``include/ssamd.h``).

    import simplestereo_amd as ss
    disparity = ss.passive.StereoASW(winSize=35, maxDisparity=64).compute(imgL, imgR)

Same class names, keyword names, defaults, public attributes, return type
(a fresh ``numpy.int16 [H, W]`` array) and exceptions as the reference:

=====================================  ===========================================
condition                              exception (reference ``_passive.cpp`` line)
=====================================  ===========================================
img not an ndarray / int arg a float   ``ValueError("Invalid input format!")`` (304, 712)
image dtype is not uint8               ``TypeError("Wrong type input!")`` (312, 720)
not [H,W,3] or shapes differ           ``ValueError("Wrong image dimensions!")`` (319, 727)
winSize even or <= 0                   ``ValueError("winSize must be a positive odd number!")`` (323, 731)
=====================================  ===========================================

Documented deviations that turn undefined behaviour of the reference into
defined behaviour: ``img2``'s dtype is checked too (the reference tests ``img1``
twice, ``_passive.cpp:309``), non-contiguous inputs are made contiguous (the
reference reads them as if contiguous), negative ``minDisparity`` is rejected
(out-of-bounds reads in the reference), and ``StereoASW`` refuses ``gammaC <= 0`` / ``gammaP <= 0`` with
``ValueError`` (the reference computes with any value, ``_passive.cpp:47-50``: a negative gamma turns the support
weights into GROWING exponentials whose products overflow fp32 where the reference's fp64 does not, a zero gamma
makes every cost NaN -- neither is a matcher anyone runs; ``StereoGSW`` accepts negative ``gamma`` like the
reference, golden G9f).

Extensions (not in the reference): ``compute`` also accepts two CUDA/HIP
``torch.uint8`` tensors ``[H,W,3]`` already resident in HBM and then returns a
``torch.int16`` tensor on the same device without any host round trip; and both
classes take one extra trailing keyword, ``device`` (default ``None`` = the
process's current HIP device), the GPU index that host-array calls run on.
``compute`` of both classes also takes an optional keyword ``devices=[i, j, ...]`` (host arrays only): the frame
is cut into one row strip per listed GPU, each strip is matched on its own device concurrently and the map is
reassembled on the host -- bit-identical to the one-GPU result, because rows are independent jobs in the
reference as well (``_passive.cpp:372-374``).
``StereoASW`` also takes ``alternate`` (default ``False``): the faster
"every other pixel" variant that the reference's docstring sketches as a todo
(reference ``passive.py:43-46``) and never implemented.
"""
import ctypes
import operator

import numpy as np

from . import _native

__all__ = ["StereoASW", "StereoGSW", "set_autotune"]

_INT_MIN, _INT_MAX = -2 ** 31, 2 ** 31 - 1


def _c_int(v):
    """PyArg_ParseTuple 'i': anything with __index__ that fits a C int; floats are refused.
    Every parse failure surfaces as ValueError("Invalid input format!") (_passive.cpp:304)."""
    try:
        i = operator.index(v)
    except TypeError:
        raise ValueError("Invalid input format!") from None
    if not _INT_MIN <= i <= _INT_MAX:
        raise ValueError("Invalid input format!")
    return i


def _c_double(v):
    """PyArg_ParseTuple 'd' / 'f': anything float() accepts."""
    try:
        return float(v)
    except (TypeError, ValueError):
        raise ValueError("Invalid input format!") from None


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and bool(x.is_cuda)


def _check_pair(img1, img2):
    """The checks of _passive.cpp:301-321 in the reference's order; returns contiguous arrays."""
    if not isinstance(img1, np.ndarray) or not isinstance(img2, np.ndarray):
        raise ValueError("Invalid input format!")                       # "O!" with PyArray_Type
    if img1.dtype != np.uint8 or img2.dtype != np.uint8:
        raise TypeError("Wrong type input!")
    if (img1.ndim != 3 or img2.ndim != 3 or img1.shape[2] != 3 or img2.shape[2] != 3
            or img1.shape[0] != img2.shape[0] or img1.shape[1] != img2.shape[1]):
        raise ValueError("Wrong image dimensions!")
    return np.ascontiguousarray(img1), np.ascontiguousarray(img2)


def _check_pair_tensors(t1, t2):
    import torch
    if t1.dtype != torch.uint8 or t2.dtype != torch.uint8:
        raise TypeError("Wrong type input!")
    if (t1.dim() != 3 or t2.dim() != 3 or t1.shape[2] != 3 or t2.shape[2] != 3
            or t1.shape[0] != t2.shape[0] or t1.shape[1] != t2.shape[1] or t1.device != t2.device):
        raise ValueError("Wrong image dimensions!")
    return t1.contiguous(), t2.contiguous()


def _device_index(device):
    """None -> -1 (current device of the calling thread); otherwise a non-negative GPU index."""
    if device is None:
        return -1
    i = _c_int(device)
    if i < 0:
        raise ValueError("device must be None or a non-negative GPU index")
    return i


def _device_list(devices):
    """devices=[...] of compute(): distinct non-negative GPU indices as a C int array"""
    try:
        idx = [_c_int(d) for d in devices]
    except TypeError:
        raise ValueError("devices must be a sequence of GPU indices") from None
    if not idx or any(i < 0 for i in idx):
        raise ValueError("devices must be a non-empty sequence of distinct non-negative GPU indices")
    return (ctypes.c_int * len(idx))(*idx), len(idx)


def set_autotune(on=True):
    """Extension: launch-geometry autotuning of ``StereoASW`` and ``StereoGSW``.  ``True``: the first ``compute`` call of
    every problem shape times its candidate geometries on the GPU (ASW: up to ~50 extra kernel launches, once; GSW: up to
    36 candidate strips, one warm-up launch each -- those twice as slow as the best are dropped there -- and two timed
    rounds of the rest) instead of trusting the cost model; ``False``: never; ``None``: the default -- only for small calls
    (at most 6e10 window taps, i.e. a few milliseconds of kernel time), where that costs at most ~0.4 s once.  Maps are
    unaffected: every geometry accumulates the same taps in the same order.  Returns the previous mode (True / False /
    None).  The environment variable ``SSAMD_AUTOTUNE=1 / 0 / -1`` sets the initial mode."""
    before = _native.lib().ssamd_autotune(-1 if on is None else (1 if on else 0))
    return None if before < 0 else bool(before)


def _raise_native(e):
    if e.code == -1:
        raise ValueError(e.message) from None
    raise e


class StereoASW():
    """
    Adaptive Support-Weight stereo matching (K. Yoon, I. Kweon, 2006) -- drop-in for
    ``simplestereo.passive.StereoASW`` (reference ``passive.py:16-92``).

    Parameters (same names, order and defaults as the reference constructor)
    ----------
    winSize : int
        Edge length of the square support window in pixels; odd and positive (default 35).
    maxDisparity : int
        Largest disparity that is tried, inclusive (default 16).
    minDisparity : int
        Smallest disparity that is tried, inclusive (default 0; negative values are refused).
    gammaC : float
        Scale of the colour term exp(-dLab / gammaC) of the support weights (default 5).
    gammaP : float
        Scale of the spatial term exp(-dist / gammaP) of the support weights (default 17.5).
    device : int or None
        Extension: GPU index for host-array calls (default None: the current HIP device).
    alternate : bool
        Extension, the todo of the reference's docstring (``passive.py:43-46``): match every other image
        row exactly and let each pixel of the rows in between search only the disparities between the
        results above and below it (copied when they agree).  About twice as fast; not the reference's
        output (about 2-3 % of the pixels differ on Tsukuba, bad-1.0 against the ground truth does not
        get worse).  With ``consistent`` the exact rows go through the left-right check and the occlusion
        filling first.  Whole images only (default False).
    consistent : bool
        Also match with the right image as reference, invalidate left pixels whose match does
        not agree, and fill each invalid run with the smaller of its two valid neighbours
        (default False).  On the GPU this costs one extra reduction, not a second aggregation,
        because the aggregated cost is symmetric in the (left pixel, right pixel) pair.
    exact : bool or "auto"
        Extension (default ``"auto"`` = on, except with ``alternate``): the aggregation kernels queue every candidate whose fp32
        cost is a near-tie of its pixel's winner (1.5e-5 relative at the defaults), the queue is re-evaluated in fp64 with the
        reference's own expression and summation order (reference ``_passive.cpp:37-50, 57-88``) and those argmins are redone
        -- the map is then the reference's wherever double precision can tell the candidates apart: bit for bit on every golden
        map, the whole bench frames and 38 642 random frames (``ssamd_asw_exact*``, include/ssamd.h).  Costs about
        1 % at 1080p / 193 disparities and O(H * W) scratch.  ``False``: the fp32 argmin only (>= 99.99 % of the pixels identical on
        full frames, ~99.9 % on a class-default photograph).  ``True`` with ``alternate=True`` raises.  If the candidate queue
        overflows (a frame of saturated noise), the fp32 map is returned and a ``RuntimeWarning`` is issued on host-array calls
        (``_native.counter("exact_overflow")`` for device tensors).
    """

    def __init__(self, winSize=35, maxDisparity=16, minDisparity=0, gammaC=5, gammaP=17.5, consistent=False,
                 device=None, alternate=False, exact="auto"):
        if not (winSize > 0 and winSize % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        self.device = device
        self.alternate = alternate
        self.exact = exact
        self.winSize = winSize
        self.maxDisparity = maxDisparity
        self.minDisparity = minDisparity
        self.gammaC = gammaC
        self.gammaP = gammaP
        self.consistent = consistent

    def _params(self):
        win, maxd, mind = _c_int(self.winSize), _c_int(self.maxDisparity), _c_int(self.minDisparity)
        gc, gp = _c_double(self.gammaC), _c_double(self.gammaP)
        if not (gc > 0 and gp > 0):          # documented deviation (module docstring)
            raise ValueError("gammaC and gammaP must be positive")
        return win, maxd, mind, gc, gp, 1 if self.consistent else 0

    def _alternate(self, cons):
        return bool(getattr(self, "alternate", False))

    def _exact(self):
        ex = getattr(self, "exact", "auto")
        alt = bool(getattr(self, "alternate", False))
        if isinstance(ex, str):
            if ex != "auto":
                raise ValueError('exact must be True, False or "auto"')
            return not alt
        if ex and alt:
            raise ValueError("exact=True is not available with alternate=True")
        return bool(ex)

    @staticmethod
    def _warn_on_overflow(dev):
        """host-array calls are synchronous: report a queue overflow of the tie-break pass (the fp32 map was kept)"""
        try:
            if _native.counter("exact_overflow", dev if dev is not None and dev >= 0 else -1):
                import warnings
                warnings.warn("StereoASW(exact): the near-tie queue overflowed (%d candidates); the fp32 map was returned"
                              % _native.counter("exact_entries", dev if dev is not None and dev >= 0 else -1), RuntimeWarning, stacklevel=3)
        except Exception:      # noqa: BLE001  -- a diagnostic must not fail the call
            pass

    def compute(self, img1, img2, devices=None, rectify=None, interpolation=1):
        """
        Disparity map of a rectified BGR pair.

        img1, img2: left and right image, ``numpy.uint8`` arrays ``[H, W, 3]`` in OpenCV channel
        order (or two device tensors, see the module docstring).  Returns a new ``numpy.int16``
        array ``[H, W]`` of left-referenced disparities.  ``devices`` (extension, host arrays only):
        list of GPU indices that share the frame as row strips.

        ``rectify`` (extension, device tensors only): a ``RectifiedStereoRig`` whose maps are computed -- ``img1`` / ``img2``
        are then the RAW frames, and ``rig.rectifyImages(img1, img2, interpolation)`` followed by ``compute`` (the reference's
        pipeline, examples/009) runs as one call in which rectification and Lab conversion are a single launch and the
        rectified frames never exist in HBM.  Same map, bit for bit, as the two calls.
        """
        if rectify is not None:
            if not (_is_device_tensor(img1) and _is_device_tensor(img2)) or devices is not None:
                raise ValueError("rectify=rig applies to two device tensors (raw frames resident in HBM)")
            if tuple(img1.shape) != tuple(img2.shape):
                # cameras of different resolutions (res1 != res2, which rectifyImages supports): the fused launch takes one
                # source size for both frames, so such rigs go through the two calls it otherwise replaces -- same map
                return self.compute(*rectify.rectifyImages(img1, img2, interpolation))
            return self._compute_rectified_device(rectify, img1, img2, interpolation)
        if _is_device_tensor(img1) and _is_device_tensor(img2):
            if devices is not None:
                raise ValueError("devices=[...] applies to host arrays; device tensors are matched where they live")
            return self._compute_device(img1, img2)
        lib = _native.lib()
        if not isinstance(img1, np.ndarray) or not isinstance(img2, np.ndarray):
            raise ValueError("Invalid input format!")
        win, maxd, mind, gc, gp, cons = self._params()
        exact = self._exact()                            # (raises for exact + alternate)
        dev = _device_index(getattr(self, "device", None))
        a, b = _check_pair(img1, img2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        H, W = a.shape[:2]
        out = np.empty((H, W), np.int16)
        try:
            if devices is not None:
                arr, n = _device_list(devices)
                multi = lib.ssamd_asw_alternate_multi if self._alternate(cons) else (lib.ssamd_asw_exact_multi if exact else lib.ssamd_asw_multi)
                _native.check(multi(a.ctypes.data, b.ctypes.data, H, W, win, maxd, mind, gc, gp, cons, out.ctypes.data, arr, n))
                return out
            if self._alternate(cons):
                _native.check(lib.ssamd_asw_alternate(a.ctypes.data, b.ctypes.data, H, W, win, maxd, mind, gc, gp, cons,
                                                      out.ctypes.data, dev))
                return out
            op = lib.ssamd_asw_exact if exact else lib.ssamd_asw
            _native.check(op(a.ctypes.data, b.ctypes.data, H, W, win, maxd, mind, gc, gp, cons, out.ctypes.data, dev))
        except _native.NativeError as e:
            _raise_native(e)
        if exact:
            self._warn_on_overflow(dev)
        return out

    def _compute_rectified_device(self, rig, raw1, raw2, interpolation=1):
        """raw frames -> (rectification + Lab records in one launch) -> matcher: ssamd_asw_rectified_device"""
        import torch
        lib = _native.lib()
        win, maxd, mind, gc, gp, cons = self._params()
        if self._alternate(cons):
            raise ValueError("rectify=rig is not available with alternate=True")
        exact = self._exact()
        a, b = _check_pair_tensors(raw1, raw2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        if interpolation not in (0, 1):
            raise NotImplementedError("only INTER_NEAREST (0) and INTER_LINEAR (1) are available")
        if getattr(rig, "mapx1", None) is None:
            raise ValueError("the rig has no rectification maps: call computeRectificationMaps() first")
        mx1, my1 = rig._device_maps(1, a.device)
        mx2, my2 = rig._device_maps(2, a.device)
        if mx1.shape != mx2.shape:
            raise ValueError("Wrong image dimensions!")
        H, W = int(mx1.shape[0]), int(mx1.shape[1])
        out = torch.empty((H, W), dtype=torch.int16, device=a.device)
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            try:
                op = lib.ssamd_asw_exact_rectified_device if exact else lib.ssamd_asw_rectified_device
                _native.check(op(a.data_ptr(), b.data_ptr(), int(a.shape[0]), int(a.shape[1]),
                                 mx1.data_ptr(), my1.data_ptr(), mx2.data_ptr(), my2.data_ptr(), H, W,
                                 int(interpolation), win, maxd, mind, gc, gp, cons, out.data_ptr(), ctypes.c_void_p(stream)))
            except _native.NativeError as e:
                _raise_native(e)
        return out

    def _compute_device(self, t1, t2, out_row0=0, out_rows=None, row_parity=0, out=None, skip=None):
        """Operands already in HBM (torch tensors): returns a torch.int16 tensor on the device.
        Rows [out_row0, out_row0+out_rows) of the given (sub-)image are matched.  ``row_parity`` (alternate=True only):
        parity of the sub-image's first row in the whole image, whose even rows are the exactly matched ones; such a
        sub-image carries ``winSize // 2 + 1`` halo rows.  ``out``: a contiguous int16 [out_rows, W] tensor to write into
        instead of a new one.  ``skip = (row, n)``: rows [row, row + n) of the range are left untouched in ``out`` and the two
        bands either side of them run as ONE launch (``ssamd_asw_device_rows2``; row strips, strips.py)."""
        import torch
        lib = _native.lib()
        win, maxd, mind, gc, gp, cons = self._params()
        a, b = _check_pair_tensors(t1, t2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        H, W = int(a.shape[0]), int(a.shape[1])
        rows = H - out_row0 if out_rows is None else int(out_rows)
        alt = self._alternate(cons)
        exact = self._exact()                            # (raises for exact + alternate)
        if out is None:
            out = torch.empty((rows, W), dtype=torch.int16, device=a.device)
        elif out.dtype != torch.int16 or tuple(out.shape) != (rows, W) or not out.is_contiguous() or out.device != a.device:
            raise ValueError("out must be a contiguous int16 [out_rows, W] tensor on the images' device")
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            try:
                if skip is not None and skip[1] > 0:
                    if alt:
                        raise ValueError("two row ranges: not with alternate=True")
                    op = lib.ssamd_asw_exact_device_rows2 if exact else lib.ssamd_asw_device_rows2
                    _native.check(op(a.data_ptr(), b.data_ptr(), H, W, int(out_row0), rows, int(skip[0]), int(skip[1]),
                                     win, maxd, mind, gc, gp, cons, out.data_ptr(), ctypes.c_void_p(stream)))
                    return out
                if alt:
                    _native.check(lib.ssamd_asw_alternate_rows_device(a.data_ptr(), b.data_ptr(), H, W, int(out_row0), rows,
                                                                      int(row_parity) & 1, win, maxd, mind, gc, gp, cons,
                                                                      out.data_ptr(), ctypes.c_void_p(stream)))
                    return out
                op = lib.ssamd_asw_exact_device if exact else lib.ssamd_asw_device
                _native.check(op(a.data_ptr(), b.data_ptr(), H, W, int(out_row0), rows, win, maxd, mind, gc, gp, cons, out.data_ptr(),
                                 ctypes.c_void_p(stream)))
            except _native.NativeError as e:
                _raise_native(e)
        return out


class StereoGSW():
    """
    Geodesic Support-Weight matching as implemented by the reference
    (``simplestereo.passive.StereoGSW``, reference ``passive.py:99-158``): left- and
    right-referenced winner-take-all on geodesically weighted, truncated colour
    distances, left-right check and occlusion filling (always on).

    Parameters (same names, order and defaults as the reference constructor)
    ----------
    winSize : int
        Edge length of the square support window; odd and positive (default 11).
    maxDisparity, minDisparity : int
        Inclusive disparity search range (defaults 16 and 0).
    gamma : int
        Scale of exp(-geodesic distance / gamma); must be an ``int`` because the reference parses
        it with the "i" format (default 10).
    fMax : int or float
        Cap of the per-pixel colour distance (default 120).
    iterations : int
        Relaxation sweeps of the reference's distance transform; any value >= 1 gives the same
        weights, 0 keeps only the window centre (default 3).
    bins : int
        Accepted for signature compatibility; the reference never reads it (default 20).
    device : int or None
        Extension: GPU index for host-array calls (default None: the current HIP device).
    """

    def __init__(self, winSize=11, maxDisparity=16, minDisparity=0, gamma=10, fMax=120, iterations=3, bins=20,
                 device=None):
        if not (winSize > 0 and winSize % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        self.device = device
        self.winSize = winSize
        self.gamma = gamma
        self.maxDisparity = maxDisparity
        self.minDisparity = minDisparity
        self.fMax = fMax
        self.iterations = iterations
        self.bins = bins

    def _params(self):
        return (_c_int(self.winSize), _c_int(self.maxDisparity), _c_int(self.minDisparity), _c_int(self.gamma),
                _c_double(self.fMax), _c_int(self.iterations), _c_int(self.bins))

    def compute(self, img1, img2, devices=None, rectify=None, interpolation=1):
        """Disparity map of a rectified 3-channel pair (uint8 [H,W,3]); returns int16 [H,W].
        ``devices`` (extension, host arrays only): list of GPU indices that share the frame as row strips.
        ``rectify`` (extension, device tensors only): a ``RectifiedStereoRig`` with computed maps -- ``img1`` / ``img2`` are then
        the RAW frames, rectified and matched in one call (see ``StereoASW.compute``); same map as the two calls."""
        if rectify is not None:
            if not (_is_device_tensor(img1) and _is_device_tensor(img2)) or devices is not None:
                raise ValueError("rectify=rig applies to two device tensors (raw frames resident in HBM)")
            if tuple(img1.shape) != tuple(img2.shape):
                # cameras of different resolutions (res1 != res2, which rectifyImages supports): the fused launch takes one
                # source size for both frames, so such rigs go through the two calls it otherwise replaces -- same map
                return self.compute(*rectify.rectifyImages(img1, img2, interpolation))
            return self._compute_rectified_device(rectify, img1, img2, interpolation)
        if _is_device_tensor(img1) and _is_device_tensor(img2):
            if devices is not None:
                raise ValueError("devices=[...] applies to host arrays; device tensors are matched where they live")
            return self._compute_device(img1, img2)
        lib = _native.lib()
        if not isinstance(img1, np.ndarray) or not isinstance(img2, np.ndarray):
            raise ValueError("Invalid input format!")
        win, maxd, mind, gamma, fmax, it, bins = self._params()
        dev = _device_index(getattr(self, "device", None))
        a, b = _check_pair(img1, img2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        H, W = a.shape[:2]
        out = np.empty((H, W), np.int16)
        try:
            if devices is not None:
                arr, n = _device_list(devices)
                _native.check(lib.ssamd_gsw_multi(a.ctypes.data, b.ctypes.data, H, W, win, maxd, mind, gamma, fmax, it,
                                                  bins, out.ctypes.data, arr, n))
                return out
            _native.check(lib.ssamd_gsw(a.ctypes.data, b.ctypes.data, H, W, win, maxd, mind, gamma, fmax, it, bins,
                                        out.ctypes.data, dev))
        except _native.NativeError as e:
            _raise_native(e)
        return out

    def _compute_rectified_device(self, rig, raw1, raw2, interpolation=1):
        """raw frames -> (rectification + pixel packing in one launch) -> matcher: ssamd_gsw_rectified_device"""
        import torch
        lib = _native.lib()
        win, maxd, mind, gamma, fmax, it, bins = self._params()
        a, b = _check_pair_tensors(raw1, raw2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        if interpolation not in (0, 1):
            raise NotImplementedError("only INTER_NEAREST (0) and INTER_LINEAR (1) are available")
        if getattr(rig, "mapx1", None) is None:
            raise ValueError("the rig has no rectification maps: call computeRectificationMaps() first")
        mx1, my1 = rig._device_maps(1, a.device)
        mx2, my2 = rig._device_maps(2, a.device)
        if mx1.shape != mx2.shape:
            raise ValueError("Wrong image dimensions!")
        H, W = int(mx1.shape[0]), int(mx1.shape[1])
        out = torch.empty((H, W), dtype=torch.int16, device=a.device)
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            try:
                _native.check(lib.ssamd_gsw_rectified_device(a.data_ptr(), b.data_ptr(), int(a.shape[0]), int(a.shape[1]),
                                                             mx1.data_ptr(), my1.data_ptr(), mx2.data_ptr(), my2.data_ptr(), H, W,
                                                             int(interpolation), win, maxd, mind, gamma, fmax, it, bins, out.data_ptr(),
                                                             ctypes.c_void_p(stream)))
            except _native.NativeError as e:
                _raise_native(e)
        return out

    def _compute_device(self, t1, t2, out_row0=0, out_rows=None, out=None, skip=None):
        """see StereoASW._compute_device: ``out`` = a contiguous int16 [out_rows, W] tensor to write into, ``skip = (row, n)`` =
        rows left untouched between two bands (``ssamd_gsw_device_rows2``; the overlapped strip step of strips.py)"""
        import torch
        lib = _native.lib()
        win, maxd, mind, gamma, fmax, it, bins = self._params()
        a, b = _check_pair_tensors(t1, t2)
        if not (win > 0 and win % 2 == 1):
            raise ValueError("winSize must be a positive odd number!")
        H, W = int(a.shape[0]), int(a.shape[1])
        rows = H - out_row0 if out_rows is None else int(out_rows)
        if out is None:
            out = torch.empty((rows, W), dtype=torch.int16, device=a.device)
        elif out.dtype != torch.int16 or tuple(out.shape) != (rows, W) or not out.is_contiguous() or out.device != a.device:
            raise ValueError("out must be a contiguous int16 [out_rows, W] tensor on the images' device")
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            try:
                if skip is not None and skip[1] > 0:
                    _native.check(lib.ssamd_gsw_device_rows2(a.data_ptr(), b.data_ptr(), H, W, int(out_row0), rows, int(skip[0]), int(skip[1]),
                                                             win, maxd, mind, gamma, fmax, it, bins, out.data_ptr(), ctypes.c_void_p(stream)))
                    return out
                _native.check(lib.ssamd_gsw_device(a.data_ptr(), b.data_ptr(), H, W, int(out_row0), rows, win,
                                                   maxd, mind, gamma, fmax, it, bins, out.data_ptr(),
                                                   ctypes.c_void_p(stream)))
            except _native.NativeError as e:
                _raise_native(e)
        return out
