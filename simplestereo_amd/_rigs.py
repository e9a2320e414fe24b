"""
_rigs
=====
Stereo rig parameter containers with the API of ``simplestereo.StereoRig`` /
``simplestereo.RectifiedStereoRig`` (reference ``simplestereo/_rigs.py:22-338``,
``341-628``) -- the plumbing that feeds rectified pairs to ``passive.StereoASW``
(``examples/009 StereoMatchingASW.py:20-39``).

Own code, numpy only: OpenCV is not a dependency (it is absent on the MI355X
image).  Where the reference calls OpenCV, the published OpenCV camera model is
restated here:

====================================  ==========================================
reference call                         restated by
====================================  ==========================================
cv2.undistortPoints (4 corners)        :func:`_undistort_points`
cv2.initUndistortRectifyMap            :func:`_init_undistort_rectify_map`
cv2.remap (constant border)            :func:`_remap`
cv2.reprojectImageTo3D                 :meth:`RectifiedStereoRig.get3DPoints`
====================================  ==========================================

Parity status: JSON round trip and the 3x3 algebra are pinned by the reference's
own rig files (loaded as data in the tests); pixel values of ``rectifyImages`` can
not be pinned against cv2 in this environment ("parity unpinned"):
they are checked by self-consistency on synthetic rigs instead.
"""
import json

import numpy as np

__all__ = ["StereoRig", "RectifiedStereoRig"]

# OpenCV interpolation flag values, so callers can pass cv2.INTER_* or these
INTER_NEAREST = 0
INTER_LINEAR = 1


def _dist_vector(d):
    """OpenCV order (k1,k2,p1,p2[,k3[,k4,k5,k6[,s1,s2,s3,s4[,tx,ty]]]]) padded to 14."""
    v = np.zeros(14)
    d = np.asarray(d, dtype=np.float64).ravel()
    if d.size not in (0, 4, 5, 8, 12, 14):
        raise ValueError("distortion coefficients must have 4, 5, 8, 12 or 14 elements")
    v[:d.size] = d
    if v[12] != 0 or v[13] != 0:
        raise NotImplementedError("tilted sensor model (tauX, tauY) is not supported")
    return v


def _distort(x, y, dist):
    """Apply the OpenCV radial/tangential/thin-prism model to normalised coordinates."""
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = dist[:12]
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    kr = (1 + k1 * r2 + k2 * r4 + k3 * r6) / (1 + k4 * r2 + k5 * r4 + k6 * r6)
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r4
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r4
    return xd, yd


def _undistort_points(pts, K, distCoeffs, R=None, iterations=5):
    """cv2.undistortPoints semantics: pixel -> normalised, iterative undistortion, then R.
    ``iterations`` = 5 is OpenCV's default termination (TermCriteria(MAX_ITER, 5, 0.01) in cv::undistortPoints,
    modules/calib3d/src/undistort.dispatch.cpp), which the reference's ``_getCorners`` inherits
    (reference rectification.py:125-156)."""
    K = np.asarray(K, dtype=np.float64)
    dist = _dist_vector(distCoeffs)
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    x0 = (pts[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (pts[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = dist[:12]
    for _ in range(iterations):
        r2 = x * x + y * y
        icdist = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r2 * r2
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    if R is not None:
        R = np.asarray(R, dtype=np.float64)
        X = R[0, 0] * x + R[0, 1] * y + R[0, 2]
        Y = R[1, 0] * x + R[1, 1] * y + R[1, 2]
        Wc = R[2, 0] * x + R[2, 1] * y + R[2, 2]
        x, y = X / Wc, Y / Wc
    return np.stack([x, y], axis=1)


def _init_undistort_rectify_map(K, distCoeffs, R, newK, size):
    """cv2.initUndistortRectifyMap(..., CV_32FC1): float32 (mapx, mapy) of shape (h, w)."""
    K = np.asarray(K, dtype=np.float64)
    dist = _dist_vector(distCoeffs)
    w, h = int(size[0]), int(size[1])
    iR = np.linalg.inv(np.asarray(newK, dtype=np.float64)[:3, :3].dot(np.asarray(R, dtype=np.float64)))
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[0, 0] * u + iR[0, 1] * v + iR[0, 2]
    Y = iR[1, 0] * u + iR[1, 1] * v + iR[1, 2]
    Wc = iR[2, 0] * u + iR[2, 1] * v + iR[2, 2]
    x, y = X / Wc, Y / Wc
    xd, yd = _distort(x, y, dist)
    mapx = K[0, 0] * xd + K[0, 2]
    mapy = K[1, 1] * yd + K[1, 2]
    return mapx.astype(np.float32), mapy.astype(np.float32)


def _remap(img, mapx, mapy, interpolation=INTER_LINEAR):
    """cv2.remap with BORDER_CONSTANT(0), restating OpenCV's published arithmetic
    (modules/imgproc/src/imgwarp.cpp: ``remap`` converts float maps with ``cvRound(map * INTER_TAB_SIZE)``,
    INTER_BITS = 5; ``initInterTab2D`` builds 15-bit integer weights, INTER_REMAP_COEF_BITS = 15 -- with 5-bit
    fractions they are exactly a*b*32 and sum to 32768; ``remapBilinear`` for uint8 ends in
    ``FixedPtCast<int, uchar, 15>``: ``(sum + (1 << 14)) >> 15``, so ties round UP, not to even).
    uint8 images take that fixed-point path, with the common factor 32 divided out: ``(S + 512) >> 10``.  Every other
    dtype goes the way of OpenCV's non-uchar instantiations: float32 weights ``(1-fy)(1-fx)`` ... from the 1/32-pixel
    fractions, a float32 sum in tap order, and for uint16 / int16 ... ``saturate_cast`` = round half to EVEN
    (``np.rint``) + clip; float images keep the float sum.  cv2 is absent in this environment: parity with cv2.remap
    itself stays unpinned here (reference call: _rigs.py:564-565) -- tests/test_cv2_pins.py pins it wherever cv2 exists."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    src = img[:, :, None] if squeeze else img
    H, W = src.shape[:2]
    if interpolation == INTER_NEAREST:
        xi = np.rint(mapx).astype(np.int64)
        yi = np.rint(mapy).astype(np.int64)
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        out = np.zeros(mapx.shape + (src.shape[2],), src.dtype)
        out[ok] = src[yi[ok], xi[ok]]
    elif interpolation == INTER_LINEAR:
        q = np.rint(mapx.astype(np.float64) * 32).astype(np.int64)
        r = np.rint(mapy.astype(np.float64) * 32).astype(np.int64)
        x0, fx = q >> 5, q & 31
        y0, fy = r >> 5, r & 31
        fixed = src.dtype == np.uint8                       # OpenCV's 15-bit FixedPtCast path exists for uchar only
        acc = np.zeros(mapx.shape + (src.shape[2],), np.int64 if fixed else (np.float64 if src.dtype == np.float64 else np.float32))
        for dy, wy in ((0, 32 - fy), (1, fy)):
            for dx, wx in ((0, 32 - fx), (1, fx)):
                xx, yy = x0 + dx, y0 + dy
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                val = np.zeros_like(acc)
                val[ok] = src[yy[ok], xx[ok]]
                if fixed:
                    acc += val * (wy * wx)[..., None]      # a*b in 0..1024
                else:                                       # float32 weights a*b/1024 (exact), float32 products and sum
                    acc = acc + val * ((wy * wx).astype(np.float32) * np.float32(1.0 / 1024.0)).astype(acc.dtype)[..., None]
        if fixed:
            out = np.clip((acc + 512) >> 10, 0, 255).astype(np.uint8)
        elif np.issubdtype(src.dtype, np.integer):
            info = np.iinfo(src.dtype)
            out = np.clip(np.rint(acc), info.min, info.max).astype(src.dtype)      # saturate_cast: ties to even
        else:
            out = acc.astype(src.dtype)
    else:
        raise NotImplementedError("only INTER_NEAREST (0) and INTER_LINEAR (1) are available without OpenCV")
    return np.ascontiguousarray(out[:, :, 0] if squeeze else out)


def _cross_matrix(v):
    v = np.asarray(v, dtype=np.float64).ravel()
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _corners_after(H, K, dims, distCoeffs):
    """Image corners (clockwise from top-left) after undistortion and homography H
    (reference rectification.py:125-156)."""
    w, h = dims
    corners = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], dtype=np.float32)
    out = _undistort_points(corners, K, distCoeffs if distCoeffs is not None else np.zeros(5),
                            R=np.asarray(H).dot(K))
    return [(float(x), float(y)) for x, y in out.astype(np.float32)]


def _fitting_matrix(K1, K2, H1, H2, dims1, dims2, dist1=None, dist2=None, destDims=None, alpha=1):
    """Affine fit of both rectified images into destDims without losing rectification
    (reference rectification.py:17-122: common y scale/offset, common x scale, flips)."""
    if destDims is None:
        destDims = dims1
    c1 = _corners_after(H1, K1, dims1, dist1)
    c2 = _corners_after(H2, K2, dims2, dist2)
    xs1, xs2 = [p[0] for p in c1], [p[0] for p in c2]
    ys = [p[1] for p in c1 + c2]
    minX1, maxX1, minX2, maxX2 = min(xs1), max(xs1), min(xs2), max(xs2)
    minY, maxY = min(ys), max(ys)
    tL1, tR1, _, bL1 = c1
    flipX = -1 if tL1[0] > tR1[0] else 1
    flipY = -1 if tL1[1] > bL1[1] else 1
    spanX = max(maxX2 - minX2, maxX1 - minX1)
    scaleX = flipX * destDims[0] / spanX
    scaleY = flipY * destDims[1] / (maxY - minY)
    tX = -min(minX1, minX2) * scaleX if flipX == 1 else -min(maxX1, maxX2) * scaleX
    tY = -minY * scaleY if flipY == 1 else -maxY * scaleY
    Fit = np.array([[scaleX, 0, tX], [0, scaleY, tY], [0, 0, 1]], dtype=np.float64)
    if alpha >= 1:
        return Fit
    alpha = max(alpha, 0)
    tL1, tR1, bR1, bL1 = _corners_after(Fit.dot(H1), K1, destDims, dist1)
    tL2, tR2, bR2, bL2 = _corners_after(Fit.dot(H2), K2, destDims, dist2)
    left = max(tL1[0], bL1[0], tL2[0], bL2[0])
    right = min(tR1[0], bR1[0], tR2[0], bR2[0])
    top = max(tL1[1], tR1[1], tL2[1], tR2[1])
    bottom = min(bL1[1], bR1[1], bL2[1], bR2[1])
    s = max(destDims[0] / (right - left), destDims[1] / (bottom - top))
    s = (s - 1) * (1 - alpha) + 1
    Kz = np.array([[s, 0, -s * left], [0, s, -s * top], [0, 0, 1]], dtype=np.float64)
    return Kz.dot(Fit)


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and bool(x.is_cuda)


class StereoRig:
    """
    Keep together and manage all parameters of a calibrated stereo rig
    (same constructor, attributes and JSON keys as ``simplestereo.StereoRig``).

    Parameters
    ----------
    res1, res2 : sequence
        Resolution of camera as (width, height)
    intrinsic1, intrinsic2 : list or numpy.ndarray
        3x3 intrinsic camera matrix in the form [[fx, 0, tx], [0, fy, ty], [0, 0, 1]].
    distCoeffs1, distCoeffs2 : list or numpy.ndarray
        List of distortion coefficients of 4, 5, 8, 12 or 14 elements (OpenCV order).
    R : list or numpy.ndarray.
        3x3 rotation matrix between the 1st and the 2nd camera coordinate systems.
    T : list or numpy.ndarray
        Translation vector between the coordinate systems of the cameras.
    F, E : numpy.ndarray, optional
        Fundamental / essential matrix.
    reprojectionError : float, optional
        Total reprojection error resulting from calibration.
    """

    def __init__(self, res1, res2, intrinsic1, intrinsic2, distCoeffs1, distCoeffs2, R, T, F=None, E=None,
                 reprojectionError=None):
        self.res1 = res1
        self.res2 = res2
        self.intrinsic1 = intrinsic1
        self.intrinsic2 = intrinsic2
        self.distCoeffs1 = distCoeffs1
        self.distCoeffs2 = distCoeffs2
        self.R = R
        self.T = T
        self.F = F
        self.E = E
        self.reprojectionError = reprojectionError

    # ndarray-coercing properties (reference _rigs.py:68-130)
    intrinsic1 = property(lambda s: s._intrinsic1, lambda s, v: setattr(s, "_intrinsic1", np.asarray(v)))
    intrinsic2 = property(lambda s: s._intrinsic2, lambda s, v: setattr(s, "_intrinsic2", np.asarray(v)))
    distCoeffs1 = property(lambda s: s._distCoeffs1,
                           lambda s, d: setattr(s, "_distCoeffs1", np.asarray(d) if d is not None else np.zeros(5)))
    distCoeffs2 = property(lambda s: s._distCoeffs2,
                           lambda s, d: setattr(s, "_distCoeffs2", np.asarray(d) if d is not None else np.zeros(5)))
    R = property(lambda s: s._R, lambda s, v: setattr(s, "_R", np.asarray(v).reshape((3, 3))))
    T = property(lambda s: s._T, lambda s, v: setattr(s, "_T", np.asarray(v).reshape((-1, 1))))
    F = property(lambda s: s._F, lambda s, v: setattr(s, "_F", np.asarray(v).reshape((3, 3)) if v is not None else None))
    E = property(lambda s: s._E, lambda s, v: setattr(s, "_E", np.asarray(v).reshape((3, 3)) if v is not None else None))

    _KEYS = ("res1", "res2", "intrinsic1", "intrinsic2", "distCoeffs1", "distCoeffs2", "R", "T", "F", "E",
             "reprojectionError")

    @classmethod
    def fromFile(cls, filepath):
        """Alternative initialization of StereoRig object from JSON file."""
        with open(filepath, "r") as f:
            data = json.load(f)
        args = [data.get(k) for k in cls._KEYS]
        args[0], args[1] = tuple(args[0]), tuple(args[1])
        return cls(*args)

    def _as_dict(self):
        out = {}
        out["res1"] = self.res1
        out["res2"] = self.res2
        for k in ("intrinsic1", "intrinsic2", "R", "T", "distCoeffs1", "distCoeffs2"):
            out[k] = getattr(self, k).tolist()
        if self.F is not None:
            out["F"] = self.F.tolist()
        if self.E is not None:
            out["E"] = self.E.tolist()
        if self.reprojectionError:
            out["reprojectionError"] = self.reprojectionError
        return out

    def save(self, filepath):
        """Save configuration to JSON file (same keys as the reference, _rigs.py:164-191)."""
        with open(filepath, "w") as f:
            json.dump(self._as_dict(), f, indent=4)

    def getProjectionMatrices(self):
        """3x4 projection matrices of camera 1 and camera 2 (world origin in camera 1)."""
        Po1 = np.hstack((self.intrinsic1, np.zeros((3, 1))))
        Po2 = self.intrinsic2.dot(np.hstack((self.R, self.T)))
        return Po1, Po2

    def getCenters(self):
        """Camera centres in world coordinates (the first is always zero)."""
        _, Po2 = self.getProjectionMatrices()
        return np.zeros(3), -np.linalg.inv(Po2[:, :3]).dot(Po2[:, 3])

    def getBaseline(self):
        """Norm of the vector from camera 1 to camera 2."""
        return np.linalg.norm(self.getCenters()[1])

    def getFundamentalMatrix(self):
        """F (computed from the camera parameters when not set; Hartley & Zisserman form)."""
        if self.F is None:
            vv = _cross_matrix(self.intrinsic1.dot(self.R.T).dot(self.T))
            self.F = (np.linalg.inv(self.intrinsic2).T).dot(self.R).dot(self.intrinsic1.T).dot(vv)
        return self.F

    def getEssentialMatrix(self):
        """E = K2^T F K1 when not set."""
        if self.E is None:
            self.E = self.intrinsic2.T.dot(self.getFundamentalMatrix()).dot(self.intrinsic1)
        return self.E


class RectifiedStereoRig(StereoRig):
    """
    A calibrated and rectified stereo rig: all the parameters of :class:`StereoRig` plus the
    common orientation and the two rectifying *homographies* (pixel domain), as in
    ``simplestereo.RectifiedStereoRig``.

    Parameters
    ----------
    Rcommon : numpy.ndarray
        3x3 new common camera orientation after rectification.
    rectHomography1, rectHomography2 : numpy.ndarray
        3x3 rectification homographies.
    StereoRig : StereoRig or sequence
        A StereoRig object or, *alternatively*, all the parameters of :class:`StereoRig` (in order).
    """

    def __init__(self, Rcommon, rectHomography1, rectHomography2, *args):
        self.Rcommon = Rcommon
        self.rectHomography1 = rectHomography1
        self.rectHomography2 = rectHomography2
        self.K1 = None
        self.K2 = None
        if isinstance(args[0], StereoRig):
            r = args[0]
            super().__init__(r.res1, r.res2, r.intrinsic1, r.intrinsic2, r.distCoeffs1, r.distCoeffs2, r.R, r.T,
                             r.F, r.E, r.reprojectionError)
        else:
            super().__init__(*args)
        self.computeRectificationMaps()

    Rcommon = property(lambda s: s._Rcommon, lambda s, v: setattr(s, "_Rcommon", np.asarray(v).reshape((3, 3))))
    rectHomography1 = property(lambda s: s._rectHomography1,
                               lambda s, v: setattr(s, "_rectHomography1", np.asarray(v).reshape((3, 3))))
    rectHomography2 = property(lambda s: s._rectHomography2,
                               lambda s, v: setattr(s, "_rectHomography2", np.asarray(v).reshape((3, 3))))

    @classmethod
    def fromFile(cls, filepath):
        """Alternative initialization from a JSON file written by :meth:`save` (or by the reference)."""
        with open(filepath, "r") as f:
            data = json.load(f)
        head = [data.get(k) for k in ("Rcommon", "rectHomography1", "rectHomography2")]
        return cls(*(head + [data.get(k) for k in StereoRig._KEYS]))

    def save(self, filepath):
        """Save configuration to JSON file (keys as reference _rigs.py:439-469)."""
        out = {"Rcommon": self.Rcommon.tolist(), "rectHomography1": self.rectHomography1.tolist(),
               "rectHomography2": self.rectHomography2.tolist()}
        out.update(self._as_dict())
        with open(filepath, "w") as f:
            json.dump(out, f, indent=4)

    def getRectifiedProjectionMatrices(self):
        """Projection matrices after rectification: common orientation, horizontal displacement only."""
        C1, C2 = self.getCenters()
        P1 = self.K1.dot(self.Rcommon).dot(np.hstack((np.eye(3), -C1[:, None])))
        P2 = self.K2.dot(self.Rcommon).dot(np.hstack((np.eye(3), -C2[:, None])))
        return P1, P2

    def computeRectificationMaps(self, destDims=None, alpha=1):
        """
        Compute the two maps to undistort and rectify the stereo pair (reference
        ``_rigs.py:491-541``): fitting affinity, the new camera matrices ``K1``/``K2`` that
        track every transformation applied after rectification, then the float32 maps
        ``mapx1, mapy1, mapx2, mapy2`` of shape (height, width).
        """
        if destDims is None:
            destDims = self.res1
        Fit = _fitting_matrix(self.intrinsic1, self.intrinsic2, self.rectHomography1, self.rectHomography2,
                              self.res1, self.res2, self.distCoeffs1, self.distCoeffs2, destDims, alpha)
        self.K1 = Fit.dot(self.rectHomography1).dot(self.intrinsic1).dot(self.Rcommon.T)
        self.K2 = Fit.dot(self.rectHomography2).dot(self.intrinsic2.dot(self.R)).dot(self.Rcommon.T)
        R1 = self.Rcommon
        R2 = self.Rcommon.dot(self.R.T)
        self.mapx1, self.mapy1 = _init_undistort_rectify_map(self.intrinsic1, self.distCoeffs1, R1, self.K1, destDims)
        self.mapx2, self.mapy2 = _init_undistort_rectify_map(self.intrinsic2, self.distCoeffs2, R2, self.K2, destDims)
        # device copies of the previous maps are stale (a new array may reuse a freed address) and would stay resident in HBM
        self.__dict__["_maps_gen"] = self.__dict__.get("_maps_gen", 0) + 1
        self.__dict__.pop("_dev_maps", None)

    def rectifyImages(self, img1, img2, interpolation=INTER_LINEAR):
        """
        Undistort, rectify and fit a couple of images coming from the stereo rig.  Returns two
        C-contiguous arrays of the destination resolution, ready for ``StereoASW.compute``.

        Extension: two CUDA/HIP ``torch.uint8 [H,W,3]`` tensors are remapped on the GPU
        (``remap_bgr_kernel``; the maps are uploaded once per rig and device) and returned as
        device tensors, so the rectified pair can go straight into ``compute`` without a host
        round trip.
        """
        if _is_device_tensor(img1) and _is_device_tensor(img2):
            return (self._remap_device(img1, 1, interpolation), self._remap_device(img2, 2, interpolation))
        return (_remap(img1, self.mapx1, self.mapy1, interpolation),
                _remap(img2, self.mapx2, self.mapy2, interpolation))

    def _device_maps(self, which, device):
        """float32 maps of camera `which` (1 / 2) as tensors on `device`, uploaded once per rig, device and map array"""
        import torch
        mx, my = (self.mapx1, self.mapy1) if which == 1 else (self.mapx2, self.mapy2)
        key = (which, str(device), self.__dict__.get("_maps_gen", 0), mx.ctypes.data, mx.shape)
        cache = self.__dict__.setdefault("_dev_maps", {})
        if key not in cache:                      # maps change only through computeRectificationMaps
            cache[key] = (torch.from_numpy(np.ascontiguousarray(mx)).to(device),
                          torch.from_numpy(np.ascontiguousarray(my)).to(device))
        return cache[key]

    def _remap_device(self, img, which, interpolation):
        import ctypes
        import torch
        from . import _native
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("device rectification expects uint8 [H,W,3] tensors")
        if interpolation not in (INTER_NEAREST, INTER_LINEAR):
            raise NotImplementedError("only INTER_NEAREST (0) and INTER_LINEAR (1) are available")
        dmx, dmy = self._device_maps(which, img.device)
        src = img.contiguous()
        h, w = (self.mapx1 if which == 1 else self.mapx2).shape
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=img.device)
        with torch.cuda.device(img.device):
            stream = torch.cuda.current_stream(img.device).cuda_stream
            _native.check(_native.lib().ssamd_remap_bgr_device(src.data_ptr(), int(src.shape[0]), int(src.shape[1]),
                                                               dmx.data_ptr(), dmy.data_ptr(), h, w, int(interpolation),
                                                               out.data_ptr(), ctypes.c_void_p(stream)))
        return out

    def getQ(self):
        """The 4x4 disparity-to-depth matrix built by the reference in get3DPoints (_rigs.py:604-625)."""
        b = self.getBaseline()
        fx, fy = self.K1[0, 0], self.K2[1, 1]
        cx1, cx2 = self.K1[0, 2], self.K2[0, 2]
        a1, a2 = self.K1[0, 1], self.K2[0, 1]
        cy = self.K1[1, 2]
        Q = np.eye(4, dtype="float64")
        Q[0, 1] = -a1 / fy
        Q[0, 3] = a1 * cy / fy - cx1
        Q[1, 1] = fx / fy
        Q[1, 3] = -cy * fx / fy
        Q[2, 2] = 0
        Q[2, 3] = -fx
        Q[3, 1] = (a2 - a1) / (fy * b)
        Q[3, 2] = 1 / b
        Q[3, 3] = ((a1 - a2) * cy + (cx2 - cx1) * fy) / (fy * b)
        return Q

    def get3DPoints(self, disparityMap):
        """
        3D points (height, width, 3) float32 from a disparity map, world origin in the left camera:
        [X Y Z W]^T = Q [x y d 1]^T, point = (X/W, Y/W, Z/W)  (cv2.reprojectImageTo3D semantics).
        """
        if _is_device_tensor(disparityMap):
            return self._get3DPoints_device(disparityMap)
        d = np.asarray(disparityMap)
        h, w = d.shape[:2]
        Q = self.getQ()
        x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        v = np.stack([x, y, d.astype(np.float64), np.ones_like(x)], axis=-1)
        p = v.dot(Q.T)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = p[..., :3] / p[..., 3:4]
        return out.astype(np.float32)

    def _get3DPoints_device(self, disp):
        """int16 disparity tensor on the GPU -> float32 [H,W,3] point tensor on the GPU (reproject_kernel)."""
        import ctypes
        import torch
        from . import _native
        if disp.dtype != torch.int16 or disp.dim() != 2:
            raise ValueError("device reprojection expects an int16 [H,W] tensor")
        d = disp.contiguous()
        h, w = int(d.shape[0]), int(d.shape[1])
        Q = np.ascontiguousarray(self.getQ(), dtype=np.float64)
        out = torch.empty((h, w, 3), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            stream = torch.cuda.current_stream(d.device).cuda_stream
            _native.check(_native.lib().ssamd_reproject_device(d.data_ptr(), h, w,
                                                               Q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                               out.data_ptr(), ctypes.c_void_p(stream)))
        return out
