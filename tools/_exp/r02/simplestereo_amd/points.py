"""
points
======
Point-cloud helpers at the end of the matching pipeline, with the call signatures of
``simplestereo.points`` (reference ``simplestereo/points.py:10-176``): ASCII PLY export / import
and the rig-less reprojection of a disparity map.  Own, vectorised code (the reference formats one
point per Python loop iteration); files written here use the same header lines, so they can be read
back by the reference's ``importPLY`` and vice versa.
"""
import numpy as np

__all__ = ["exportPLY", "importPLY", "getAdimensional3DPoints"]


def exportPLY(points3D, filepath, referenceImage=None, precision=6):
    """
    Export raw point cloud to PLY file (ASCII).

    Parameters
    ----------
    points3D : numpy.ndarray
        Array of 3D points. The last dimension must contain ordered x,y,z coordinates.
    filepath : str
        File path for the PLY file (absolute or relative).
    referenceImage : numpy.ndarray, optional
        Image to take colours from: same number of points as `points3D`; last dimension 3 (BGR) or
        1 / absent (intensity; like the reference, only int64 images are written as ``int`` -- every
        other dtype, uint8 included, as ``float`` in a field of width `precision`, reference points.py:62-80).
    precision : int
        Decimal places of the coordinates. Default 6.
    """
    points3D = np.asarray(points3D)
    shape = points3D.shape
    pts = points3D.reshape(-1, 3).astype(np.float64)
    n = pts.shape[0]
    head = ["ply", "format ascii 1.0", "comment SimpleStereo point cloud export",
            "comment Original array shape " + "x".join(str(d) for d in shape),
            "element vertex %d" % n, "property double x", "property double y", "property double z"]
    fmt = ["%.{p}f".format(p=precision)] * 3
    cols = [pts]
    if referenceImage is not None:
        img = np.asarray(referenceImage)
        if img.size == pts.size:                               # BGR -> RGB like the reference
            head += ["property uchar red", "property uchar green", "property uchar blue"]
            cols.append(img.reshape(-1, 3)[:, ::-1].astype(np.float64))
            fmt += ["%d"] * 3
        elif np.issubdtype(img.dtype, np.int64):               # the reference's test (points.py:62)
            head.append("property int intensity")
            cols.append(img.reshape(-1, 1).astype(np.float64))
            fmt.append("%d")
        else:
            head.append("property float intensity")
            cols.append(img.reshape(-1, 1).astype(np.float64))
            fmt.append("%{p}f".format(p=precision))            # "{:{p}f}": width p, six decimals (points.py:78)
    head.append("end_header")
    with open(filepath, "w") as f:
        f.write("\n".join(head) + "\n")
        np.savetxt(f, np.hstack(cols), fmt=" ".join(fmt))


def importPLY(filename, *properties):
    """
    Import values from an ASCII PLY file: the property columns `properties` (default 0, 1, 2 =
    x, y, z) of every vertex line, as a float array of shape (number of vertices, len(properties));
    an empty cloud gives shape (0,) like the reference's ``np.asarray([], dtype=float)`` (points.py:121).
    """
    if not properties:
        properties = (0, 1, 2)
    with open(filename, "r") as f:
        for line in f:
            if line.rstrip().lower() == "end_header":
                break
        body = f.read()
    if not body.strip():
        return np.zeros((0,), dtype=float)
    import io
    data = np.loadtxt(io.StringIO(body), dtype=float, ndmin=2)
    return np.ascontiguousarray(data[:, list(properties)]) if data.size else np.zeros((0,), dtype=float)


def getAdimensional3DPoints(disparityMap):
    """
    Adimensional 3D points from a disparity map when the rig is not known (reference
    ``points.py:124-176``): the Q matrix of a unit-baseline rig with fx = fy = width and the principal
    point in the image centre, then [X Y Z W] = Q [x y d 1], point = (X/W, Y/W, Z/W).
    """
    d = np.asarray(disparityMap)
    height, width = d.shape[:2]
    fx = fy = float(width)
    cx, cy = width / 2, height / 2
    Q = np.eye(4)
    Q[0, 3] = -cx
    Q[1, 1] = fx / fy
    Q[1, 3] = -cy * fx / fy
    Q[2, 2] = 0
    Q[2, 3] = -fx
    Q[3, 2] = 1.0
    Q[3, 3] = 0.0
    x, y = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    v = np.stack([x, y, d.astype(np.float64), np.ones_like(x)], -1).dot(Q.T)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (v[..., :3] / v[..., 3:4]).astype(np.float32)
