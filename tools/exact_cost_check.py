"""GPU box: the tie-break pass's fp64 costs and Lab values against the oracle's, bit for bit, on the soak's failing shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
from oracle import oracle
rng = np.random.default_rng(11)
for (H, W, win, maxd, gc, gp, flat) in [(10, 221, 35, 56, 5.0, 17.5, True), (14, 253, 35, 19, 5.0, 3.0, True), (53, 66, 21, 39, 0.7, 17.5, True),
                                         (30, 120, 9, 30, 5.0, 17.5, False), (5, 199, 21, 50, 0.7, 3.0, False)]:
    L, R, _ = make_pair(H, W, max(1, maxd), int(rng.integers(0, 1 << 30)))
    if flat:
        L = (L // 64 * 64).astype(np.uint8); R = np.ascontiguousarray(R[:, ::-1])
    L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
    ref, cref = oracle.asw(L, R, winSize=win, maxDisparity=maxd, gammaC=gc, gammaP=gp, return_costs=True)
    ys, xs = rng.integers(0, H, 3000), rng.integers(0, W, 3000)
    ds = np.minimum(rng.integers(0, maxd + 1, 3000), xs)
    yxd = np.ascontiguousarray(np.stack([ys, xs, ds], 1).astype(np.int32))
    costs = np.empty(3000, np.float64); l1 = np.empty((H, W, 3)); l2 = np.empty((H, W, 3))
    _native.check(_native.lib().ssamd_debug_exact_costs(L.ctypes.data, R.ctypes.data, H, W, win, gc, gp, 3000, yxd.ctypes.data, costs.ctypes.data,
                                                        l1.ctypes.data, l2.ctypes.data))
    want = cref[ys, xs, ds]
    nb = int(np.count_nonzero(costs.view(np.uint64) != want.view(np.uint64)))
    ol1, ol2 = oracle.bgr2lab(L), oracle.bgr2lab(R)
    nl = int(np.count_nonzero(l1.view(np.uint64) != ol1.view(np.uint64))) + int(np.count_nonzero(l2.view(np.uint64) != ol2.view(np.uint64)))
    print("%dx%d win %d D 0..%d gammaC %g gammaP %g flat=%s: %d of 3000 fp64 costs differ from the oracle's (max rel %.3g); %d of %d fp64 Lab values differ" %
          (W, H, win, maxd, gc, gp, flat, nb, float(np.max(np.abs(costs - want) / np.maximum(want, 1e-300))), nl, 2 * l1.size))
    if nb:
        k = int(np.flatnonzero(costs.view(np.uint64) != want.view(np.uint64))[0])
        print("   first: (y, x, d) =", yxd[k], "gpu %.17g oracle %.17g" % (costs[k], want[k]))
