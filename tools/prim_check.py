"""GPU box: are the device's fp64 sqrt and division the host's (IEEE, correctly rounded)?  ssamd_debug_libm which = 2 / 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplestereo_amd import _native
rng = np.random.default_rng(3)
x = np.concatenate([rng.random(2000000) * 3.0e4, rng.random(1000000) * 50.0, rng.random(500000) * 1e-3,
                    (rng.integers(0, 400, 500000) ** 2).astype(np.float64), np.array([0.0, 1.0, 4.0, 2.0, 1e-300, 1e300])])
out = np.empty_like(x)
_native.check(_native.lib().ssamd_debug_libm(2, x.size, x.ctypes.data, out.ctypes.data))
want = np.sqrt(x)
bad = np.count_nonzero(out.view(np.uint64) != want.view(np.uint64))
print("sqrt (with the residual correction): %d of %d differ from the host's" % (bad, x.size))
_native.check(_native.lib().ssamd_debug_libm(3, x.size, x.ctypes.data, out.ctypes.data))
want = x / 0.7 + x / 5.0
print("x / 0.7 + x / 5.0: %d of %d differ from the host's" % (np.count_nonzero(out.view(np.uint64) != want.view(np.uint64)), x.size))
