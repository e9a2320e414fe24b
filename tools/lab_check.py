import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/oracle") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from simplestereo_amd import _native
from oracle import oracle
ramp = np.arange(256, dtype=np.uint8)
cube = np.stack(np.meshgrid(ramp, ramp, ramp[::3], indexing="ij"), -1).reshape(1, -1, 3).copy()
lab = np.empty(cube.shape, np.float32)
_native.check(_native.lib().ssamd_bgr2lab(cube.ctypes.data, 1, cube.shape[1], lab.ctypes.data, -1))
ref = oracle.bgr2lab(cube)
ref32 = ref.astype(np.float32)
d = np.abs(lab.astype(np.float64) - ref)
print(os.environ.get("SSAMD_LIB", "default"), "colours", cube.shape[1], "max|d| %.3e" % d.max(), "mean|d| %.3e" % d.mean(),
      "equal to float(ref): %.5f" % float(np.mean(lab == ref32)))
