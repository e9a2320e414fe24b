import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
L, R, _ = make_pair(1080, 1920, 192, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
m = ss.passive.StereoGSW(winSize=11, maxDisparity=192)
lib = _native.lib()
def run(geom, n=12):
    _native.set_option("SSAMD_GSW_GEOM", geom)
    try:
        for _ in range(3): out = m.compute(tL, tR)
        torch.cuda.synchronize()
        lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
        t0 = time.perf_counter()
        for _ in range(n): out = m.compute(tL, tR)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        ms, launches = _native.profile_read(); lib.ssamd_profile_enable(0)
        return wall, ms[_native.K_GSW_AGG] / n, out
    finally:
        _native.set_option("SSAMD_GSW_GEOM", None)
base = None
geoms = [None, "20,25,2,1", "20,25,2,2,1", "10,49,4,1", "8,49,4,1", "10,33,4,1", "15,33,4,1", "12,41,4,1", "20,25,4,1", "10,25,4,1", "6,49,4,1"]
for rep in range(2):
    for gm in geoms:
        try:
            w, k, out = run(gm)
        except Exception as e:
            print(gm, "ERR", repr(e)[:100]); continue
        if base is None: base = out
        print("%-16s wall %.3f ms  kernels %.3f ms  same=%s" % (gm, w, k, bool(torch.equal(out, base))), flush=True)
