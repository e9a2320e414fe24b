#!/usr/bin/env python3
"""GPU box: StereoGSW kernel ms over forced launch geometries (SSAMD_GSW_GEOM="XG,DG,Ty[,Hy]") for one problem; maps must not change.
usage: tools/gsw_geom_sweep.py maxD win "XG,DG,Ty,Hy" ...   (1080p frame)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
maxD, win = int(sys.argv[1]), int(sys.argv[2])
H, W = (int(os.environ.get("GSW_H", 1080)), int(os.environ.get("GSW_W", 1920)))
L, R, _ = make_pair(H, W, 192, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
lib = _native.lib()
m = ss.passive.StereoGSW(winSize=win, maxDisparity=maxD)
base = None
for geom in [None] + sys.argv[3:]:
    try:
        _native.set_option("SSAMD_GSW_GEOM", geom)
        for _ in range(2): d = m.compute(tL, tR)
        torch.cuda.synchronize()
        lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
        for _ in range(8): d = m.compute(tL, tR)
        torch.cuda.synchronize()
        ms, n = _native.profile_read(); lib.ssamd_profile_enable(0)
        if base is None: base = d
        g = _native.gsw_geometry(W, H, win, maxD, 0)
        print("D 0..%d win %d  %-12s %.3f ms  same=%s  threads %d lds %d tile %d x %d" % (maxD, win, geom, ms[_native.K_GSW_AGG] / 8, bool(torch.equal(d, base)), g["threads"], g["lds_bytes"], g["tile_x"], g["chunk_d"]), flush=True)
    except Exception as e:
        print(geom, "ERR", repr(e)[:80])
    finally:
        _native.set_option("SSAMD_GSW_GEOM", None)
