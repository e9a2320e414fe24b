"""GPU box (one GPU): time the kernels of ONE rank's strip (strip + halo rows resident) for world sizes 1, 2, 4, 8 --
the compute part of a strong-scaling step; the halo exchange and the gather are not included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import strips
from simplestereo_amd.synth import make_pair

for (H, W, maxd) in [(1080, 1920, 192), (2160, 4096, 256)]:
    L, R, _ = make_pair(H, W, maxd, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=35, maxDisparity=maxd)
    base = None
    for world in (1, 2, 4, 8):
        worst = 0.0
        for rank in sorted({0, world // 2, world - 1}):
            r0, r1 = strips.strip_bounds(H, world, rank)
            h0, h1 = strips.halo_bounds(H, r0, r1, 17)
            a, b = tL[h0:h1].contiguous(), tR[h0:h1].contiguous()
            for _ in range(2):
                m._compute_device(a, b, out_row0=r0 - h0, out_rows=r1 - r0)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5):
                m._compute_device(a, b, out_row0=r0 - h0, out_rows=r1 - r0)
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t) / 5 * 1e3)
        base = base or worst
        print("%dx%d D0..%d  world %d: slowest strip %.3f ms  -> compute-only speed-up %.2fx" % (W, H, maxd, world, worst, base / worst))
