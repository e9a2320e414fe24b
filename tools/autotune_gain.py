"""GPU box: model-chosen vs autotuned launch geometry across disparity ranges (1080p, win 35)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd.synth import make_pair

def run(m, tL, tR, n=4):
    m.compute(tL, tR); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t = time.perf_counter(); m.compute(tL, tR); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best

for (H, W, win, ds) in [(1080, 1920, 35, (7, 16, 24, 32, 40, 47, 55, 64, 80, 96, 128, 160, 192)), (1080, 1920, 21, (16, 64, 128, 192)), (1080, 1920, 11, (16, 64, 192)),
                        (480, 640, 35, (16, 32, 64)), (288, 384, 15, (16,)), (720, 1280, 21, (32, 64, 128)), (2160, 4096, 35, (256,))]:
    for maxd in ds:
        L, R, _ = make_pair(H, W, maxd, 1)
        tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        m = ss.passive.StereoASW(winSize=win, maxDisparity=maxd)
        ss.passive.set_autotune(False)
        t_model = run(m, tL, tR)
        ss.passive.set_autotune(True)
        t0 = time.perf_counter(); m.compute(tL, tR); torch.cuda.synchronize(); t_tune = (time.perf_counter() - t0) * 1e3
        t_tuned = run(m, tL, tR)
        ss.passive.set_autotune(False)
        print("%dx%d win %d D0..%d: model %.3f ms, autotuned %.3f ms (%.1f %%), tuning call %.0f ms" % (W, H, win, maxd, t_model, t_tuned, 100 * (t_model - t_tuned) / t_model, t_tune))
