import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd.synth import make_pair
for name, (H, W, maxD, win) in {"default d16": (1080, 1920, 16, 35), "c3 d192": (1080, 1920, 192, 35), "tsukuba": (288, 384, 16, 15)}.items():
    L, R, _ = make_pair(H, W, maxD, 1)
    m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    for _ in range(3): m.compute(L, R); m.compute(tL, tR)
    torch.cuda.synchronize()
    n = 20 if maxD < 100 else 5
    t = time.perf_counter()
    for _ in range(n): d = m.compute(L, R)
    th = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n): d2 = m.compute(tL, tR)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t) / n * 1e3
    print("%-12s host arrays in/out %.3f ms   resident tensors %.3f ms   (copies + sync: %.3f ms)" % (name, th, td, th - td))
