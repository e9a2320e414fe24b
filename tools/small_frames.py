"""GPU box: wall time per compute() vs kernel time (library HIP events) for small frames -- how much of a
small-frame call is launch / host overhead"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair

for (H, W, maxd, win, cons, gsw) in [(288, 384, 16, 15, False, False), (288, 384, 16, 35, True, False), (480, 640, 64, 35, False, False),
                                     (288, 384, 16, 11, True, True), (480, 640, 64, 11, True, True)]:
    L, R, _ = make_pair(H, W, maxd, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoGSW(winSize=win, maxDisparity=maxd) if gsw else ss.passive.StereoASW(winSize=win, maxDisparity=maxd, consistent=cons)
    for _ in range(3):
        m.compute(tL, tR)
    torch.cuda.synchronize()
    n = 50
    t = time.perf_counter()
    for _ in range(n):
        d = m.compute(tL, tR)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n):
        d = m.compute(tL, tR)
        torch.cuda.synchronize()
    wall_sync = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n):
        dn = m.compute(L, R)
    wall_host = (time.perf_counter() - t) / n * 1e3
    _native.lib().ssamd_profile_enable(1); _native.lib().ssamd_profile_reset()
    for _ in range(10):
        m.compute(tL, tR)
    torch.cuda.synchronize()
    ms, cnt = _native.profile_read()
    _native.lib().ssamd_profile_enable(0)
    ker = sum(ms) / 10
    print("%dx%d D%d win%d %s%s: back-to-back %.3f ms/call, synced %.3f, host numpy %.3f, kernels %.3f ms  %s" % (
        W, H, maxd, win, "GSW" if gsw else "ASW", " consistent" if cons else "", wall, wall_sync, wall_host, ker,
        " ".join("%s=%.3f" % (k, v / 10) for k, v in zip(("lab", "agg", "fin", "gagg", "gfin", "remap", "reproj"), ms) if v)))
