#!/usr/bin/env python3
"""GPU box: StereoGSW (win 11, class defaults otherwise) at 1080p over disparity ranges: kernel ms, VALU fraction at 2 lane-ops per tap"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
import bench
H, W = 1080, 1920
L, R, _ = make_pair(H, W, 192, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
lib = _native.lib()
for maxD, win in ((16, 11), (7, 11), (32, 11), (64, 11), (192, 11), (16, 21), (16, 5)):
    m = ss.passive.StereoGSW(winSize=win, maxDisparity=maxD)
    for _ in range(2): d = m.compute(tL, tR)
    torch.cuda.synchronize()
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    n = 8
    for _ in range(n): d = m.compute(tL, tR)
    torch.cuda.synchronize()
    ms, cnt = _native.profile_read(); lib.ssamd_profile_enable(0)
    k = ms[_native.K_GSW_AGG] / n
    taps = 2 * bench.count_taps(H, W, win, maxD, 0)
    print("GSW 1080p win %d D 0..%d: %.3f ms (all kernels %.3f)  valu_frac %.3f  geometry %s" %
          (win, maxD, k, sum(ms) / n, 2 * taps / (k * 1e-3) / bench.VALU_PEAK_LANEOPS, _native.gsw_geometry(W, H, win, maxD, 0)), flush=True)
