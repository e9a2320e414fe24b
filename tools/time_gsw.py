#!/usr/bin/env python3
"""GPU box: kernel-slot timing of StereoGSW config 4 (1080p, D 0..192, win 11); env selects variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
H, W, maxD, win = 1080, 1920, 192, 11
if len(sys.argv) > 1: maxD = int(sys.argv[1])
L, R, _ = make_pair(H, W, 192, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
m = ss.passive.StereoGSW(winSize=win, maxDisparity=maxD)
lib = _native.lib()
d = m.compute(tL, tR); torch.cuda.synchronize()
lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
for _ in range(5): d = m.compute(tL, tR)
torch.cuda.synchronize()
ms, n = _native.profile_read()
print({lib.ssamd_kernel_name(i).decode(): round(ms[i] / 5, 3) for i in range(_native.K_COUNT) if n[i]}, "checksum", int(d.long().sum()))
