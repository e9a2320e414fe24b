"""profiling helper (GPU box): run the bench workload's kernels a few times without the CPU baseline"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd.synth import make_pair
ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=1080); ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--maxd", type=int, default=192); ap.add_argument("--win", type=int, default=35)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--consistent", action="store_true")
ap.add_argument("--gsw", action="store_true"); ap.add_argument("--alternate", action="store_true")
a = ap.parse_args()
L, R, _ = make_pair(a.h, a.w, a.maxd, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
m = ss.passive.StereoGSW(winSize=a.win, maxDisparity=a.maxd) if a.gsw else ss.passive.StereoASW(winSize=a.win, maxDisparity=a.maxd, consistent=a.consistent, alternate=a.alternate)
for i in range(a.steps):
    torch.cuda.synchronize(); t = time.perf_counter()
    d = m.compute(tL, tR); torch.cuda.synchronize()
    print("step %d: %.3f ms  checksum %d" % (i, (time.perf_counter() - t) * 1e3, int(d.long().sum())))
