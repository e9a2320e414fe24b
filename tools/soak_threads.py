#!/usr/bin/env python3
"""GPU box: four Python threads on four streams hammer the library with mixed calls (ASW wave / phase-shifted / round-1
kernels, consistent, alternate, row ranges, GSW, host-array and device entry points) for N seconds; every result must
equal the one computed single-threaded beforehand.  usage: tools/soak_threads.py [seconds]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simplestereo_amd as ss
from simplestereo_amd.synth import make_pair

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cases = []
rng = np.random.default_rng(7)
for k in range(14):
    H, W = int(rng.integers(20, 120)), int(rng.integers(40, 400))
    maxd = int(rng.choice([3, 8, 16, 24, 40, 64, 100]))
    L, R, _ = make_pair(H, W, maxd, k)
    kind = ["asw", "asw_cons", "asw_alt", "gsw", "asw_rows", "asw_host", "asw_alt_rows"][k % 7]
    win = int(rng.choice([5, 9, 15, 21, 35]))
    cases.append((kind, L, R, win, maxd))


def run(case, stream=None):
    kind, L, R, win, maxd = case
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    H = L.shape[0]
    if kind == "asw":
        return ss.passive.StereoASW(winSize=win, maxDisparity=maxd).compute(tL, tR).cpu().numpy()
    if kind == "asw_cons":
        return ss.passive.StereoASW(winSize=win, maxDisparity=maxd, consistent=True).compute(tL, tR).cpu().numpy()
    if kind == "asw_alt":
        return ss.passive.StereoASW(winSize=win, maxDisparity=maxd, alternate=True).compute(tL, tR).cpu().numpy()
    if kind == "gsw":
        return ss.passive.StereoGSW(winSize=min(win, 11), maxDisparity=maxd).compute(tL, tR).cpu().numpy()
    if kind == "asw_rows":
        return ss.passive.StereoASW(winSize=win, maxDisparity=maxd)._compute_device(tL, tR, out_row0=H // 3, out_rows=H // 2).cpu().numpy()
    if kind == "asw_alt_rows":
        return ss.passive.StereoASW(winSize=win, maxDisparity=maxd, alternate=True)._compute_device(tL, tR, out_row0=H // 3, out_rows=H // 2).cpu().numpy()
    return ss.passive.StereoASW(winSize=win, maxDisparity=maxd, consistent=True).compute(L, R)


want = [run(c) for c in cases]
errors, counts = [], [0, 0, 0, 0]
t_end = time.time() + budget


def worker(tid):
    st = torch.cuda.Stream()
    r = np.random.default_rng(100 + tid)
    with torch.cuda.stream(st):
        while time.time() < t_end and not errors:
            k = int(r.integers(0, len(cases)))
            got = run(cases[k])
            if not np.array_equal(got, want[k]):
                errors.append((tid, k, cases[k][0], int((got != want[k]).sum())))
            counts[tid] += 1


th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
for t in th:
    t.start()
for t in th:
    t.join()
print("thread soak: %d calls from 4 threads in %.0f s, errors: %s" % (sum(counts), budget, errors or "none"))
sys.exit(1 if errors else 0)
