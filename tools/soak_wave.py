#!/usr/bin/env python3
"""GPU box: randomised equality soak of asw_aggregate_wave_kernel against the workgroup kernels.
usage: tools/soak_wave.py [seconds]   -- random image sizes, windows, disparity ranges, modes; both register tiles, 1-4 waves
per workgroup, counted / straight-line build; every map (and, every 8th case, the raw cost volume) must equal the
workgroup kernels' bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
lib = _native.lib()
t_end = time.time() + budget
n = skipped = 0
HOOKS = ("SSAMD_ASW_WAVE", "SSAMD_ASW_WAVE_RX", "SSAMD_ASW_WAVE_WG", "SSAMD_ASW_WAVE_UNROLL", "SSAMD_ASW_WAVE_MERGE", "SSAMD_ASW_WAVE_RD", "SSAMD_ASW_WAVE_CREG")
while time.time() < t_end:
    H, W = int(rng.integers(1, 140)), int(rng.integers(1, 700))
    if rng.random() < 0.1:
        W = int(rng.integers(1800, 2100))
        H = int(rng.integers(1, 30))
    win = int(rng.choice([1, 3, 5, 7, 9, 11, 15, 21, 27, 35, 41, 63]))
    nD = int(rng.integers(1, 65)) if rng.random() < 0.8 else int(rng.choice([17, 18]))
    mind = int(rng.choice([0, 0, 0, 1, 3, 17]))
    maxd = mind + nD - 1
    cons = bool(rng.random() < 0.5)
    alt = (not cons) and rng.random() < 0.15
    L, R, _ = make_pair(H, W, max(1, maxd), int(rng.integers(0, 1 << 30)))
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=win, maxDisparity=maxd, minDisparity=mind, consistent=cons, alternate=alt,
                             gammaC=float(rng.choice([5.0, 7.0, 0.5, 50.0])), gammaP=float(rng.choice([17.5, 3.0, 100.0])))
    with_costs = n % 8 == 0 and not alt

    def costs():
        c = np.empty((H, W, nD), np.float32)
        _native.check(lib.ssamd_asw_costs(L.ctypes.data, R.ctypes.data, H, W, win, maxd, mind, m.gammaC, m.gammaP, c.ctypes.data, -1))
        return c
    try:
        _native.set_option("SSAMD_ASW_WAVE", "0")
        want = m.compute(tL, tR)
        want_c = costs() if with_costs else None
        _native.set_option("SSAMD_ASW_WAVE", "1")
        rx = str(rng.choice([8, 4]))
        _native.set_option("SSAMD_ASW_WAVE_RX", rx)
        wg_, unroll_ = str(rng.integers(1, 5)), str(rng.integers(0, 2))
        _native.set_option("SSAMD_ASW_WAVE_WG", wg_)
        _native.set_option("SSAMD_ASW_WAVE_UNROLL", unroll_)
        _native.set_option("SSAMD_ASW_WAVE_MERGE", None if rng.random() < 0.7 else "0")      # round 3: merged build rounds (default) / separate
        _native.set_option("SSAMD_ASW_WAVE_RD", None if rng.random() < 0.7 else "4")         # six disparities per lane where the host takes them / never
        _native.set_option("SSAMD_ASW_WAVE_CREG", None if rng.random() < 0.75 else "0")       # round 4: window centres in registers (default) / in LDS
        if _native.asw_kernel_form(W, H, win, maxd, mind)["wave_kernel"] != int(rx):
            skipped += 1
            continue
        got = m.compute(tL, tR)
        if not torch.equal(got, want):
            print("MISMATCH", dict(H=H, W=W, win=win, maxd=maxd, mind=mind, cons=cons, alt=alt, rx=rx,
                                   wg=wg_, unroll=unroll_,
                                   differing=int((got != want).sum())))
            sys.exit(1)
        if with_costs and not np.array_equal(costs(), want_c, equal_nan=True):
            print("COST MISMATCH", dict(H=H, W=W, win=win, maxd=maxd, mind=mind, rx=rx))
            sys.exit(1)
        n += 1
    finally:
        for k in HOOKS:
            _native.set_option(k, None)
print("soak ok: %d random cases equal (%d skipped: tile does not fit LDS) in %.0f s" % (n, skipped, budget))
