#!/usr/bin/env python3
"""GPU box: randomised equality soak of asw_aggregate_pipe_kernel (lane order, border-tile thread deal, compile-time strides,
TAD volume / in-kernel e tiles, both chunk lengths and wave orders) against the round-1 workgroup kernel on the same tile.
usage: tools/soak_pipe.py [seconds]   -- random image sizes (incl. 1800-2100 and 4000-4200 wide rows), windows 9..63, 49..300
disparities, minDisparity 0..40, consistent on / off; every map (and every 6th case the raw cost volume) must be equal bit for bit."""
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
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "2")))
lib = _native.lib()
t_end = time.time() + budget
n = skipped = 0
HOOKS = ("SSAMD_ASW_PIPE", "SSAMD_ASW_DEPHASE", "SSAMD_ASW_STATIC", "SSAMD_ASW_EVOL", "SSAMD_ASW_GEOM")
while time.time() < t_end:
    H = int(rng.integers(1, 60))
    W = int(rng.choice([rng.integers(8, 900), rng.integers(1800, 2100), rng.integers(4000, 4200)], p=[0.6, 0.3, 0.1]))
    win = int(rng.choice([9, 11, 15, 17, 21, 27, 35, 41, 63]))
    nD = int(rng.integers(49, 301)) if rng.random() < 0.8 else int(rng.choice([193, 257, 65]))
    mind = int(rng.choice([0, 0, 0, 1, 7, 40]))
    maxd = mind + nD - 1
    cons = bool(rng.random() < 0.5)
    L, R, _ = make_pair(H, W, max(1, maxd), int(rng.integers(0, 1 << 30)))
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=win, maxDisparity=maxd, minDisparity=mind, consistent=cons,
                             gammaC=float(rng.choice([5.0, 7.0, 0.5, 50.0])), gammaP=float(rng.choice([17.5, 3.0, 100.0])))
    with_costs = n % 6 == 0 and H * W * nD < 4e7

    def costs():
        c = np.empty((H, W, nD), np.float32)
        _native.check(lib.ssamd_asw_costs(L.ctypes.data, R.ctypes.data, H, W, win, maxd, mind, m.gammaC, m.gammaP, c.ctypes.data, -1))
        return c
    try:
        form = _native.asw_kernel_form(W, H, win, maxd, mind)
        if not form["phase_shifted"]:
            skipped += 1
            continue
        got = m.compute(tL, tR)                      # the default: phase-shifted kernel as the host chooses it
        got_c = costs() if with_costs else None
        g = _native.asw_geometry(W, H, win, maxd, mind)
        _native.set_option("SSAMD_ASW_GEOM", "%d,%d,%d,8" % (g["tile_x"] // 8, g["chunk_d"] // 4, form["chunk_columns"]))
        _native.set_option("SSAMD_ASW_PIPE", "0")   # same tile and tap-column chunks, round-1 kernel
        try:
            want = m.compute(tL, tR)
        except ValueError:
            raise
        except _native.NativeError:                 # the round-1 layout of this tile does not fit LDS
            skipped += 1
            continue
        want_c = costs() if with_costs else None
        # a random variant of the phase-shifted kernel on the same tile
        var = {"SSAMD_ASW_PIPE": str(rng.choice([8, 16])), "SSAMD_ASW_DEPHASE": str(rng.integers(0, 2)),
               "SSAMD_ASW_STATIC": str(rng.integers(0, 2)), "SSAMD_ASW_EVOL": str(rng.integers(0, 2))}
        for k, v in var.items():
            _native.set_option(k, v)
        alt = m.compute(tL, tR) if _native.asw_kernel_form(W, H, win, maxd, mind)["phase_shifted"] else want
        if not (torch.equal(got, want) and torch.equal(alt, want)):
            print("MISMATCH", dict(H=H, W=W, win=win, maxd=maxd, mind=mind, cons=cons, geom=g, var=var,
                                   differing=int((got != want).sum()), differing_variant=int((alt != want).sum())))
            sys.exit(1)
        if with_costs and not np.array_equal(got_c, want_c, equal_nan=True):
            print("COST MISMATCH", dict(H=H, W=W, win=win, maxd=maxd, mind=mind))
            sys.exit(1)
        n += 1
    finally:
        for k in HOOKS:
            _native.set_option(k, None)
print("soak ok: %d random cases equal (%d skipped: not served by the phase-shifted kernel) in %.0f s" % (n, skipped, budget))
