#!/usr/bin/env python3
"""A/B of the half-width last round (SSAMD_ASW_TAIL) on the strips an N-GPU run of config 3 / config 5 gives its slowest rank:
kernel ms per strip with the split off / on, measured on ONE GPU (strip + halo resident), alternating, and the compute-side
strong scaling that follows.  python tools/strip_tail_ab.py > profiles/r04_strip_tail_ab.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplestereo_amd as ss                     # noqa: E402
from simplestereo_amd import _native, strips      # noqa: E402
from simplestereo_amd.synth import make_pair      # noqa: E402


def strip_ms(m, tL, tR, H, world, rank, reps):
    r0, r1 = strips.strip_bounds(H, world, rank)
    h0, h1 = strips.halo_bounds(H, r0, r1, m.winSize // 2)
    a, b = tL[h0:h1].contiguous(), tR[h0:h1].contiguous()
    lib = _native.lib()
    for _ in range(2):
        m._compute_device(a, b, out_row0=r0 - h0, out_rows=r1 - r0)
    torch.cuda.synchronize()
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    for _ in range(reps):
        m._compute_device(a, b, out_row0=r0 - h0, out_rows=r1 - r0)
    torch.cuda.synchronize()
    ms, launches = _native.profile_read(); lib.ssamd_profile_enable(0)
    return ms[_native.K_ASW_AGG] / reps


for name, (H, W, maxd) in (("config 3 1920x1080 D 0..192", (1080, 1920, 192)), ("config 5 4096x2160 D 0..256", (2160, 4096, 256))):
    L, R, _ = make_pair(H, W, maxd, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=35, maxDisparity=maxd)
    base = {}
    for world in (1, 2, 4, 8):
        rank = world // 2          # an interior rank: full halo on both sides (the slowest kind)
        reps = 6 if H * W < 4e6 else 3
        res = {}
        for rnd in range(2):
            for tail in ("0", "-1"):
                with _native.options(SSAMD_ASW_TAIL=tail):
                    n0 = _native.counter("tail_splits")
                    t = strip_ms(m, tL, tR, H, world, rank, reps)
                    res.setdefault(tail, []).append((t, _native.counter("tail_splits") - n0))
        off = min(t for t, _ in res["0"]); on = min(t for t, _ in res["-1"]); split = res["-1"][0][1] > 0
        base.setdefault("off", off if world == 1 else base.get("off")); base.setdefault("on", on if world == 1 else base.get("on"))
        print("%s  world %d  rows %d: split off %.3f ms, host's choice %.3f ms (%s)  ->  compute-side scaling %.2fx / %.2fx" %
              (name, world, strips.strip_bounds(H, world, rank)[1] - strips.strip_bounds(H, world, rank)[0], off, on,
               "split" if split else "no split", base["off"] / off, base["on"] / on), flush=True)
