#!/usr/bin/env python3
"""GPU box (round 5): what the overlapped strip step pays in kernel time for its two aggregation launches -- the slowest (interior)
rank's strip of an N-GPU run of config 3 / 5, on ONE GPU: one launch over the strip against the interior rows (cut to whole rounds
of workgroups by StripContext) + the border bands (one two-range launch with its half-width last round); and against the naive
cut (every halo-free row in the interior launch, SSAMD_STRIP_OVERLAP=force).  kernel ms = library HIP events, all launches of a step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native, strips
from simplestereo_amd.synth import make_pair

lib = _native.lib()


def timed(fn, reps=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n = _native.profile_read(); lib.ssamd_profile_enable(0)
    return (ms[_native.K_ASW_AGG] + ms[_native.K_LAB]) / reps, n[_native.K_ASW_AGG] / reps


for name, (H, W, maxd) in (("config 3 1920x1080 D 0..192", (1080, 1920, 192)), ("config 5 4096x2160 D 0..256", (2160, 4096, 256))):
    L, R, _ = make_pair(H, W, maxd, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=35, maxDisparity=maxd)
    print(name)
    for world in (2, 4, 8):
        rank = world // 2
        res = {}
        for mode in ("1", "force"):
            os.environ["SSAMD_STRIP_OVERLAP"] = mode
            ctx = strips.StripContext(m, H, W, rank, world, torch.device("cuda", 0))
            a, b = tL[ctx.h0:ctx.h1].contiguous(), tR[ctx.h0:ctx.h1].contiguous()
            o0, n = ctx.r0 - ctx.h0, ctx.r1 - ctx.r0
            out = torch.empty((n, W), dtype=torch.int16, device="cuda")
            one, l1 = timed(lambda: m._compute_device(a, b, out_row0=o0, out_rows=n, out=out))
            want = out.clone()

            def two():
                if ctx.interior > 0:
                    m._compute_device(a, b, out_row0=o0 + ctx.top, out_rows=ctx.interior, out=out[ctx.top:ctx.top + ctx.interior])
                m._compute_device(a, b, out_row0=o0, out_rows=n, out=out, skip=(o0 + ctx.top, ctx.interior))
            out.fill_(-5)
            t2, l2 = timed(two)
            assert torch.equal(out, want)
            res[mode] = (one, t2, l2, ctx.top, ctx.interior, ctx.bot)
        one, t2, l2, top, it, bot = res["1"]
        _, t2f, l2f, topf, itf, botf = res["force"]
        print("  world %d rank %d, strip of %d rows: one launch %.3f ms | whole-round cut %d + %d + %d rows: %.3f ms (%+.1f %%, %.0f launches) | naive cut %d + %d + %d rows: %.3f ms (%+.1f %%)" %
              (world, rank, top + it + bot, one, top, it, bot, t2, 100 * (t2 / one - 1), l2, topf, itf, botf, t2f, 100 * (t2f / one - 1)))
