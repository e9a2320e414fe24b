#!/usr/bin/env python3
"""GPU box (round 5): randomised soak of what round 5 added.  usage: tools/soak_r05.py [seconds]
Per random case (sizes 8..900 wide, 4..70 rows, windows 3..35, 1..120 disparities, plain / consistent, random gammas):
  1. two row ranges in one launch (ssamd_asw_device_rows2) + an ordinary call for the rows in between == one launch over the range;
  2. Lab records + TAD volume in one launch == the two dependent launches (SSAMD_ASW_PREPASS_FUSE=0);
  3. StereoASW(exact=True) == the fp64 C restatement's map (oracle/, bit-exact with the reference on every golden), every pixel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "5")))
t_end = time.time() + budget
n = n_exact = n_rows2 = flagged = n_bad = 0
while time.time() < t_end:
    H = int(rng.integers(4, 70)); W = int(rng.integers(8, 900))
    win = int(rng.choice([3, 5, 9, 11, 15, 21, 27, 35]))
    nD = int(rng.integers(1, 121)); mind = int(rng.choice([0, 0, 1, 5])); maxd = mind + nD - 1
    cons = bool(rng.random() < 0.5)
    gc, gp = float(rng.choice([5.0, 7.0, 0.7, 50.0])), float(rng.choice([17.5, 3.0, 100.0]))
    style = rng.random()
    L, R, _ = make_pair(H, W, max(1, maxd), int(rng.integers(0, 1 << 30)))
    if style < 0.25:                      # flat / saturated content: ties everywhere
        L = (L // 64 * 64).astype(np.uint8); R = np.ascontiguousarray(R[:, ::-1])
    L, R = np.ascontiguousarray(L), np.ascontiguousarray(R)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    p = dict(winSize=win, maxDisparity=maxd, minDisparity=mind, consistent=cons, gammaC=gc, gammaP=gp)
    m = ss.passive.StereoASW(**p)
    whole = m.compute(tL, tR)
    # 1. rows2
    if H >= 6:
        r0 = int(rng.integers(0, H // 3)); rows = int(rng.integers(3, H - r0 + 1))
        s0 = r0 + int(rng.integers(0, rows)); ns = int(rng.integers(0, r0 + rows - s0 + 1))
        out = torch.full((rows, W), -9, dtype=torch.int16, device="cuda")
        m._compute_device(tL, tR, out_row0=r0, out_rows=rows, out=out, skip=(s0, ns))
        if ns:
            m._compute_device(tL, tR, out_row0=s0, out_rows=ns, out=out[s0 - r0:s0 - r0 + ns])
        assert torch.equal(out, whole[r0:r0 + rows]), ("rows2", H, W, p, r0, rows, s0, ns)
        n_rows2 += 1
    # 2. fused pre-pass
    with _native.options(SSAMD_ASW_PREPASS_FUSE="0"):
        assert torch.equal(m.compute(tL, tR), whole), ("prepass", H, W, p)
    # 3. exact == oracle (bounded CPU time)
    if H * W * nD * win * win < 6e8:
        d = ss.passive.StereoASW(exact=True, **p).compute(L, R)
        ref = oracle.asw(L, R, **p)
        if _native.counter("exact_overflow") == 0:
            if not np.array_equal(d, ref):
                n_bad += 1
                print("EXACT MISMATCH", H, W, p, int(np.count_nonzero(d != ref)), "style", style)
                if not cons:
                    _, cref = oracle.asw(L, R, return_costs=True, **p)
                    c32 = np.empty((H, W, nD), np.float32)
                    _native.check(_native.lib().ssamd_asw_costs(L.ctypes.data, R.ctypes.data, H, W, win, maxd, mind, gc, gp, c32.ctypes.data, -1))
                    d32 = m.compute(L, R)
                    for y, x in np.argwhere(d != ref)[:4]:
                        kg, kr, k32 = int(d[y, x]) - mind, int(ref[y, x]) - mind, int(d32[y, x]) - mind
                        f, g = c32[y, x], cref[y, x]
                        print("   (%d,%d) exact d=%d ref d=%d fp32 d=%d | fp64: at exact's %.17g at ref's %.17g at fp32's %.17g | fp32: at exact's %.9g at ref's %.9g at fp32's %.9g | rel err of fp32 at ref's %.3g, at fp32's %.3g" %
                              (y, x, d[y, x], ref[y, x], d32[y, x], g[kg], g[kr], g[k32], f[kg], f[kr], f[k32], (f[kr] - g[kr]) / max(g[kr], 1e-300), (f[k32] - g[k32]) / max(g[k32], 1e-300)))
            n_exact += 1
            flagged += _native.counter("exact_flagged_left")
    n += 1
print("soak_r05: %d cases (%d two-range cuts ok, %d exact-mode maps against the fp64 oracle: %d NOT identical, %d pixels tie-broken) in %.0f s" % (n, n_rows2, n_exact, n_bad, flagged, budget))
