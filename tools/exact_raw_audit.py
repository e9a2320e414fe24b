"""GPU box (round 6): what the RAW near-tie queue of a merging exact call holds (consistent=True, config 3) -- entries by side, by
cost class, per right pixel -- to see what the tile-local selection passes that the final winners then reject."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair

H, W, maxD = 1080, 1920, 192
L, R, _ = make_pair(H, W, maxD, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
m = ss.passive.StereoASW(winSize=35, maxDisparity=maxD, consistent=True, exact=True)
d = m.compute(tL, tR); torch.cuda.synchronize()
lib = _native.lib()
for which in (1, 0):
    n = ctypes.c_longlong(0)
    cap = 40_000_000
    ents = np.empty(cap, np.uint64); keys = np.empty(cap, np.uint32)
    _native.check(lib.ssamd_debug_exact_queue(which, cap, ents.ctypes.data, keys.ctypes.data, ctypes.byref(n)))
    k = min(n.value, cap)
    ents, keys = ents[:k], keys[:k]
    pix = (ents & 0xffffffff).astype(np.int64); dd = ((ents >> 32) & 0xffff).astype(np.int64); sides = ((ents >> 48) & 3).astype(np.int64)
    y, x = pix // W, pix % W
    print("queue %d: %d entries; sides L %d R %d both %d" % (which, n.value, np.sum(sides == 1), np.sum(sides == 2), np.sum(sides == 3)))
    if which == 0 or k == 0:
        continue
    HIGH = 0xC0000000 - 0x41A00000
    hi = keys >= HIGH
    inv = (np.uint32(0xC0000000) - keys[hi]).view(np.float32)
    cost_lo = keys[~hi].view(np.float32)
    print("  saturated-side images (cost > 20): %d (%.1f %%); of these 40 - cost: ==0: %d, <1e-10: %d, <1e-6: %d, <1e-3: %d, <1: %d" %
          (hi.sum(), 100.0 * hi.mean(), np.sum(inv == 0), np.sum(inv < 1e-10), np.sum(inv < 1e-6), np.sum(inv < 1e-3), np.sum(inv < 1)))
    if cost_lo.size:
        print("  cost <= 20 images: %d, cost quantiles %s" % (cost_lo.size, np.quantile(cost_lo, [0, .1, .5, .9, 1]).round(4).tolist()))
    r_only = sides >= 2
    xr = x - dd
    rp = (y * W + xr)[r_only]
    u, c = np.unique(rp, return_counts=True)
    print("  right pixels with raw R entries: %d, entries per pixel quantiles %s" % (u.size, np.quantile(c, [0, .5, .9, .99, 1]).tolist()))
    # where are they: histogram over columns of xr and over d
    print("  xr column histogram (10 bins):", np.histogram(xr[r_only], bins=10, range=(0, W))[0].tolist())
    print("  d histogram (8 bins):", np.histogram(dd[r_only], bins=8, range=(0, maxD + 1))[0].tolist())
    print("  rows histogram (10 bins):", np.histogram(y[r_only], bins=10, range=(0, H))[0].tolist())
    # is the right pixel in a disocclusion hole of the right image?  (hole filler = uniform noise: large local variance)
    Rf = R.astype(np.float32).sum(2)
    lap = np.abs(Rf[:, 1:-1] * 2 - Rf[:, :-2] - Rf[:, 2:])
    noisy = np.zeros((H, W), bool); noisy[:, 1:-1] = lap > 150
    print("  fraction of raw-R right pixels that look like hole noise: %.3f (frame: %.3f)" % (noisy.reshape(-1)[u].mean(), noisy.mean()))
