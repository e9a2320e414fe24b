"""GPU box (round 5): StereoASW(exact=True) at config-5 size (4096 x 2160, D 0..256: a 9.1 GB cost-image volume) -- time, counters, and
the known-shift property (a right image = the left one shifted by 37 columns: the map is 37 on the interior, exact or not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
H, W, maxD = 2160, 4096, 256
L, R, _ = make_pair(H, W, maxD, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
for exact in (False, True):
    m = ss.passive.StereoASW(winSize=35, maxDisparity=maxD, exact=exact)
    d = m.compute(tL, tR); torch.cuda.synchronize()
    t = time.perf_counter(); d = m.compute(tL, tR); torch.cuda.synchronize()
    print("exact=%s: %.1f ms per call, checksum %d" % (exact, (time.perf_counter() - t) * 1e3, int(d.long().sum())), flush=True)
    if exact:
        print("   candidates re-evaluated %d, pixels flagged %d, overflow %d, pixels changed %d, free memory %.1f GB" %
              (_native.counter("exact_entries"), _native.counter("exact_flagged_left"), _native.counter("exact_overflow"),
               int((d != d0).sum()), torch.cuda.mem_get_info()[0] / 2**30))
    d0 = d
Ls = torch.from_numpy(np.ascontiguousarray(L)).cuda()
Rs = torch.roll(Ls, -37, dims=1).contiguous()
d = ss.passive.StereoASW(winSize=35, maxDisparity=maxD, exact=True).compute(Ls, Rs)
inner = d[:, maxD + 40: W - 80]
print("known shift 37 with exact=True: %.4f %% of the interior" % (100.0 * float((inner == 37).float().mean())))
