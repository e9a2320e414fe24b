import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
L, R, _ = make_pair(1080, 1920, 192, 1)
L[:, :160] = 0; R[:, :160] = 0; L[:40] = 0; R[:40] = 0; L[-40:] = 0; R[-40:] = 0; L[:, -100:] = 0; R[:, -100:] = 0
tL, tR = torch.from_numpy(np.ascontiguousarray(L)).cuda(), torch.from_numpy(np.ascontiguousarray(R)).cuda()
for exact in (False, True):
    for cons in (False, True):
        m = ss.passive.StereoASW(winSize=35, maxDisparity=192, exact=exact, consistent=cons)
        m.compute(tL, tR); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): m.compute(tL, tR)
        torch.cuda.synchronize()
        print("1080p with black margins (160 / 100 columns, 40 rows): exact=%s consistent=%s %.2f ms per call; entries %d raw %d overflow %d" % (
            exact, cons, (time.perf_counter() - t) / 3 * 1e3, _native.counter("exact_entries") if exact else 0,
            _native.counter("exact_raw_entries") if exact else 0, _native.counter("exact_overflow") if exact else 0))
