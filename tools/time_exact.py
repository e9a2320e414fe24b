"""GPU box (round 5): cost of the fp64 tie-break pass (StereoASW(exact=True)) next to the plain call -- kernel ms per slot,
candidates re-evaluated, pixels flagged -- and the map against the full-frame reference golden where one is committed."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair

G = os.path.join(ROOT, "tests", "golden")
full = np.load(os.path.join(G, "full_cases.npz")) if os.path.exists(os.path.join(G, "full_cases.npz")) else None
lib = _native.lib()
for name, H, W, maxD, win, cons, gold in [("c3", 1080, 1920, 192, 35, False, "F3p"), ("c3 consistent", 1080, 1920, 192, 35, True, "F3c"),
                                          ("c2", 480, 640, 64, 35, False, None), ("default 1080p D16", 1080, 1920, 16, 35, False, None),
                                          ("tsukuba-size D16 win15", 288, 384, 16, 15, False, None)]:
    L, R, _ = make_pair(H, W, maxD, 1 if H == 1080 and maxD == 192 else 0)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    row = {"case": name}
    maps = {}
    for exact in (False, True):
        m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, consistent=cons, exact=exact)
        d = m.compute(tL, tR); torch.cuda.synchronize()
        lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
        t = time.perf_counter()
        for _ in range(3):
            d = m.compute(tL, tR)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / 3 * 1e3
        ms, n = _native.profile_read(); lib.ssamd_profile_enable(0)
        key = "exact" if exact else "fp32"
        row[key + "_ms_per_call"] = round(wall, 3)
        row[key + "_kernels_ms"] = {lib.ssamd_kernel_name(i).decode()[:28]: round(ms[i] / 3, 3) for i in range(_native.K_COUNT) if n[i]}
        maps[key] = d.cpu().numpy()
    row["entries"] = _native.counter("exact_entries"); row["flagged_left"] = _native.counter("exact_flagged_left")
    row["flagged_right"] = _native.counter("exact_flagged_right"); row["overflow"] = _native.counter("exact_overflow")
    row["pixels_changed_by_exact"] = int(np.count_nonzero(maps["fp32"] != maps["exact"]))
    if full is not None and gold in (full.files if full is not None else []):
        for key in ("fp32", "exact"):
            diff = np.abs(maps[key].astype(np.int32) - full[gold].astype(np.int32))
            row[key + "_vs_reference"] = {"differing": int(np.count_nonzero(diff)), "bad1": int(np.count_nonzero(diff > 1)), "pixels": int(diff.size)}
    print(json.dumps(row), flush=True)
