"""GPU box (round 6): the phase-shifted kernel's static instantiations with stride-only constants (SSAMD_ASW_STATIC=1, rounds 3-5) against
whole-geometry constants (=2: AswPipeTile / asw_pipe_geom_constexpr), alternating in ONE process, >= 20 launches each; maps compared."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
lib = _native.lib()
for name, H, W, maxD, n in (("c3 1080p/193", 1080, 1920, 192, 24), ("c2 480p/65", 480, 640, 64, 40), ("c5 4K/257", 2160, 4096, 256, 8)):
    L, R, _ = make_pair(H, W, maxD, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    for exact in (True, False):
        m = ss.passive.StereoASW(winSize=35, maxDisparity=maxD, exact=exact)
        times = {"1": [], "2": []}
        maps = {}
        for k in range(n + 2):
            for mode in ("1", "2"):
                with _native.options(SSAMD_ASW_STATIC=mode):
                    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
                    d = m.compute(tL, tR); torch.cuda.synchronize()
                    ms, launches = _native.profile_read(); lib.ssamd_profile_enable(0)
                if k >= 2:
                    times[mode].append(ms[_native.K_ASW_AGG])
                maps[mode] = d
        row = {"case": name, "exact": exact, "launches_each": n, "mismatch": _native.counter("static_tile_mismatch"),
               "maps_equal": bool(torch.equal(maps["1"], maps["2"]))}
        for mode, lab in (("1", "strides_only_ms"), ("2", "whole_geometry_ms")):
            t = np.array(times[mode]); row[lab] = {"mean": round(float(t.mean()), 4), "median": round(float(np.median(t)), 4), "min": round(float(t.min()), 4)}
        row["delta_percent_median"] = round(100.0 * (row["whole_geometry_ms"]["median"] / row["strides_only_ms"]["median"] - 1.0), 3)
        print(json.dumps(row), flush=True)
