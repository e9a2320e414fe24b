"""debug helper (GPU box): run every golden case on the GPU and save the maps to gpurun_out/gpu_maps.npz"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
G = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(G, "cases.json")))
ts = np.load(os.path.join(G, "tsukuba_pair.npz"))
L, R = np.ascontiguousarray(ts["left"]), np.ascontiguousarray(ts["right"])
inputs = {"tsukuba": (L, R), "crop": (np.ascontiguousarray(L[100:118, 150:190]), np.ascontiguousarray(R[100:118, 150:190])),
          "tsukuba_top": (np.ascontiguousarray(L[:40]), np.ascontiguousarray(R[:40])),
          "synth_96x128": make_pair(96, 128, 32, 0)[:2], "synth_64x96": make_pair(64, 96, 24, 5)[:2],
          "synth_480x640": make_pair(480, 640, 64, 0)[:2]}
out = {}
for cid, m in meta.items():
    p = dict(m["params"]); algo = p.pop("algo")
    a, b = inputs[m["input"]]
    try:
        if algo == "asw":
            out[cid] = ss.passive.StereoASW(**p).compute(a, b)
            if p.get("consistent"):
                q = dict(p); q["consistent"] = False
                out[cid + "_left"] = ss.passive.StereoASW(**q).compute(a, b)
        else:
            out[cid] = ss.passive.StereoGSW(**p).compute(a, b)
    except Exception as e:
        print(cid, "failed:", e)
a, b = inputs["synth_96x128"]
c = np.empty((96, 128, 33), np.float32)
_native.check(_native.lib().ssamd_asw_costs(a.ctypes.data, b.ctypes.data, 96, 128, 35, 32, 0, 5.0, 17.5, c.ctypes.data, -1))
out["G6_costs"] = c
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "gpu_maps.npz"), **out)
print("saved", sorted(out))
