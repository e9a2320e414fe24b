"""GPU box (round 5): the pixels of the bench frame on which StereoASW(exact=True) still differs from the reference's map
(tests/golden/full_cases.npz F3p), dumped with both disparities so that the build container can look at the oracle's fp64
costs of exactly those candidates (tools/exact_audit_cpu.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd.synth import make_pair
G = os.path.join(ROOT, "tests", "golden")
full = np.load(os.path.join(G, "full_cases.npz"))
L, R, _ = make_pair(1080, 1920, 192, 1)
out = {}
for cid, cons in (("F3p", False), ("F3c", True)):
    if cid not in full.files:
        continue
    p = dict(winSize=35, maxDisparity=192, consistent=cons)
    d32 = ss.passive.StereoASW(**p).compute(L, R)
    d64 = ss.passive.StereoASW(exact=True, **p).compute(L, R)
    ref = full[cid]
    ys, xs = np.nonzero(d64 != ref)
    out[cid] = {"fp32_differing": int(np.count_nonzero(d32 != ref)), "exact_differing": int(ys.size),
                "pixels": [[int(y), int(x), int(d64[y, x]), int(ref[y, x]), int(d32[y, x])] for y, x in zip(ys, xs)][:400]}
    print(cid, out[cid]["fp32_differing"], "->", out[cid]["exact_differing"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "exact_audit.json"), "w"))
