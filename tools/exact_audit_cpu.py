"""Build container (round 5): the pixels on which StereoASW(exact=True) still differs from the reference's full-frame map
(gpurun_out/exact_audit.json, written on the GPU box by tools/exact_audit.py) against the fp64 costs of the C restatement
(oracle/, bit-exact with the reference on every golden): how far apart are the costs of the two choices, and how many
candidates of the pixel are within 1e-14 of its minimum?  -> profiles/r05_exact_mode_audit.txt"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from simplestereo_amd.synth import make_pair
a = json.load(open(os.path.join(ROOT, "gpurun_out", "exact_audit.json")))
L, R, _ = make_pair(1080, 1920, 192, 1)
px = a["F3p"]["pixels"]
print("F3p (bench frame, ASW win 35, D 0..192): fp32 path %d pixels differ from the reference's map, exact mode %d (of 2 073 600)" %
      (a["F3p"]["fp32_differing"], a["F3p"]["exact_differing"]))
ys = sorted(set(p[0] for p in px))
r0, r1 = max(0, ys[0] - 17), min(1080, ys[-1] + 18)
m, c = oracle.asw(np.ascontiguousarray(L[r0:r1]), np.ascontiguousarray(R[r0:r1]), 35, 192, 0, 5, 17.5, False, hoist=True, return_costs=True)
full = np.load(os.path.join(ROOT, "tests", "golden", "full_cases.npz"))["F3p"]
lo = ys[0] - r0
print("restatement on rows %d..%d (with halo) equals the golden rows: %s" % (ys[0], ys[-1], np.array_equal(m[lo:lo + ys[-1] - ys[0] + 1], full[ys[0]:ys[-1] + 1])))
print("%5s %5s %6s %6s  %-22s %-22s %9s %s" % ("y", "x", "d_gpu", "d_ref", "fp64 cost at d_gpu", "fp64 cost at d_ref", "rel diff", "candidates within 1e-14 of the minimum / all"))
worst = 0.0
for y, x, dg, dr, d32 in px:
    row = c[y - r0, x]
    n = min(x, 192) + 1
    cg, cr = row[dg], row[dr]
    worst = max(worst, abs(cg - cr) / cr)
    print("%5d %5d %6d %6d  %-22.17g %-22.17g %9.2g %d / %d" % (y, x, dg, dr, cg, cr, (cg - cr) / cr, int(np.sum(np.abs(row[:n] - np.nanmin(row[:n])) <= 1e-14 * 40)), n))
print("largest relative difference between the two choices: %.3g (one ulp of 40.0 is 1.78e-16)" % worst)
