#!/usr/bin/env python3
"""GPU box: the SAME workloads on two source trees of the package (e.g. the previous round's, exported with
`git archive <commit> simplestereo_amd include | tar -x -C tools/_exp/r02/` and built there, against the working tree),
each in its own process, alternating, 30+ calls per timing -- box-to-box differences of the pool (several percent) make
numbers from different gpurun calls incomparable.   usage: tools/ab_tree.py <other_tree_root> [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, json, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
sys.path.insert(0, sys.argv[2])
from simplestereo_amd.synth import make_pair
lib = _native.lib()
res = {}
L, R, _ = make_pair(1080, 1920, 192, 1)
tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
jobs = [("gsw_c4", ss.passive.StereoGSW(winSize=11, maxDisparity=192), 30), ("asw_c3", ss.passive.StereoASW(winSize=35, maxDisparity=192), 8),
        ("asw_d16", ss.passive.StereoASW(winSize=35, maxDisparity=16), 40), ("asw_d7", ss.passive.StereoASW(winSize=35, maxDisparity=7), 40),
        ("asw_d64", ss.passive.StereoASW(winSize=35, maxDisparity=64), 15)]
for name, m, n in jobs:
    for _ in range(3): d = m.compute(tL, tR)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): d = m.compute(tL, tR)
    torch.cuda.synchronize()
    res[name] = {"wall_ms": (time.perf_counter() - t0) / n * 1e3, "checksum": int(d.long().sum())}
print(json.dumps(res))
"""


def main():
    others = [a for a in sys.argv[1:] if not a.isdigit()]
    if len(others) > 1:
        return main_many(others, int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2)
    other = os.path.abspath(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    acc = {}
    for r in range(rounds):
        for label, tree in (("other", other), ("this", ROOT)):
            p = subprocess.run([sys.executable, "-c", WORKER, tree, ROOT], capture_output=True, text=True)
            if p.returncode:
                print(label, "FAILED", p.stderr[-1500:]); continue
            for k, v in json.loads(p.stdout.strip().splitlines()[-1]).items():
                acc.setdefault(k, {}).setdefault(label, []).append(v)
    print("%-10s %12s %12s   (wall ms per call, best of %d rounds; checksums equal?)" % ("workload", "other", "this", rounds))
    for k, v in acc.items():
        o = min(x["wall_ms"] for x in v.get("other", [{"wall_ms": float("nan")}])); t = min(x["wall_ms"] for x in v.get("this", [{"wall_ms": float("nan")}]))
        same = len({x["checksum"] for lab in v.values() for x in lab}) == 1
        print("%-10s %12.3f %12.3f   %+.1f %%  %s" % (k, o, t, 100 * (t / o - 1), "same map checksum" if same else "CHECKSUMS DIFFER"))


def main_many(trees, rounds):
    """several trees (label=path ...) + the working tree, alternating"""
    specs = [(t.split("=", 1)[0], os.path.abspath(t.split("=", 1)[1])) for t in trees] + [("this", ROOT)]
    acc = {}
    for r in range(rounds):
        for label, tree in specs:
            p = subprocess.run([sys.executable, "-c", WORKER, tree, ROOT], capture_output=True, text=True)
            if p.returncode:
                print(label, "FAILED", p.stderr[-800:]); continue
            for k, v in json.loads(p.stdout.strip().splitlines()[-1]).items():
                acc.setdefault(k, {}).setdefault(label, []).append(v["wall_ms"])
    print("%-10s" % "workload" + "".join("%12s" % l for l, _ in specs) + "   (wall ms per call, best of %d rounds)" % rounds)
    for k, v in acc.items():
        print("%-10s" % k + "".join("%12.3f" % min(v.get(l, [float("nan")])) for l, _ in specs))


if __name__ == "__main__":
    main()
