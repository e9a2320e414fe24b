#!/usr/bin/env python3
"""GPU box: A/B of ASW kernel variants selected by environment variables (each variant in its own process, because launch
geometries are cached per shape).  usage: tools/ab_asw.py [--quick] [--only=case,..] "name=ENV1=v;ENV2=v" ...   (name "base" = no env)
Prints kernel ms per configuration and checks that every variant's maps equal the first variant's bit for bit."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # name, H, W, maxD, minD, win, consistent
    ("c3", 1080, 1920, 192, 0, 35, False), ("c3c", 1080, 1920, 192, 0, 35, True), ("c5", 2160, 4096, 256, 0, 35, False),
    ("c2", 480, 640, 64, 0, 35, False), ("d16", 1080, 1920, 16, 0, 35, False), ("d7", 1080, 1920, 7, 0, 35, False),
    ("d32", 1080, 1920, 32, 0, 35, False), ("d64", 1080, 1920, 64, 0, 35, False), ("d128w21", 1080, 1920, 128, 0, 21, False),
    ("tsu", 288, 384, 16, 0, 15, False), ("d100w17c", 600, 800, 100, 3, 17, True), ("d40w25", 300, 1000, 40, 0, 25, False),
    ("d16c", 1080, 1920, 16, 0, 35, True), ("d16w11", 1080, 1920, 16, 0, 11, False), ("d3", 1080, 1920, 3, 0, 35, False),
    ("d24m5c", 500, 777, 24, 5, 21, True), ("d1", 300, 400, 1, 1, 9, True),
]

WORKER = r"""
import sys, json, time, os
sys.path.insert(0, %r)
import numpy as np, torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair
cases = json.loads(sys.argv[1]); out = sys.argv[2]
lib = _native.lib()
res, maps = {}, {}
for name, H, W, maxD, minD, win, cons in cases:
    L, R, _ = make_pair(H, W, maxD, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD, minDisparity=minD, consistent=cons)
    d = m.compute(tL, tR); torch.cuda.synchronize()
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    n = 3 if H * W * (maxD - minD + 1) > 3e8 else 6
    for _ in range(n): d = m.compute(tL, tR)
    torch.cuda.synchronize()
    ms, launches = _native.profile_read(); lib.ssamd_profile_enable(0)
    res[name] = {"ms": ms[_native.K_ASW_AGG] / max(1, launches[_native.K_ASW_AGG]), "geom": _native.asw_geometry(W, H, win, maxD, minD)}
    maps[name] = d.cpu().numpy()
np.savez(out, **maps)
print(json.dumps(res))
""" % ROOT


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    quick = "--quick" in sys.argv
    cases = [c for c in CASES if not quick or c[0] in ("c3", "c2", "d16", "tsu", "d100w17c")]
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    if only:
        cases = [c for c in CASES if c[0] in only[0]]
    for a in sys.argv[1:]:
        if a.startswith("--cases="):          # --cases=name:H:W:maxD:minD:win:consistent(0/1),...
            cases = [(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), bool(int(f[6])))
                     for f in (c.split(":") for c in a.split("=", 1)[1].split(","))]
    variants = []
    for a in args or ["base"]:
        name, _, envs = a.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(";") if kv) if envs else {}      # name=ENV1=v;ENV2=v
        variants.append((name, env))
    import numpy as np
    results, ref = {}, None
    tmp = tempfile.mkdtemp()
    for name, env in variants:
        e = dict(os.environ); e.update(env)
        if "SSAMD_LIB" in env:
            e["SSAMD_EXPERIMENT"] = "1"          # the binding refuses another build of the library without it
        out = os.path.join(tmp, name + ".npz")
        p = subprocess.run([sys.executable, "-c", WORKER, json.dumps(cases), out], env=e, capture_output=True, text=True)
        if p.returncode != 0:
            print(name, "FAILED", p.stderr[-2000:]); continue
        results[name] = json.loads(p.stdout.strip().splitlines()[-1])
        maps = np.load(out)
        if ref is None:
            ref = {k: maps[k] for k in maps.files}
        else:
            for k in maps.files:
                same = np.array_equal(maps[k], ref[k])
                results[name][k]["equal_first"] = bool(same)
                if not same:
                    results[name][k]["differing_pixels"] = int(np.count_nonzero(maps[k] != ref[k]))
    names = [c[0] for c in cases]
    print("%-10s" % "case" + "".join("%16s" % v for v in results))
    for k in names:
        row = "%-10s" % k
        for v in results:
            r = results[v].get(k, {})
            row += "%12.3f%4s" % (r.get("ms", float("nan")), "" if r.get("equal_first", True) else " !!")
        print(row)
    if "--json" in sys.argv:
        print(json.dumps(results))
    else:      # geometry of every variant, compactly
        for v in results:
            print(v, {k: (r["geom"]["tile_x"], r["geom"]["chunk_d"], r["geom"]["threads"], r["geom"]["lds_bytes"]) for k, r in results[v].items()})


if __name__ == "__main__":
    main()
