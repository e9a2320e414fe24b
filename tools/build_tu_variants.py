#!/usr/bin/env python3
"""Build container: libssamd variants that differ in the compile flags of ONE translation unit (the others as in build.py).
usage: tools/build_tu_variants.py <unit: pipe|wave6|api> name1:"flag flag" name2:"..."   -> tools/_exp/libssamd_<name>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplestereo_amd import build as B
unit = {"pipe": "asw_pipe_tu.hip", "wave6": "asw_wave6_tu.hip", "api": "ssamd_api.hip"}[sys.argv[1]]
out = os.path.join(ROOT, "tools", "_exp"); os.makedirs(out, exist_ok=True)
flags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
base = {}
for src, extra in B.UNITS:
    if src != unit:
        o = os.path.join(out, os.path.splitext(src)[0] + ".o")
        subprocess.check_call(["hipcc"] + flags + extra + ["-c", "-o", o, os.path.join(B._SRC, src)])
        base[src] = o
procs = []
for spec in sys.argv[2:]:
    name, _, fl = spec.partition(":")
    o = os.path.join(out, "unit_%s.o" % name)
    procs.append((name, o, subprocess.Popen(["hipcc"] + flags + fl.split() + ["-c", "-o", o, os.path.join(B._SRC, unit)])))
for name, o, p in procs:
    if p.wait() != 0:
        print("FAILED", name); continue
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libssamd_%s.so" % name), o] + list(base.values()))
    print("built", name)
