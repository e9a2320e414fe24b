#!/usr/bin/env python3
"""GPU box, round 3: one-off probes.  (1) is the bare v_sqrt_f32 correctly rounded on every integer 0 .. 195075 (the
GSW colour distances)?  (2) host <-> device copy times of a 1080p frame pair: pageable, pinned, hipHostRegister."""
import ctypes
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simplestereo_amd import _native

lib = _native.lib()
n = 3 * 255 * 255 + 1
out = np.empty(n, np.float32)
_native.check(lib.ssamd_debug_gsw_sqrt(-n, out.ctypes.data))
want = np.sqrt(np.arange(n, dtype=np.float64)).astype(np.float32)
bad = np.flatnonzero(out != want)
res = {"v_sqrt_f32_exact_on_all_integers": bool(bad.size == 0), "mismatches": int(bad.size), "first": bad[:10].tolist()}
_native.check(lib.ssamd_debug_gsw_sqrt(n, out.ctypes.data))
res["gsw_sqrt_int_exact"] = bool(np.array_equal(out, want))

H, W = 1080, 1920
a = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
b = a.copy()
d = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
d2 = torch.empty_like(d)
o = torch.zeros((H, W), dtype=torch.int16, device="cuda")


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


ta, tb = torch.from_numpy(a), torch.from_numpy(b)
res["h2d_pageable_2x6.2MB_ms"] = t(lambda: (d.copy_(ta, non_blocking=True), d2.copy_(tb, non_blocking=True)))
pa, pb = ta.pin_memory(), tb.pin_memory()
res["h2d_pinned_2x6.2MB_ms"] = t(lambda: (d.copy_(pa, non_blocking=True), d2.copy_(pb, non_blocking=True)))
res["memcpy_to_pinned_2x6.2MB_ms"] = t(lambda: (pa.copy_(ta), pb.copy_(tb)))
ho = torch.empty((H, W), dtype=torch.int16)
po = ho.pin_memory()
res["d2h_pageable_4.1MB_ms"] = t(lambda: ho.copy_(o))
res["d2h_pinned_4.1MB_ms"] = t(lambda: po.copy_(o, non_blocking=True))
rt = torch.cuda.cudart()


def reg():
    rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); rt.cudaHostRegister(b.ctypes.data, b.nbytes, 0)
    rt.cudaHostUnregister(a.ctypes.data); rt.cudaHostUnregister(b.ctypes.data)


try:
    res["hostRegister+Unregister_2x6.2MB_ms"] = t(reg, 10)
except Exception as e:      # noqa: BLE001
    res["hostRegister"] = repr(e)[:100]
print(json.dumps(res))
