#!/usr/bin/env python3
"""GPU box: host-side cost of one operator call.  On a 32 x 32 pair the kernels take a few microseconds, so a loop of
asynchronous calls is bound by the CPU: wall / call = Python + ctypes + library host path + launches."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import simplestereo_amd as ss
from simplestereo_amd import _native
from simplestereo_amd.synth import make_pair

lib = _native.lib()
res = {}
for name, (H, W, maxD, win) in {"tiny_32x32_d4_w5": (32, 32, 4, 5), "tsukuba_384x288_d16_w15": (288, 384, 16, 15)}.items():
    L, R, _ = make_pair(H, W, maxD, 1)
    tL, tR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = ss.passive.StereoASW(winSize=win, maxDisparity=maxD)
    out = torch.empty((H, W), dtype=torch.int16, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def py_call():
        return m.compute(tL, tR)

    def c_call():
        _native.check(lib.ssamd_asw_device(tL.data_ptr(), tR.data_ptr(), H, W, 0, H, win, maxD, 0, 5.0, 17.5, 0, out.data_ptr(), stream))

    def host_call():
        return m.compute(L, R)
    r = {}
    for label, fn, n in (("python_compute_device_tensors", py_call, 300), ("ctypes_ssamd_asw_device_only", c_call, 300),
                         ("python_compute_host_arrays", host_call, 100)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        r[label] = {"issue_us_per_call": 1e6 * t_issue / n, "wall_us_per_call": 1e6 * t_all / n}
    lib.ssamd_profile_enable(1); lib.ssamd_profile_reset()
    for _ in range(20):
        py_call()
    torch.cuda.synchronize()
    ms, n = _native.profile_read(); lib.ssamd_profile_enable(0)
    r["kernel_us_per_call"] = {lib.ssamd_kernel_name(i).decode()[:40]: 1e3 * ms[i] / 20 for i in range(_native.K_COUNT) if n[i]}
    res[name] = r
print(json.dumps(res, indent=1))
