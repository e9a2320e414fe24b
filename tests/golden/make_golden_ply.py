#!/usr/bin/env python3
"""Golden PLY files written by the UNMODIFIED reference `simplestereo/points.py` (exportPLY, points.py:10-80).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_ply.py

`points.py` does `import cv2` at module level although exportPLY / importPLY never touch it, and cv2 is not
installed here; the module is therefore loaded by path with an EMPTY placeholder module registered under the name
cv2 for the duration of the import.  Nothing of the reference is modified or copied, and the cv2-backed function of
that module (getAdimensional3DPoints -> cv2.reprojectImageTo3D) is NOT used: it stays "parity unpinned".

Output: tests/golden/ply_cases.npz (the inputs) and tests/golden/ply_*.ply (the reference's files, as data).
tests/test_rigs_cpu.py writes the same inputs with simplestereo_amd.points.exportPLY and compares byte for byte,
and reads the reference's files with importPLY.
"""
import importlib.util
import os
import sys
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_points():
    placeholder = "cv2" not in sys.modules
    if placeholder:
        sys.modules["cv2"] = types.ModuleType("cv2")
    try:
        spec = importlib.util.spec_from_file_location("ref_points", "/root/reference/simplestereo/points.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if placeholder:
            del sys.modules["cv2"]
    return mod


def main():
    ref = load_reference_points()
    rng = np.random.default_rng(7)
    pts = np.round(rng.normal(scale=50.0, size=(3, 4, 3)), 5)
    pts[0, 0] = [0.0, -0.0000004, 1234567.125]
    bgr = rng.integers(0, 256, (3, 4, 3)).astype(np.uint8)
    gray_u8 = rng.integers(0, 256, (3, 4)).astype(np.uint8)
    gray_i64 = gray_u8.astype(np.int64) * 3 - 100
    gray_f32 = (gray_u8 / 7.0).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ply_cases.npz"), pts=pts, bgr=bgr, gray_u8=gray_u8, gray_i64=gray_i64,
                        gray_f32=gray_f32)
    flat = pts.reshape(-1, 3)
    cases = {"ply_none_p6.ply": (None, 6), "ply_bgr_p4.ply": (bgr, 4), "ply_gray_i64_p3.ply": (gray_i64, 3),
             "ply_gray_u8_p6.ply": (gray_u8, 6), "ply_gray_f32_p2.ply": (gray_f32, 2)}
    for name, (img, prec) in cases.items():
        # the reference iterates `for x,y,z in points3D` / indexes points3D[i,0]: it needs the flattened (n, 3) array
        ref.exportPLY(flat, os.path.join(OUT, name), img, precision=prec)
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")
    # importPLY of the reference on its own file, for the record
    print(ref.importPLY(os.path.join(OUT, "ply_bgr_p4.ply"), 0, 1, 2, 3)[:2])


if __name__ == "__main__":
    main()
