"""Test helpers shared by the CPU and GPU tiers: literal Python restatements of the reference's left-right check /
occlusion filling and of its two winner-take-all scans, driven by a cost volume."""
import numpy as np


def lr_check_fill_literal(left, right_match):
    """literal restatement of _passive.cpp:250-285 on one map: sequential invalidation by the right-referenced matches,
    then every run of invalid pixels takes min(left neighbour, right neighbour) / the single neighbour at a border
    (a fully invalid row keeps -1: the reference reads out of bounds there)"""
    H, W = left.shape
    out = left.astype(np.int32).copy()
    for y in range(H):
        row = out[y]
        for x in range(W):
            b = int(right_match[y, x])
            if row[b] != b - x:
                row[b] = -1
        for j in range(W):
            if row[j] == -1:
                lo, hi = j - 1, j + 1
                while lo >= 0 and row[lo] == -1:
                    lo -= 1
                while hi < W and row[hi] == -1:
                    hi += 1
                if lo < 0 and hi > W - 1:
                    break
                if lo < 0:
                    row[:hi] = row[hi]
                elif hi > W - 1:
                    row[lo + 1:] = row[lo]
                else:
                    row[lo + 1:hi] = min(row[lo], row[hi])
    return out.astype(np.int16)


def oracle_argmins(cref, minD, maxD):
    """both winner-take-all results from the oracle's fp64 left-referenced costs: the right-referenced cost of the
    pair (xr, xl) is the same number as the left-referenced cost of (xl, d = xl - xr) (_passive.cpp:160-170 vs
    225-235); first minimum wins in both scans (smallest disparity / smallest left column)"""
    H, W, nD = cref.shape
    left = np.empty((H, W), np.int32)
    right = np.zeros((H, W), np.int32)
    for y in range(H):
        for x in range(W):
            row = cref[y, x]
            ok = np.isfinite(row)
            left[y, x] = (minD + int(np.argmin(np.where(ok, row, np.inf)))) if ok.any() else x
            cand = [(cref[y, xl, xl - x - minD], xl) for xl in range(x + minD, min(W - 1, x + maxD) + 1)]
            if cand:
                right[y, x] = min(cand, key=lambda t: (t[0], t[1]))[1]
    return left, right


