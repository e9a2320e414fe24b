"""Deterministic synthetic rectified stereo pairs (own code, numpy only).

There is no network for datasets, so ``bench.py`` and the full-size parity tests
run on seeded synthetic pairs with a known integer ground-truth disparity:
a smooth multi-octave random texture (so that adaptive support weights have real
colour structure to latch onto) carrying fronto-parallel rectangles at different
depths.  The right view is a forward warp of the left one with a z-buffer
(nearest surface wins); disoccluded holes are filled with fresh noise.

The generator only uses ``numpy.random.default_rng(seed)`` with ``integers`` and
``random`` so that the CPU oracle and the GPU path see identical bytes on any box
running the same numpy.
"""
import numpy as np

__all__ = ["make_pair"]


def _bilinear_upsample(grid, cell, H, W):
    """grid [gh,gw,C] sampled every `cell` px -> [H,W,C] float64."""
    ys = np.arange(H) / cell
    xs = np.arange(W) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    return (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy


def make_pair(height, width, maxDisparity, seed=0, rectangles=12):
    """Return ``(left_bgr_u8, right_bgr_u8, gt_disparity_int16)``.

    Ground truth is the left-referenced disparity: ``right[y, x - d] == left[y, x]``
    wherever the surface is visible in both views.
    """
    H, W, maxD = int(height), int(width), int(maxDisparity)
    rng = np.random.default_rng(seed)

    tex = np.zeros((H, W, 3), np.float64)
    for o in range(4):
        cell = max(2, 64 >> o)
        gh, gw = H // cell + 3, W // cell + 3
        grid = rng.random((gh, gw, 3)) * 255.0
        tex += (0.5 ** o) * _bilinear_upsample(grid, cell, H, W)
    lo, hi = tex.min(), tex.max()
    tex = 16.0 + (tex - lo) * (224.0 / max(hi - lo, 1e-9))
    tex += rng.integers(-8, 9, size=(H, W, 3))
    left = np.clip(np.rint(tex), 0, 255).astype(np.uint8)

    disp = np.full((H, W), int(0.1 * maxD), np.int16)
    dlo, dhi = int(0.15 * maxD), max(int(0.9 * maxD), int(0.15 * maxD) + 1)
    rects = []
    for _ in range(rectangles):
        h = int(rng.integers(max(2, H // 8), max(3, H // 3) + 1))
        w = int(rng.integers(max(2, H // 8), max(3, H // 3) + 1))
        y0 = int(rng.integers(0, max(1, H - h)))
        x0 = int(rng.integers(0, max(1, W - w)))
        d = int(rng.integers(dlo, dhi + 1))
        rects.append((d, y0, x0, h, w))
    for d, y0, x0, h, w in sorted(rects):            # far to near
        disp[y0:y0 + h, x0:x0 + w] = d

    right = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)   # hole filler
    xs = np.arange(W)
    for d in np.unique(disp):                         # ascending: nearer overwrites
        yy, xx = np.nonzero(disp == d)
        keep = xx - d >= 0
        right[yy[keep], xx[keep] - int(d)] = left[yy[keep], xx[keep]]
    del xs
    return np.ascontiguousarray(left), np.ascontiguousarray(right), disp
