"""CPU tier: the work accounting bench.py reports -- count_taps must equal a literal count of the reference's
loops (_passive.cpp:56-68: candidates d = x-min .. max(0, x-max) descending in the right column, taps skipped
unless the row, the left column and the right column are inside the image)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _literal(H, W, win, maxD, minD, row0, rows):
    p, n = win // 2, 0
    for y in range(row0, row0 + rows):
        for x in range(W):
            for d in range(minD, maxD + 1):
                if x - d < 0:
                    continue
                for i in range(win):
                    if not 0 <= y - p + i < H:
                        continue
                    for j in range(win):
                        if 0 <= x - p + j < W and 0 <= x - d - p + j < W:
                            n += 1
    return n


@pytest.mark.parametrize("H,W,win,maxD,minD,row0,rows", [(7, 9, 5, 4, 0, 0, 7), (6, 11, 7, 6, 2, 1, 3), (3, 8, 9, 10, 0, 0, 3),
                                                       (10, 6, 3, 2, 1, 4, 5), (5, 5, 1, 3, 0, 0, 5)])
def test_count_taps_equals_the_reference_loop_count(H, W, win, maxD, minD, row0, rows):
    assert bench.count_taps(H, W, win, maxD, minD, row0, rows) == _literal(H, W, win, maxD, minD, row0, rows)


def test_full_frame_counts_used_in_the_bench_line():
    assert bench.count_taps(1080, 1920, 35, 192, 0) == 459753752628          # config 3 ("taps_per_launch")
    assert bench.ALGO_BYTES_PER_PIXEL == 34 and bench.CONFIGS["c3_1080p_d192_w35"] == (1080, 1920, 192, 0, 35)
