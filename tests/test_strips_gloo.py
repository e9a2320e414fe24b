"""Multi-GPU path on CPU: row-strip partition, halo exchange and gather over torch.distributed
with the gloo backend, world_size 2 and 3 (the GPU box runs the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplestereo_amd import strips


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeMatcher:
    """stands in for StereoASW on CPU: output row y = a function of input rows y-pad..y+pad"""

    def __init__(self, winSize):
        self.winSize = winSize

    @staticmethod
    def whole(L, R, pad):
        H = L.shape[0]
        acc = torch.zeros(L.shape[:2], dtype=torch.int64)
        for dy in range(-pad, pad + 1):
            lo, hi = max(0, -dy), min(H, H - dy)
            if lo >= hi:
                continue
            w = (dy + pad + 1)
            acc[lo:hi] += w * (L[lo + dy:hi + dy].sum(-1).long() - 2 * R[lo + dy:hi + dy, :, 0].long())
        return (acc % 30011).to(torch.int16)

    def _compute_device(self, subL, subR, out_row0=0, out_rows=None):
        if subL.shape[0] == 0 or not out_rows:              # like check_common of the real operators
            raise ValueError("Wrong image dimensions!")
        pad = self.winSize // 2
        full = self.whole(subL, subR, pad)
        return full[out_row0:out_row0 + out_rows].contiguous()


def _worker(rank, world, port, H, W, pad, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(seed)
        L = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        R = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        r0, r1 = strips.strip_bounds(H, world, rank)
        subL, subR, o0, orows = strips.exchange_halos(L[r0:r1].contiguous(), R[r0:r1].contiguous(), H, pad, rank, world)
        h0, h1 = strips.halo_bounds(H, r0, r1, pad)
        ok = torch.equal(subL, L[h0:h1]) and torch.equal(subR, R[h0:h1]) and o0 == r0 - h0 and orows == r1 - r0
        m = _FakeMatcher(2 * pad + 1)
        full = strips.match_strip(m, L[r0:r1].contiguous(), R[r0:r1].contiguous(), H, rank, world, gather=True)
        ok = ok and torch.equal(full, _FakeMatcher.whole(L, R, pad))
        ctx = strips.StripContext(m, H, W, rank, world, torch.device("cpu"))
        for _ in range(2):                                   # reusable across frames
            again = ctx.step(L[r0:r1].contiguous(), R[r0:r1].contiguous())
            ok = ok and torch.equal(again, _FakeMatcher.whole(L, R, pad))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,pad", [(2, 37, 5), (3, 10, 17), (3, 2, 3)])
def test_halo_exchange_and_gather_gloo(world, H, pad):
    """includes strips thinner than the halo (rows then come from several ranks) and, with more ranks than image
    rows, a rank whose strip is empty: it skips the kernels but still joins the gather"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, 23, pad, 1234, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_strip_bounds_partition_rows():
    for H in (1, 7, 288, 1080, 2160):
        for world in (1, 2, 3, 8):
            b = [strips.strip_bounds(H, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == H
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1


def test_transfer_plan_covers_exactly_the_halos():
    H, world, pad = 50, 4, 6
    plan = strips.transfer_plan(H, world, pad)
    for dst in range(world):
        r0, r1 = strips.strip_bounds(H, world, dst)
        h0, h1 = strips.halo_bounds(H, r0, r1, pad)
        need = set(range(h0, r0)) | set(range(r1, h1))
        got = []
        for s, d, lo, hi in plan:
            if d == dst:
                s0, s1 = strips.strip_bounds(H, world, s)
                assert s0 <= lo < hi <= s1          # the sender owns what it sends
                got += list(range(lo, hi))
        assert sorted(got) == sorted(need)
