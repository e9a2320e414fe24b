"""CPU tier: properties of the row-strip partition and of the halo transfer plan for arbitrary image heights,
world sizes and window sizes (hypothesis) -- the multi-GPU path is correct by construction only if these hold."""
import numpy as np
from hypothesis import given, settings, strategies as st

from simplestereo_amd import strips


@settings(max_examples=300, deadline=None)
@given(H=st.integers(1, 400), world=st.integers(1, 16), pad=st.integers(0, 127))
def test_partition_and_transfer_plan(H, world, pad):
    own = [strips.strip_bounds(H, world, r) for r in range(world)]
    # contiguous partition of [0, H), sizes differ by at most one row
    assert own[0][0] == 0 and own[-1][1] == H
    assert all(own[r][1] == own[r + 1][0] for r in range(world - 1))
    sizes = [b - a for a, b in own]
    assert max(sizes) - min(sizes) <= 1
    plan = strips.transfer_plan(H, world, pad)
    got = {r: np.zeros(H, np.int32) for r in range(world)}
    for src, dst, lo, hi in plan:
        assert src != dst and 0 <= lo < hi <= H
        assert own[src][0] <= lo and hi <= own[src][1]            # the sender owns what it sends
        got[dst][lo:hi] += 1
    for r in range(world):
        r0, r1 = own[r]
        h0, h1 = strips.halo_bounds(H, r0, r1, pad)
        need = np.zeros(H, np.int32)
        if r1 > r0:
            need[h0:r0] = 1
            need[r1:h1] = 1
            assert h0 == max(0, r0 - pad) and h1 == min(H, r1 + pad)
        assert np.array_equal(got[r], need)                        # every halo row exactly once, nothing else
    # deterministic and identical on every rank: the plan is a pure function of (H, world, pad)
    assert plan == strips.transfer_plan(H, world, pad)
