"""CPU: the scheduler behind ngp_grid_encode_forward_sched (host code of libngp_hip.so, no device work).  For any per-level cost vector the
per-XCD work lists must PARTITION the launch: every (level, tile) in exactly one segment, at most 8 segments per XCD, whole levels on XCD
(level mod 8) unless moved, the slot ends cumulative -- an error here would make the kernel skip or repeat tiles, or index past a level."""
import ctypes

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import _ngp_capi as capi


def work_lists(L, tiles, costs):
    lev = (ctypes.c_uint16 * 64)()
    t0 = (ctypes.c_uint32 * 64)()
    end = (ctypes.c_uint32 * 64)()
    arr = None if costs is None else (ctypes.c_float * L)(*costs)
    n = capi.lib.ngp_grid_forward_work_lists(L, tiles, None if arr is None else ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lev, ctypes.c_void_p),
                                             ctypes.cast(t0, ctypes.c_void_p), ctypes.cast(end, ctypes.c_void_p))
    return n, np.array(lev).reshape(8, 8), np.array(t0).reshape(8, 8), np.array(end).reshape(8, 8)


def check_partition(L, tiles, costs):
    max_slots, lev, t0, end = work_lists(L, tiles, costs)
    seen = np.zeros((L, tiles), np.int32)
    longest = 0
    for x in range(8):
        begin = 0
        for sg in range(8):
            e = int(end[x, sg])
            assert e >= begin, 'cumulative ends'
            if lev[x, sg] == 0xffff:
                assert e == begin, 'unused segment consumes no slot'
                continue
            n = e - begin
            l, a = int(lev[x, sg]), int(t0[x, sg])
            assert 0 <= l < L and n >= 1 and a + n <= tiles, (l, a, n)
            seen[l, a:a + n] += 1
            begin = e
        longest = max(longest, begin)
    assert (seen == 1).all(), 'every (level, tile) exactly once'
    assert max_slots == longest
    return lev, end


def test_whole_levels_without_costs():
    for L, tiles in ((16, 272), (4, 7), (32, 1), (9, 100)):
        lev, end = check_partition(L, tiles, None)
        for x in range(8):
            own = [l for l in range(L) if l % 8 == x]
            assert [int(v) for v in lev[x] if v != 0xffff] == own      # level mod 8 placement, in order


def test_ray_model_moves_work_to_the_lightly_loaded_xcds():
    import torch  # noqa: F401
    costs = ctypes.cast(capi.ray_level_costs(16, float(np.log2(1.3819)), 16, 3.0 ** 0.5 / 1024), ctypes.POINTER(ctypes.c_float * 16)).contents
    lev, end = check_partition(16, 272, list(costs))
    loads = [sum(costs[int(lev[x, sg])] * (int(end[x, sg]) - (int(end[x, sg - 1]) if sg else 0)) for sg in range(8) if lev[x, sg] != 0xffff) for x in range(8)]
    whole = [(costs[x] + costs[x + 8]) * 272 for x in range(8)]
    assert max(loads) < 0.93 * max(whole) and max(loads) / min(loads) < 1.06     # balanced to a few per cent
    assert sum(1 for x in range(8) for sg in range(8) if lev[x, sg] != 0xffff) > 16  # some levels are split


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 32), st.integers(1, 3000), st.data())
def test_any_cost_vector_yields_a_partition(L, tiles, data):
    costs = data.draw(st.lists(st.floats(2.0 ** -13, 2.0 ** 13, allow_nan=False, allow_infinity=False, width=32), min_size=L, max_size=L))
    check_partition(L, tiles, costs)


def test_bad_arguments():
    assert work_lists(0, 10, None)[0] == 0 and work_lists(33, 10, None)[0] == 0
