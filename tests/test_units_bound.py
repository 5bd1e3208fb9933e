"""The launcher of the tile-fed grad_value kernel sizes its grid from the pixel count S alone; the kernel derives its
units from the level shapes on the device (gv_level_grid, vnx_common.h).  A unit past the grid would silently lose its
rows, so the kernel's count is held to the launcher's bound here, on the host, through the debug ABI -- no GPU needed."""
import ctypes
import random

import numpy as np
import pytest

from vnext_amd import _lib


def used_and_bound(shapes, Lq, B, M, units_min=2):
    arr = np.asarray(shapes, dtype=np.int64)
    used, bound = ctypes.c_int(), ctypes.c_int()
    rows, rows_bound = ctypes.c_longlong(), ctypes.c_longlong()
    rc = _lib.lib().vnx_debug_gvtiles_units(arr.ctypes.data, len(shapes), Lq, B, M, units_min, ctypes.byref(used), ctypes.byref(bound),
                                            ctypes.byref(rows), ctypes.byref(rows_bound))
    assert rc == 0
    # round 4: the query pieces of the split levels store fp32 partial rows into a slab per (batch, head) whose size the host
    # derives from S and L alone (gv_partial_rows_bound): a piece past its slab would write into the next (batch, head)'s
    assert 0 <= rows.value <= rows_bound.value, (shapes, Lq, B, units_min, rows.value, rows_bound.value)
    return used.value, bound.value


def test_baseline_pyramids():
    for shapes in ([(48, 80), (24, 40), (12, 20), (6, 10)], [(92, 160), (46, 80), (23, 40), (12, 20)]):
        S = sum(h * w for h, w in shapes)
        for B in (1, 2, 5, 10):
            used, bound = used_and_bound(shapes, S, B, 8)
            assert 0 < used <= bound, (shapes, B, used, bound)


@pytest.mark.parametrize("seed", range(8))
def test_random_pyramids_never_pass_the_bound(seed):
    rnd = random.Random(seed)
    for _ in range(4000):
        L = 4
        h0, w0 = rnd.randint(1, 400), rnd.randint(1, 1500)
        shapes = [(max(1, -(-h0 // (1 << l)) + rnd.randint(0, 1)), max(1, -(-w0 // (1 << l)) + rnd.randint(0, 1))) for l in range(L)]
        if rnd.random() < 0.2:
            rnd.shuffle(shapes)
        S = sum(h * w for h, w in shapes)
        Lq = rnd.choice([S, 1024, 3000, 100000])
        B, M = rnd.choice([1, 2, 5, 16]), 8
        um = rnd.choice([1, 2, 3, 16])
        used, bound = used_and_bound(shapes, Lq, B, M, um)
        assert used <= bound, (shapes, Lq, B, um, used, bound)


def test_degenerate_levels_keep_their_partial_rows_inside_the_slab():
    """flat, one-row, one-pixel and equal-sized levels, with every query count around the split threshold"""
    cases = [[(1, 1)] * 4, [(1, 1024)] * 4, [(1024, 1)] * 4, [(16, 64)] * 4, [(8, 128), (4, 256), (2, 512), (1, 1024)],
             [(3, 300), (300, 3), (1, 70000), (7, 7)], [(32, 32), (31, 33), (1, 1), (2, 1000)]]
    for shapes in cases:
        S = sum(h * w for h, w in shapes)
        for Lq in (1023, 1024, S, 5 * S + 7, 1 << 20):
            for B, M in ((1, 1), (1, 8), (4, 8), (64, 8)):
                for um in (1, 2, 16):
                    used_and_bound(shapes, Lq, B, M, um)
