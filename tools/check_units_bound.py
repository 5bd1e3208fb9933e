"""Exhaustive check behind the grid bound of the tile-fed grad_value kernel (msda_d32_gvtiles.hip, gvtiles_units_bound):
a level of H x W pixels is cut by gv_level_grid (vnx_common.h; restated here with numpy) into at most
3 * H * W / 256 + units_min units.  Every H <= 1 200 (and a few larger) x W <= 20 000, units_min in {1, 2, 16}."""
import numpy as np


def units(H, W, um, rows_max=256, minw=32, bw0=32):
    W = W.astype(np.int64)
    narrow = W < minw
    target = np.where(H >= rows_max // bw0, bw0, rows_max // max(H, 1))
    nbx = (W + target - 1) // target
    bw = (W + nbx - 1) // nbx
    nbx = (W + bw - 1) // bw
    nbx = np.where(narrow, 1, nbx)
    bw = np.where(narrow, W, bw)
    bh = np.clip(rows_max // bw, 1, H)
    nby = (H + bh - 1) // bh
    small = (nbx * nby < um) & (nby < H)
    want = np.minimum((um + nbx - 1) // nbx, H)
    bh2 = (H + want - 1) // want
    nby = np.where(small, (H + bh2 - 1) // bh2, nby)
    return nbx * nby


if __name__ == "__main__":
    W = np.arange(1, 20001)
    worst = -1e9
    for um in (1, 2, 16):
        for H in list(range(1, 1201)) + [2000, 5000, 20000]:
            e = (units(H, W, um) - 3 * H * W / 256 - um).max()
            worst = max(worst, e)
    print("max over all shapes of units - (3 n / 256 + units_min) =", worst)
    assert worst <= 0
