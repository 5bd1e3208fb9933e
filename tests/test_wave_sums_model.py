"""Host model of the mask head backward's transposed wave sums (vnext_amd/csrc/mask_head.hip: fold32 .. wave_sums,
tree_index).  Each fold is restated on 64-element arrays exactly as tools/fold_probe.hip observed the instruction on the
hardware (v_permlane32_swap / v_permlane16_swap, DPP row_shr / row_shl with quad-granular bank masks, quad_perm); the tree
built from them must leave the complete sum of register tree_index(l) in lane l for every register count the kernel uses.
The kernel itself is checked against the oracle on the GPU (tests/test_mask_head.py); this pins the lane bookkeeping."""
import numpy as np
import pytest

LANES = np.arange(64)


def fold32(a, b):      # swap a[32:] with b[:32], add: lanes 0..31 a's pair sums, 32..63 b's
    a2 = np.concatenate([a[:32], b[:32]]); b2 = np.concatenate([a[32:], b[32:]])
    return a2 + b2


def fold16(a, b):      # swap a's odd rows with b's even rows (rows of 16 lanes), add
    ar, br = a.reshape(4, 16), b.reshape(4, 16)
    a2 = np.stack([ar[0], br[0], ar[2], br[2]]); b2 = np.stack([ar[1], br[1], ar[3], br[3]])
    return (a2 + b2).reshape(64)


def fold8(a, b=None):  # lanes 8..15 of a row: a[l - 8] + a[l]; lanes 0..7: b[l + 8] + b[l]
    r = a.copy()
    hi = (LANES % 16) >= 8
    r[hi] = a[LANES[hi] - 8] + a[hi]
    if b is not None:
        lo = ~hi
        r[lo] = b[LANES[lo] + 8] + b[lo]
    return r


def fold4(a, b=None):  # quads 1, 3 of a row: a[l - 4] + a[l]; quads 0, 2: b[l + 4] + b[l]
    r = a.copy()
    odd = ((LANES % 16) // 4) % 2 == 1
    r[odd] = a[LANES[odd] - 4] + a[odd]
    if b is not None:
        even = ~odd
        r[even] = b[LANES[even] + 4] + b[even]
    return r


def fold_quad(a, b, dist):   # keep + swap(send) on the lane's bit `dist`
    bit = (LANES & dist) != 0
    if b is None:
        return a + a[LANES ^ dist]
    keep = np.where(bit, b, a); send = np.where(bit, a, b)
    return keep + send[LANES ^ dist]


def wave_sums(regs):
    zero = np.zeros(64)
    pair = lambda r, f: [f(r[i], r[i + 1] if i + 1 < len(r) else None) for i in range(0, len(r), 2)]   # noqa: E731
    regs = pair(regs, lambda a, b: fold32(a, zero if b is None else b))
    regs = pair(regs, lambda a, b: fold16(a, zero if b is None else b))
    regs = pair(regs, fold8)
    regs = pair(regs, fold4)
    regs = pair(regs, lambda a, b: fold_quad(a, b, 2))
    assert len(regs) <= 2
    return fold_quad(regs[0], regs[1] if len(regs) > 1 else None, 1)


def tree_index(l):
    return ((l >> 5) & 1) | (((l >> 4) & 1) << 1) | ((((l >> 3) & 1) ^ 1) << 2) | ((((l >> 2) & 1) ^ 1) << 3) | \
        (((l >> 1) & 1) << 4) | ((l & 1) << 5)


@pytest.mark.parametrize("n", [64, 16, 27, 5, 2, 1, 33, 63])
def test_lane_l_holds_the_sum_of_register_tree_index_l(n):
    rng = np.random.default_rng(n)
    regs = [rng.integers(-1000, 1000, 64).astype(np.float64) for _ in range(n)]
    out = wave_sums([r.copy() for r in regs])
    seen = set()
    for l in range(64):
        i = tree_index(l)
        if i < n:
            assert out[l] == regs[i].sum(), (n, l, i)
            seen.add(i)
    assert seen == set(range(n))            # every register's total is in some lane


def test_tree_index_is_a_permutation_of_the_lanes():
    assert sorted(tree_index(l) for l in range(64)) == list(range(64))
