"""Host model of the self-decoding grad_value kernel (vnext_amd/csrc/msda_d32_gvdirect.hip): the unit split of every
level (gvd_level_split, read through the debug ABI), the launcher's grid bound, and the kernel's index arithmetic --
ranks inside a row, segment offsets allocated per wave in any order, 2-lane groups walking the rows 256 slots at a time once per
channel half (the first half's sums held until the second's are ready), a row spread over 1 << gshift adjacent groups, later
passes adding onto the rows of the first -- replayed in numpy and held to the C oracle's
grad_value.  No GPU: what is checked here is the scheme (a tap applied twice or never shows up as a wrong sum), the GPU
tests (tests/test_msda_gvdirect.py) check the kernel.  Reference semantics: ms_deform_im2col_cuda.cuh:87-159,253-298."""
import ctypes
import random

import numpy as np
import pytest

from oracle import msda_oracle as O
from vnext_amd import _lib

QC, ROWS = 304, 768      # VNX_GVD_QC, VNX_GVD_ROWS (vnx_common.h)


def level_table(shapes, Lq, P, batch_heads=2):
    arr = np.asarray(shapes, dtype=np.int64)
    L = len(shapes)
    used, bound = ctypes.c_int(), ctypes.c_int()
    units, rpu, gs = (np.zeros(L, dtype=np.int32) for _ in range(3))
    rc = _lib.lib().vnx_debug_gvdirect_units(arr.ctypes.data, L, Lq, P, batch_heads, ctypes.byref(used), ctypes.byref(bound),
                                             units.ctypes.data, rpu.ctypes.data, gs.ctypes.data)
    assert rc == 0
    return used.value, bound.value, units, rpu, gs


def model_grad_value(value_shape, shapes, lsi, loc, attn, grad_out, rng):
    """grad_value [B, S, M, 32] by the kernel's scheme (float64 accumulation: only the bookkeeping is under test)."""
    B, S, M, D = value_shape
    _, Lq, _, L, P, _ = loc.shape
    used, bound, units, rpu, gs = level_table(shapes, Lq, P, B * M)
    assert used <= bound
    gv = np.full((B, S, M, D), np.nan)
    go = grad_out.reshape(B, Lq, M, D).astype(np.float64)
    qc = min(QC, (QC * 4) // P)
    for b in range(B):
        for m in range(M):
            for l in range(L):
                H, W = shapes[l]
                n = H * W
                for u in range(int(units[l])):
                    r0, r1 = u * int(rpu[l]), min((u + 1) * int(rpu[l]), n)
                    rows, gshift = r1 - r0, int(gs[l])
                    assert 0 < rows <= ROWS
                    step, gmask = 1 << gshift, (1 << gshift) - 1
                    stored = np.zeros((rows, D))
                    for pass_, q_lo in enumerate(range(0, Lq, qc)):
                        qs = np.arange(q_lo, min(q_lo + qc, Lq))
                        x = loc[b, qs, m, l, :, 0].astype(np.float32); y = loc[b, qs, m, l, :, 1].astype(np.float32)
                        a = attn[b, qs, m, l, :].astype(np.float32)
                        h = y * np.float32(H) - np.float32(0.5); w = x * np.float32(W) - np.float32(0.5)
                        inside = (h > -1) & (w > -1) & (h < H) & (w < W)
                        h0 = np.floor(h).astype(np.int64); w0 = np.floor(w).astype(np.int64)
                        lh = (h - np.floor(h)).astype(np.float32); lw = (w - np.floor(w)).astype(np.float32)
                        hh, hw = 1 - lh, 1 - lw
                        taps = []          # (row, slot, weight)
                        for t, (dy, dx, wt) in enumerate(((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw))):
                            yy, xx = h0 + dy, w0 + dx
                            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
                            p = yy * W + xx
                            ok &= (p >= r0) & (p < r1)
                            qi, ki = np.nonzero(ok)
                            for i, k in zip(qi, ki):
                                taps.append((int(p[i, k] - r0), int(i), float(a[i, k] * wt[i, k])))
                        assert len(taps) <= 4 * QC * 4       # the sorted list holds every tap of a pass
                        rng.shuffle(taps)      # arrival order of the rank atomics
                        cnt = np.zeros(rows, dtype=np.int64)
                        ranked = []
                        for row, slot, wt in taps:
                            ranked.append((row, int(cnt[row]), slot, wt))
                            cnt[row] += 1
                        # offsets: a wave scans 64 consecutive rows, waves allocate their totals in any order
                        offs = np.zeros(rows, dtype=np.int64)
                        waves = list(range((rows + 63) // 64))
                        rng.shuffle(waves)
                        base = 0
                        for wv in waves:
                            lo, hi = wv * 64, min(wv * 64 + 64, rows)
                            offs[lo:hi] = base + np.concatenate(([0], np.cumsum(cnt[lo:hi])[:-1]))
                            base += int(cnt[lo:hi].sum())
                        lst = [None] * max(base, 1)
                        for row, rk, slot, wt in ranked:
                            assert lst[offs[row] + rk] is None
                            lst[offs[row] + rk] = (slot, wt)
                        # the walk, once per channel half: slot = row * groups-per-row + part, 256 slots per round (2-lane
                        # groups); parts meet, part 0 stores -- the first half's sums wait for the second's
                        n_slots = rows << gshift
                        assert n_slots <= 3 * 256      # what a group's held sums cover (kIters)
                        held = {}
                        for cp in (0, 1):
                            ch = slice(16 * cp, 16 * cp + 16)
                            for sb in range(0, n_slots, 256):
                                partial = {}
                                for grp in range(256):
                                    slot = sb + grp
                                    if slot >= n_slots:
                                        continue
                                    row, part = slot >> gshift, slot & gmask
                                    acc = np.zeros(16)
                                    i = part
                                    while i < cnt[row]:
                                        sl, wt = lst[offs[row] + i]
                                        acc += wt * go[b, qs[sl], m, ch]
                                        i += step
                                    partial.setdefault(row, []).append(acc)
                                for row, parts in partial.items():
                                    assert len(parts) == step      # the groups of a row are adjacent: one round, one wave (32 groups)
                                    if cp == 0:
                                        held[row] = sum(parts)
                                    else:
                                        both = np.concatenate((held.pop(row), sum(parts)))
                                        stored[row] = both + (stored[row] if pass_ > 0 else 0)
                        assert not held
                    gv[b, lsi[l] + r0: lsi[l] + r1, m] = stored
    assert not np.isnan(gv).any(), "a row without an owner"
    return gv


def make_case(shapes, B, Lq, M, P, seed, concentrate=False):
    rs = np.random.RandomState(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    lsi = O.level_start_index(np.asarray(shapes, dtype=np.int64))
    value = rs.randn(B, S, M, 32).astype(np.float32)
    loc = rs.rand(B, Lq, M, L, P, 2).astype(np.float32) * 1.3 - 0.15       # some samples outside the map
    if concentrate:
        loc[:] = 0.5 + 0.01 * rs.rand(*loc.shape)
    attn = rs.rand(B, Lq, M, L, P).astype(np.float32)
    attn /= attn.reshape(B, Lq, M, -1).sum(-1).reshape(B, Lq, M, 1, 1)
    go = rs.randn(B, Lq, M * 32).astype(np.float32)
    return value, np.asarray(shapes, dtype=np.int64), lsi, loc, attn, go


@pytest.mark.parametrize("shapes,Lq,P,concentrate", [
    ([(6, 10), (3, 5), (2, 3), (1, 1)], 37, 4, False),            # small levels: rows spread over several groups
    ([(12, 40), (6, 20)], 50, 4, True),                           # every sample on one spot
    ([(9, 45), (5, 23), (3, 12)], 21, 3, False),                  # another point count
    ([(33, 40), (4, 4)], 30, 4, False),                           # 1 320 pixels: three units of rows
    ([(4, 4)], 5, 4, False),
    ([(3, 5), (2, 2)], 330, 4, False),                            # two passes: the second adds onto the rows of the first
])
def test_scheme_reproduces_the_oracle(shapes, Lq, P, concentrate):
    B, M = 1, 2
    value, sh, lsi, loc, attn, go = make_case(shapes, B, Lq, M, P, 11, concentrate)
    want = O.msda_backward(value, sh, lsi, loc, attn, go)[0]
    got = model_grad_value(value.shape, shapes, lsi, loc, attn, go, random.Random(5))
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() / scale < 2e-5


def test_baseline_level_tables():
    # T=5 decoder call at 360p (units of up to 768 rows; 40 (batch, head) pairs: the stand-alone kernel's grid fits one round,
    # the two small levels are cut in two): 5 + 2 + 2 + 2 units; a row of the 60-pixel level (80 taps) on eight groups, of the
    # 240-pixel level (20 taps) on two
    used, bound, units, rpu, gs = level_table([(48, 80), (24, 40), (12, 20), (6, 10)], 300, 4, 40)
    assert list(units) == [5, 2, 2, 2] and list(rpu) == [768, 480, 120, 30] and list(gs) == [0, 0, 1, 3]
    assert used == 11 <= bound
    # B = 10: two rounds of workgroups either way, one unit per small level
    used, bound, units, rpu, gs = level_table([(48, 80), (24, 40), (12, 20), (6, 10)], 300, 4, 80)
    assert list(units) == [5, 2, 1, 1] and list(rpu) == [768, 480, 240, 60] and used == 9 <= bound
    used, bound, units, rpu, gs = level_table([(92, 160), (46, 80), (23, 40), (12, 20)], 300, 4, 40)
    assert used <= bound and list(units)[0] == 23 and all(r <= 640 for r in rpu)      # the 720p pyramid: units of up to 640 rows


@pytest.mark.parametrize("seed", range(4))
def test_random_pyramids_never_pass_the_bound(seed):
    rnd = random.Random(seed)
    for _ in range(4000):
        L = rnd.choice([1, 2, 4, 5])
        h0, w0 = rnd.randint(1, 400), rnd.randint(1, 1500)
        shapes = [(max(1, -(-h0 // (1 << l)) + rnd.randint(0, 1)), max(1, -(-w0 // (1 << l)) + rnd.randint(0, 1))) for l in range(L)]
        if rnd.random() < 0.2:
            rnd.shuffle(shapes)
        if rnd.random() < 0.1:
            shapes[rnd.randrange(L)] = (1, rnd.choice([1, 2, 7, 70000]))
        Lq = rnd.choice([1, 7, 100, 300, 900, 1023, 5000])
        P = rnd.choice([1, 2, 4, 8])
        bm = rnd.choice([1, 2, 8, 40, 80, 1000])
        used, bound, units, rpu, gs = level_table(shapes, Lq, P, bm)
        assert used <= bound, (shapes, Lq, P, bm, used, bound)
        for (h, w), u, r, g in zip(shapes, units, rpu, gs):
            n = h * w
            assert 1 <= r <= ROWS and (u - 1) * r < n <= u * r, (shapes, Lq, P)     # every row has one owner, no unit is empty
            assert 0 <= g <= 3
