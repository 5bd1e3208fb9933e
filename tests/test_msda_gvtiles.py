"""grad_value from per-tile words (vnext_amd/csrc/msda_d32_gvtiles.hip): the path calls with >= 1 024 queries take
(the encoders': queries = pixels of the pyramid).  Everything against the fp64 oracle, through the C ABI:
every tile size the grad_loc launcher can produce, units_min variants on small levels (the latent ADVICE r2 case:
zeroing and atomics must agree on which levels are query-split), 16-bit values, ragged query counts, forced
records / tiles paths agreeing with each other, samples far from their reference point.
Reference semantics: ms_deform_im2col_cuda.cuh:87-159,253-298."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from oracle import msda_oracle as O  # noqa: E402
from vnext_amd import _lib  # noqa: E402

DEV = "cuda:0"
PYR = [(24, 40), (12, 20), (6, 10), (3, 5)]              # S = 1275
TINY = [(5, 7), (3, 4), (2, 2), (1, 1)]                  # S = 52: every level a handful of pixels


def pixel_queries(shapes, B, Lq, seed, M=8, P=4, noise=1.0, far=0.02):
    """Lq queries on (repeated) pixel centres of the pyramid, samples = centre + (head direction x (k+1) + N(0, noise))
    pixels; `far` of the queries sample anywhere."""
    g = torch.Generator().manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int(sh.prod(1).sum())
    lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    ref = []
    for h, w in shapes:
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5,
                                indexing="ij")
        ref.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
    ref = torch.cat(ref, 0)
    ref = ref[torch.arange(Lq) % S]
    th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    k = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, 1, 1, P, 1)
    off = d.view(1, 1, M, 1, 1, 2) * k + noise * torch.randn(B, Lq, M, L, P, 2, generator=g)
    is_far = torch.rand(B, Lq, 1, 1, 1, 1, generator=g) < far
    off = torch.where(is_far, 15.0 * torch.randn(B, Lq, M, L, P, 2, generator=g), off)
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float().view(1, 1, 1, L, 1, 2)
    loc = (ref.view(1, Lq, 1, 1, 1, 2) + off / wh).contiguous()
    value = torch.randn(B, S, M, 32, generator=g)
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P).contiguous()
    go = torch.randn(B, Lq, M * 32, generator=g)
    return sh, lsi, value, loc, attn, go


def scale(x):
    return max(1e-30, float(np.abs(x).max()))


def run(case, variant, vdt=torch.float32, ldt=torch.float32, packed=True):
    sh, lsi, value, loc, attn, go = case
    _lib.set_kernel_variant(variant)
    try:
        out = MSDA.ms_deform_attn_backward(value.to(DEV, vdt), sh.to(DEV), lsi.to(DEV), loc.to(DEV, ldt), attn.to(DEV, ldt),
                                           go.to(DEV, vdt), 64, levels_packed=packed)
        torch.cuda.synchronize()
    finally:
        _lib.set_kernel_variant(0)
    return [t.double().cpu().numpy() for t in out]


def oracle(case, vdt=torch.float32, ldt=torch.float32):
    sh, lsi, value, loc, attn, go = case
    args = (value.to(vdt).double().numpy(), sh.numpy(), lsi.numpy(), loc.to(ldt).double().numpy(),
            attn.to(ldt).double().numpy())
    return O.msda_backward(*args, go.to(vdt).double().numpy(), nthreads=8)


def boundary_mask(loc, sh, eps=1e-4):
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double().view(1, 1, 1, -1, 1, 2)
    px = loc.double() * wh - 0.5
    return ((px - px.round()).abs() > eps).all(-1, keepdim=True).numpy()


def check(got, want, case, tol=2e-5):
    gv, gl, ga = got
    rv, rl, ra = want
    ok = boundary_mask(case[3], case[0])
    np.testing.assert_allclose(gv, rv, rtol=0, atol=tol * scale(rv))
    np.testing.assert_allclose(gl * ok, rl * ok, rtol=0, atol=tol * scale(rl))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=tol * scale(ra))


# variant -> queries per tile of the grad_loc launcher: 0 auto, 2..5 = 32 / 16 / 8 / 4, 12..15 = 8 / 4 / 2 / 1
@pytest.mark.parametrize("variant", [0, 2, 3, 4, 5, 12, 13, 14, 15, 431])
def test_tile_sizes_against_the_oracle(variant):
    case = pixel_queries(PYR, 2, 1275, seed=3)
    check(run(case, variant), oracle(case), case)


@pytest.mark.parametrize("Lq", [1024, 1030, 1275 + 77, 2600])
def test_ragged_query_counts(Lq):
    """query counts that are no multiple of the tile, of the chunk (128) or of the pyramid"""
    case = pixel_queries(PYR, 3, Lq, seed=Lq)
    check(run(case, 0), oracle(case), case)


@pytest.mark.parametrize("units_min", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("shapes", [TINY, PYR])
def test_units_min_variants_on_small_levels(units_min, shapes):
    """variants 200 + x set the minimum number of units per level: the grad_loc kernel (which zeroes the rows of
    query-split levels), the grad_value kernel (which adds onto them) and the tile words must use ONE split
    (gv_level_split, vnx_common.h).  Lq >= 1281 puts 16-pixel levels on both sides of the threshold."""
    case = pixel_queries(shapes, 2, 1300, seed=units_min)
    check(run(case, 200 + units_min), oracle(case), case)


def test_records_path_and_tiles_path_agree():
    case = pixel_queries(PYR, 2, 1275, seed=11)
    a, b = run(case, 430), run(case, 431)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=0, atol=3e-6 * scale(y))
    small = pixel_queries(PYR, 2, 300, seed=12)                 # tiles forced below the automatic threshold
    check(run(small, 431), oracle(small), small)


@pytest.mark.parametrize("vdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("loc16", [False, True])
def test_sixteen_bit_values(vdt, loc16):
    case = pixel_queries(PYR, 2, 1275, seed=21)
    ldt = vdt if loc16 else torch.float32
    gv, gl, ga = run(case, 0, vdt, ldt)
    rv, rl, ra = oracle(case, vdt, ldt)
    np.testing.assert_allclose(gv, rv, rtol=0, atol=1e-2 * scale(rv))
    if not loc16:
        ok = boundary_mask(case[3], case[0], eps=1e-3)
        np.testing.assert_allclose(gl * ok, rl * ok, rtol=0, atol=1e-2 * scale(rl))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=1e-2 * scale(ra))


def test_uniform_locations_and_all_outside():
    """no locality at all (every tile touches every unit), and a call none of whose samples is in the map"""
    g = torch.Generator().manual_seed(5)
    sh, lsi, value, loc, attn, go = pixel_queries(PYR, 2, 1500, seed=5)
    uni = (sh, lsi, value, torch.rand(loc.shape, generator=g), attn, go)
    check(run(uni, 0), oracle(uni), uni)
    out = (sh, lsi, value, torch.full_like(loc, 5.0), attn, go)
    gv, gl, ga = run(out, 0)
    assert not gv.any() and not gl.any() and not ga.any()


def test_not_promised_packed_and_run_to_run():
    case = pixel_queries(PYR, 2, 1275, seed=31)
    check(run(case, 0, packed=False), oracle(case), case)
    # grad_loc / grad_attn have one writer and a fixed summation order: bit-identical from run to run.  grad_value rows
    # have one owner, but the order of a row's taps inside a chunk comes from LDS rank atomics (arrival order), so
    # its last bits may differ from run to run
    big = pixel_queries([(48, 80), (24, 40), (12, 20), (6, 10)], 1, 5100, seed=32)
    a, b = run(big, 0), run(big, 0)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=2e-6 * scale(a[0]))


@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16])
def test_fused_backward_records_path_and_tiles_path_agree(vdt):
    """The fused prologue never materialises sampling_locations / attention_weights; in tile mode its grad_loc kernel
    leaves them (fp32) next to the tile words for the grad_value kernel.  Same gradients as with per-sample records."""
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors
    shapes, B, Lq, M, L, P = PYR, 2, 1275, 8, 4, 4
    g = torch.Generator().manual_seed(77)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, 32, generator=g).to(vdt)
    offsets = torch.randn(B, Lq, M, L, P, 2, generator=g) * 2
    logits = torch.randn(B, Lq, M, L * P, generator=g) * 2
    ref = torch.rand(B, Lq, L, 2, generator=g)
    gout = torch.randn(B, Lq, M * 32, generator=g).to(vdt)
    shapes_t, lsi = level_tensors(shapes, "cuda")
    res = {}
    for variant in (430, 0):
        _lib.set_kernel_variant(variant)
        try:
            leaves = [value.cuda().requires_grad_(True), offsets.cuda().requires_grad_(True), logits.cuda().requires_grad_(True),
                      ref.cuda().requires_grad_(True)]
            out = MSDeformAttnFusedFunction.apply(leaves[0], shapes_t, lsi, leaves[1], leaves[2], leaves[3])
            out.backward(gout.cuda())
            torch.cuda.synchronize()
            res[variant] = [t.grad.float().cpu().numpy() for t in leaves]
        finally:
            _lib.set_kernel_variant(0)
    tol = 3e-6 if vdt == torch.float32 else 1e-2
    for a, b in zip(res[430], res[0]):
        np.testing.assert_allclose(a, b, rtol=0, atol=tol * scale(b))
    assert np.array_equal(res[430][1], res[0][1]) and np.array_equal(res[430][2], res[0][2])   # same kernel, same order


# the unit grid of gv_level_grid (vnx_common.h): levels from 128 pixels of width up are cut into blocks of about
# 32 x 8 pixels, flat wide levels (H < 8) into wider blocks, narrow levels into bands of whole image rows
WIDE = [(20, 130), (10, 65), (5, 33), (3, 17)]           # S = 3466: level 0 in blocks of 26 x 9 (5 x 3 of them)
FLAT = [(5, 300), (3, 150), (2, 75), (1, 38)]            # S = 2138: levels 0 and 1 in flat blocks, H < 8
LINE = [(1, 700), (1, 350), (1, 175), (1, 88)]           # S = 1313: one-row levels


@pytest.mark.parametrize("name,shapes", [("wide", WIDE), ("flat", FLAT), ("line", LINE)])
@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16])
def test_block_units_of_wide_levels(name, shapes, vdt):
    S = sum(h * w for h, w in shapes)
    case = pixel_queries(shapes, 2, S, seed=len(name))
    if vdt == torch.float32:
        # fp32 pixel coordinates x * W - 0.5 carry ulp(W) of rounding: 6e-5 px at W = 700, which the bilinear weights and the
        # attention gradient inherit (both grad_value paths show the same 1.3e-5 / 3.8e-5 there; the reference computes in fp32 too)
        tol = 8e-5 if name == "line" else 2e-5
        check(run(case, 0), oracle(case), case, tol)
        units3 = run(case, 203)                            # at least three units per level
        check(units3, oracle(case), case, tol)
    else:
        gv, gl, ga = run(case, 0, vdt)
        rv, rl, ra = oracle(case, vdt)
        np.testing.assert_allclose(gv, rv, rtol=0, atol=1e-2 * scale(rv))
        np.testing.assert_allclose(ga, ra, rtol=0, atol=1e-2 * scale(ra))


def test_block_units_with_far_and_border_samples():
    """samples far from their query (any block may be hit), on the border and outside of a wide level"""
    sh, lsi, value, loc, attn, go = pixel_queries(WIDE, 1, 3466, seed=9, far=0.3)
    g = torch.Generator().manual_seed(10)
    loc = loc.clone()
    loc[:, ::7] = torch.rand(loc[:, ::7].shape, generator=g) * 1.2 - 0.1       # in and around the map, incl. its edges
    case = (sh, lsi, value, loc, attn, go)
    check(run(case, 0), oracle(case), case)


def test_level_wider_than_the_tile_boxes_can_count():
    """A level side beyond 65 534 pixels: the tile boxes saturate ("or beyond") and stay conservative."""
    shapes = [(1, 70000), (1, 1100), (1, 64), (1, 8)]
    case = pixel_queries(shapes, 1, 1100, seed=3, M=8)
    sh, lsi, value, loc, attn, go = case
    g = torch.Generator().manual_seed(4)
    loc = loc.clone()
    loc[:, ::3, :, 0] = torch.rand(loc[:, ::3, :, 0].shape, generator=g)       # level 0 sampled all along its 70 000 pixels
    case = (sh, lsi, value, loc, attn, go)
    # fp32 coordinates x * 70 000 - 0.5 carry 4e-3 px of rounding: compare off the pixel grid with that margin
    gv, gl, ga = run(case, 0)
    rv, rl, ra = oracle(case)
    np.testing.assert_allclose(gv, rv, rtol=0, atol=2e-2 * scale(rv))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=2e-2 * scale(ra))
    a, b = run(case, 430), run(case, 0)                                          # the record-fed path computes the same
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-5 * scale(b[0]))
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("seed", range(8))
def test_random_pyramids_on_both_grad_value_paths(seed):
    """odd level shapes (single rows / columns, wide-and-flat, levels out of size order) through the rectangle grid"""
    import random
    rnd = random.Random(1000 + seed)
    shapes = []
    for _ in range(4):
        kind = rnd.choice(["tiny", "row", "col", "wide", "block", "plain"])
        h, w = {"tiny": (rnd.randint(1, 3), rnd.randint(1, 3)), "row": (1, rnd.randint(2, 300)), "col": (rnd.randint(2, 120), 1),
                "wide": (rnd.randint(2, 7), rnd.randint(64, 260)), "block": (rnd.randint(8, 30), rnd.randint(64, 140)),
                "plain": (rnd.randint(4, 40), rnd.randint(4, 63))}[kind]
        shapes.append((h, w))
    S = sum(h * w for h, w in shapes)
    Lq = max(1024 + rnd.randint(0, 40), min(S, 2600))
    case = pixel_queries(shapes, rnd.choice([1, 2, 3]), Lq, seed=seed, far=0.1)
    want = oracle(case)
    tol = 4e-5 if max(w for _, w in shapes) > 200 else 2e-5         # fp32 pixel coordinates of wide levels (see `line` above)
    check(run(case, 0), want, case, tol)
    check(run(case, 430), want, case, tol)
    check(run(case, 200 + rnd.choice([1, 3, 5])), want, case, tol)


@pytest.mark.parametrize("seed", range(4))
def test_random_pyramids_forward_and_decoder_backward(seed):
    """the same odd pyramids with few queries: forward and the record-fed backward against the oracle"""
    import random
    rnd = random.Random(2000 + seed)
    shapes = [(rnd.randint(1, 30), rnd.randint(1, 200)) for _ in range(4)]
    case = pixel_queries(shapes, 2, rnd.choice([7, 64, 300, 777]), seed=seed, far=0.2)
    sh, lsi, value, loc, attn, go = case
    out = MSDA.ms_deform_attn_forward(value.to(DEV), sh.to(DEV), lsi.to(DEV), loc.to(DEV), attn.to(DEV), 64)
    torch.cuda.synchronize()
    want = O.msda_forward(value.double().numpy(), sh.numpy(), lsi.numpy(), loc.double().numpy(), attn.double().numpy(), nthreads=8)
    np.testing.assert_allclose(out.double().cpu().numpy().reshape(want.shape), want, rtol=0, atol=4e-5 * scale(want))
    check(run(case, 0), oracle(case), case, 4e-5)


@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16])
def test_tile_path_backward_in_a_graph(vdt):
    """no allocation / synchronisation inside the call (workspace from the caller): the backward of a many-query call
    (grad_loc kernel, tile-fed grad_value kernel, split-level convert for 16-bit values) replays from a hipGraph"""
    sh, lsi, value, loc, attn, go = pixel_queries(PYR, 2, 1275, seed=41)
    args = (value.to(DEV, vdt), sh.to(DEV), lsi.to(DEV), loc.to(DEV), attn.to(DEV), go.to(DEV, vdt), 64)
    expect = [t.clone() for t in MSDA.ms_deform_attn_backward(*args)]
    torch.cuda.synchronize()
    static = [torch.empty_like(t) for t in expect]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for dst, src in zip(static, MSDA.ms_deform_attn_backward(*args)):
            dst.copy_(src)
    for t in static:
        t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static[1], expect[1]) and torch.equal(static[2], expect[2])
    tol = 2e-6 if vdt == torch.float32 else 1e-2
    assert float((static[0].float() - expect[0].float()).abs().max()) <= tol * float(expect[0].float().abs().max())
