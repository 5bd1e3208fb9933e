"""The encoder-shape forward with its coarse levels staged whole in LDS (vnext_amd/csrc/msda_d32.hip: msda_fwd_slab_kernel)
against the CPU oracle: all rows of the BASELINE encoder shapes, and the inputs that decide which levels are staged --
pyramids that fit entirely, pyramids of which only the last level fits, none (too large, or unpacked levels: every tap a
gather) -- plus samples outside the map, queries that are not pixels (Lq != S, not a multiple of the tile), non-finite values
next to padded taps, and the fused prologue.  Development-build variants: 730 forces the kernel on any call it is built for,
731 forbids it (the per-query gather kernel), 0 = automatic (>= 2 048 queries)."""
import numpy as np
import pytest
import torch

from test_msda_tile import S360, S720, close, encoder_case, fwd, oracle

pytestmark = pytest.mark.gpu

from vnext_amd import _lib  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _auto_variant():
    _lib.set_kernel_variant(0)
    yield
    _lib.set_kernel_variant(0)


@pytest.mark.parametrize("shapes,B", [(S360, 2), (S720, 1)])
def test_all_rows_of_the_baseline_encoder_shapes(shapes, B):
    case = encoder_case(shapes, B, seed=11)
    want = oracle(*[case[i] for i in (2, 0, 1, 3, 4)])
    close(fwd(case[2], case[0], case[1], case[3], case[4], 0), want)        # automatic = the slab kernel (>= 2 048 queries)
    close(fwd(case[2], case[0], case[1], case[3], case[4], 730), want)
    close(fwd(case[2], case[0], case[1], case[3], case[4], 731), want)      # and the kernel it replaces there


@pytest.mark.parametrize("shapes", [
    [(8, 10), (4, 5), (2, 3), (1, 2)],            # 101 rows: every level staged, no gather at all
    [(30, 40), (15, 20), (8, 10), (4, 5)],        # 1 200 + 300 + 80 + 20: the last two fit (100 rows), the last three do not
    [(20, 16), (16, 20), (1, 1), (17, 1)],        # exactly 320 rows behind level 0; a 1 x 1 level; a one-column level
    [(16, 16), (20, 16), (1, 1), (1, 1)],         # 322 rows behind level 0: only the last two
    [(7, 9), (40, 50), (3, 3), (19, 17)],         # levels out of size order: 323 + 9 + 2 000 ...: only the last (323 > 320: none)
    [(33, 47), (21, 19), (9, 13), (5, 3)],        # odd sizes
])
@pytest.mark.parametrize("uniform", [False, True])
def test_pyramids_that_decide_what_is_staged(shapes, uniform):
    case = encoder_case(shapes, 2, seed=3, uniform=uniform)      # uniform: locations in [-0.1, 1.1]: samples outside the map
    want = oracle(*[case[i] for i in (2, 0, 1, 3, 4)])
    close(fwd(case[2], case[0], case[1], case[3], case[4], 730), want)


@pytest.mark.parametrize("Lq", [1, 7, 64, 65, 300, 1000])
def test_queries_that_are_not_pixels(Lq):
    sh, lsi, value, _, _ = encoder_case(S360, 2, seed=5)
    g = torch.Generator().manual_seed(Lq)
    loc = torch.rand(2, Lq, 8, 4, 4, 2, generator=g) * 1.1 - 0.05
    attn = torch.softmax(torch.randn(2, Lq, 8, 16, generator=g), -1).view(2, Lq, 8, 4, 4).contiguous()
    want = oracle(value, sh, lsi, loc, attn)
    close(fwd(value, sh, lsi, loc, attn, 730), want)


def test_unpacked_levels_are_gathered():
    """A gap between two levels (rows nobody samples): the staged suffix would not be contiguous, so nothing is staged."""
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2)]
    sh = torch.tensor(shapes, dtype=torch.long)
    sizes = sh.prod(1)
    lsi = torch.tensor([0, int(sizes[0]) + 5, int(sizes[0] + sizes[1]) + 9, int(sizes[:3].sum()) + 9], dtype=torch.long)
    S = int(lsi[3] + sizes[3]) + 4
    g = torch.Generator().manual_seed(8)
    value = torch.randn(2, S, 8, 32, generator=g)
    loc = torch.rand(2, 150, 8, 4, 4, 2, generator=g)
    attn = torch.softmax(torch.randn(2, 150, 8, 16, generator=g), -1).view(2, 150, 8, 4, 4).contiguous()
    want = oracle(value, sh, lsi, loc, attn)
    close(fwd(value, sh, lsi, loc, attn, 730), want)


def test_non_finite_values_next_to_padded_taps_do_not_leak():
    """A tap outside the map reads the slab's zero row (or an out-of-range gather): 0 x weight, whatever its neighbours hold
    (ms_deform_im2col_cuda.cuh:55-78); a NaN INSIDE the map must reach exactly the outputs the oracle says it reaches."""
    shapes = [(30, 40), (15, 20), (8, 10), (4, 5)]
    sh, lsi, value, loc, attn = encoder_case(shapes, 1, seed=9, uniform=True)
    value[0, int(lsi[3]) + 3, :, :] = float("nan")      # a staged level
    value[0, 17, :, :] = float("inf")                     # a gathered one
    want = oracle(value, sh, lsi, loc, attn)
    got = fwd(value, sh, lsi, loc, attn, 730)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=1e-5 * float(np.abs(want[fin]).max()))


def test_the_two_kernels_agree_closely_at_the_encoder_shape():
    case = encoder_case(S360, 5, seed=21)
    a = fwd(case[2], case[0], case[1], case[3], case[4], 730)
    b = fwd(case[2], case[0], case[1], case[3], case[4], 731)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * float(np.abs(b).max()))


# ---- 16-bit values (round 6): 64-byte rows, the same staging rule with twice the rows ----------------------------------------
@pytest.mark.parametrize("vdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shapes,B", [(S360, 2), (S720, 1),
                                      ([(30, 40), (20, 16), (16, 20), (1, 1)], 2),      # 641 rows behind level 0: exactly the 16-bit cap
                                      ([(30, 40), (20, 16), (16, 20), (2, 1)], 2)])     # 642: only the last two levels
def test_sixteen_bit_values_through_the_slab_kernel(vdt, shapes, B):
    """the oracle on the SAME 16-bit-rounded values (fp64 arithmetic) at the 16-bit tolerance; against the gather kernel (731),
    which reads the same values and sums in fp32, far tighter: both round their fp32 sums once on the way out"""
    case = encoder_case(shapes, B, seed=13)
    sh, lsi, value, loc, attn = case
    v16 = value.to(vdt)
    want = oracle(v16.float(), sh, lsi, loc, attn)
    tol = 8e-3 if vdt == torch.bfloat16 else 1e-3
    got = fwd(v16, sh, lsi, loc, attn, 730)
    close(got, want, tol)
    close(fwd(v16, sh, lsi, loc, attn, 0), want, tol)      # automatic: the gather kernel for unfused 16-bit calls (msda_forward_d32)
    gather = fwd(v16, sh, lsi, loc, attn, 731)
    # one unit in the last place of the output type at most (different summation orders before the single rounding)
    ulp = 2.0 ** -8 if vdt == torch.bfloat16 else 2.0 ** -11
    assert float(np.abs(got - gather).max()) <= 2 * ulp * float(np.abs(gather).max())


@pytest.mark.parametrize("vdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shapes,B", [(S720, 1),                                             # levels 2 + 3 = 1 160 rows staged (the case it was built for)
                                      ([(40, 60), (30, 30), (15, 15), (8, 8)], 2),           # 1 189 rows behind level 0: three levels staged
                                      ([(40, 60), (30, 30), (16, 16), (4, 5)], 2),           # 1 176 + ...: 900 + 256 + 20 = 1 176: three again
                                      ([(50, 50), (35, 35), (5, 5), (1, 1)], 2)])            # 1 251 behind level 0: only the last two (26 rows)
def test_the_large_slab_of_sixteen_bit_values(vdt, shapes, B):
    """Development variant 737: a 16-wave workgroup with a 77-KB slab (1 201 rows of 64 bytes) -- the two coarsest levels of a
    720p pyramid.  Measured no faster than the small slab and therefore not the product path (msda_d32.hip: use_large_slab);
    exact all the same: the oracle on the 16-bit-rounded values, and the gather kernel to one unit in the last place."""
    sh, lsi, value, loc, attn = encoder_case(shapes, B, seed=17)
    v16 = value.to(vdt)
    want = oracle(v16.float(), sh, lsi, loc, attn)
    got = fwd(v16, sh, lsi, loc, attn, 737)
    close(got, want, 8e-3 if vdt == torch.bfloat16 else 1e-3)
    gather = fwd(v16, sh, lsi, loc, attn, 731)
    ulp = 2.0 ** -8 if vdt == torch.bfloat16 else 2.0 ** -11
    assert float(np.abs(got - gather).max()) <= 2 * ulp * float(np.abs(gather).max())


def test_fused_prologue_with_bf16_values_takes_the_slab_kernel_and_matches_the_gather_form():
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors
    shapes = [tuple(x) for x in S360]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device=DEV).manual_seed(4)
    value = torch.randn(2, S, 8, 32, device=DEV, generator=g).bfloat16()
    offsets = 2.0 * torch.randn(2, S, 8, 4, 4, 2, device=DEV, generator=g)
    logits = torch.randn(2, S, 8, 16, device=DEV, generator=g)
    ref = torch.rand(2, S, 4, 2, device=DEV, generator=g)
    shapes_t, lsi = level_tensors(shapes, DEV)
    outs = {}
    for variant in (0, 731):
        _lib.set_kernel_variant(variant)
        outs[variant] = MSDeformAttnFusedFunction.apply(value, shapes_t, lsi, offsets, logits, ref).float()
        torch.cuda.synchronize()
    assert float((outs[0] - outs[731]).abs().max()) <= 2 * 2.0 ** -8 * float(outs[731].abs().max())
