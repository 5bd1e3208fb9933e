"""The spatially tiled, LDS-staged encoder-shape forward (vnext_amd/csrc/msda_d32_tile.hip) against the
CPU oracle over ALL rows of the BASELINE encoder shapes, and on the inputs that exercise its special
paths: non-dyadic pyramids (cells of uneven size), taps outside the window (global fetch), samples
outside the map (zero padding, ms_deform_im2col_cuda.cuh:55-78,288), unpacked levels and Lq != S
(linear blocks), non-finite values next to padded taps, the fused prologue (ops/modules/ms_deform_attn.py:
99-112).  Everything goes through the C ABI; variant 700 = the first tiled kernel, 720 = the second
(vnext_amd/csrc/msda_d32_tile2.hip: 16 x 8 cells, one sample per 8-lane set, next item prefetched), 710 = the
per-query gather kernel, 0 = automatic selection."""
import numpy as np
import pytest
import torch

from oracle import msda_oracle as O

pytestmark = pytest.mark.gpu

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from vnext_amd import _lib, msda_ext  # noqa: E402
from vnext_amd.ops.functions import level_tensors  # noqa: E402

DEV = "cuda:0"
TILED = [700, 720]
S360 = [(48, 80), (24, 40), (12, 20), (6, 10)]
S720 = [(92, 160), (46, 80), (23, 40), (12, 20)]


@pytest.fixture(autouse=True)
def _auto_variant():
    _lib.set_kernel_variant(0)
    yield
    _lib.set_kernel_variant(0)


def pixel_centres(shapes):
    """[S, 2] (x, y) of every pixel of the pyramid (deformable_transformer.py:183-190)."""
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
    return torch.cat(refs, 0)


def encoder_case(shapes, B, seed, spread=1.0, uniform=False, M=8):
    """value, loc, attn for an encoder call: the query is a pixel; offsets = head direction x (k + 1) +
    N(0, spread) pixels (the module's initialisation, ops/modules/ms_deform_attn.py:65-73)."""
    g = torch.Generator().manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.long)
    S = int(sh.prod(1).sum())
    lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, M, 32, generator=g)
    if uniform:
        loc = torch.rand(B, S, M, 4, 4, 2, generator=g) * 1.2 - 0.1
    else:
        ref = pixel_centres(shapes).view(1, S, 1, 1, 1, 2)
        th = torch.arange(M) * (2 * np.pi / M)
        d = torch.stack([th.cos(), th.sin()], -1)
        d = d / d.abs().max(-1, keepdim=True)[0]
        k = torch.arange(1, 5).view(1, 1, 1, 1, 4, 1)
        offs = d.view(1, 1, M, 1, 1, 2) * k + spread * torch.randn(B, S, M, 4, 4, 2, generator=g)
        wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float().view(1, 1, 1, 4, 1, 2)
        loc = (ref + offs / wh).contiguous()
    attn = torch.softmax(torch.randn(B, S, M, 16, generator=g), -1).view(B, S, M, 4, 4).contiguous()
    return sh, lsi, value, loc, attn


def fwd(value, sh, lsi, loc, attn, variant):
    _lib.set_kernel_variant(variant)
    out = MSDA.ms_deform_attn_forward(value.to(DEV), sh.to(DEV), lsi.to(DEV), loc.to(DEV), attn.to(DEV), 64)
    torch.cuda.synchronize()
    return out.double().cpu().numpy()


def oracle(value, sh, lsi, loc, attn):
    return O.msda_forward(value.double().numpy(), sh.numpy(), lsi.numpy(), loc.double().numpy(),
                          attn.double().numpy(), nthreads=8)


def close(got, want, tol=1e-5):
    s = max(1e-30, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * s)


@pytest.mark.parametrize("shapes,B", [(S360, 5), (S720, 2)])
def test_encoder_shapes_all_rows_against_the_oracle(shapes, B):
    """BASELINE encoder shapes (Lq = S; 360p: B = T = 5), every output row."""
    sh, lsi, value, loc, attn = encoder_case(shapes, B, seed=11)
    want = oracle(value, sh, lsi, loc, attn)
    for variant in (0, 700, 710, 720):
        close(fwd(value, sh, lsi, loc, attn, variant), want)


@pytest.mark.parametrize("shapes", [
    [(13, 17), (7, 9), (4, 5), (2, 3)],        # odd sizes: cells of uneven size at every level
    [(9, 40), (5, 20), (3, 10), (2, 5)],       # wide and flat
    [(8, 8), (8, 8), (8, 8), (8, 8)],          # not a pyramid at all: every level the same size
    [(1, 1), (1, 1), (1, 1), (1, 1)],          # one pixel per level
    [(3, 5), (20, 31), (2, 2), (7, 3)],        # the finest level is not level 0
])
@pytest.mark.parametrize("uniform", [False, True])
@pytest.mark.parametrize("tiled", TILED)
def test_uneven_pyramids(shapes, uniform, tiled):
    sh, lsi, value, loc, attn = encoder_case(shapes, 3, seed=5, uniform=uniform)
    close(fwd(value, sh, lsi, loc, attn, tiled), oracle(value, sh, lsi, loc, attn))


@pytest.mark.parametrize("spread", [4.0, 25.0])
@pytest.mark.parametrize("tiled", TILED)
def test_taps_outside_the_window_come_from_global_memory(spread, tiled):
    """Large offsets: most samples leave the window of their cell (and many the map)."""
    sh, lsi, value, loc, attn = encoder_case(S360, 2, seed=7, spread=spread)
    close(fwd(value, sh, lsi, loc, attn, tiled), oracle(value, sh, lsi, loc, attn))


@pytest.mark.parametrize("tiled", TILED)
def test_linear_blocks_when_the_queries_are_not_the_pixels(tiled):
    """Lq != S (a decoder call forced onto the tiled kernel) and unpacked levels (gaps between them):
    the kernel falls back to blocks of 64 consecutive queries; the result must not change."""
    g = torch.Generator().manual_seed(3)
    sh = torch.tensor(S360, dtype=torch.long)
    S = int(sh.prod(1).sum())
    lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(2, S, 8, 32, generator=g)
    loc = torch.rand(2, 333, 8, 4, 4, 2, generator=g)
    attn = torch.softmax(torch.randn(2, 333, 8, 16, generator=g), -1).view(2, 333, 8, 4, 4)
    close(fwd(value, sh, lsi, loc, attn, tiled), oracle(value, sh, lsi, loc, attn))
    # unpacked: 7 unused rows in front of every level, Lq == S_padded
    gaps = torch.arange(1, 5) * 7
    lsi2 = lsi + gaps
    S2 = S + int(gaps[-1])
    value2 = torch.randn(2, S2, 8, 32, generator=g)
    loc2 = torch.rand(2, S2, 8, 4, 4, 2, generator=g)
    attn2 = torch.softmax(torch.randn(2, S2, 8, 16, generator=g), -1).view(2, S2, 8, 4, 4)
    close(fwd(value2, sh, lsi2, loc2, attn2, tiled), oracle(value2, sh, lsi2, loc2, attn2))


@pytest.mark.parametrize("tiled", TILED)
def test_padded_taps_never_touch_the_data(tiled):
    """Samples outside the map and NaN locations contribute exactly nothing, also when `value` holds
    non-finite numbers elsewhere: only queries whose taps really read those rows may see them."""
    sh, lsi, value, loc, attn = encoder_case(S360, 1, seed=9)
    value[0, 0] = float("inf")                     # pixel (0, 0) of level 0, all heads
    loc[0, 4000:4010] = float("nan")               # ten queries with NaN locations
    loc[0, 4010:4020] = 7.0                        # ten queries far outside the map
    got = fwd(value, sh, lsi, loc, attn, tiled)
    alt = fwd(value, sh, lsi, loc, attn, 710)
    assert np.all(got[0, 4000:4020] == 0.0) and np.all(alt[0, 4000:4020] == 0.0)
    finite = np.isfinite(alt)
    assert np.array_equal(finite, np.isfinite(got))      # the same queries see the infinity
    close(got[finite], alt[finite], 2e-5)


@pytest.mark.parametrize("ref_dim,ref_div", [(2, 1), (2, 5), (4, 1)])
def test_fused_prologue_on_the_tiled_kernel(ref_dim, ref_div):
    """vnx_msda_fused_forward at the encoder shape: raw offsets + logits + reference points, the frames
    of a clip sharing one reference row (ref_div = T), against the oracle fed with the module's own
    expressions evaluated in float64, and against the per-query fused kernel."""
    B, M, L, P = 5, 8, 4, 4
    S = sum(h * w for h, w in S360)
    g = torch.Generator().manual_seed(21)
    value = torch.randn(B, S, M, 32, generator=g)
    offsets = 2.0 * torch.randn(B, S, M, L, P, 2, generator=g)
    logits = 2.0 * torch.randn(B, S, M, L * P, generator=g)
    centres = pixel_centres(S360)
    ref = centres[None, :, None, :].expand(B // ref_div, S, L, 2).contiguous()
    if ref_dim == 4:
        ref = torch.cat([ref, 0.02 + 0.1 * torch.rand(B // ref_div, S, L, 2, generator=g)], -1).contiguous()
    shapes_t, lsi_t = level_tensors(S360, DEV)
    outs = {}
    for variant in (700, 710):
        _lib.set_kernel_variant(variant)
        outs[variant] = msda_ext.ms_deform_attn_fused_forward(value.to(DEV), shapes_t, lsi_t, offsets.to(DEV),
                                                              logits.to(DEV), ref.to(DEV)).double().cpu().numpy()
    attn = torch.softmax(logits.double(), -1).view(B, S, M, L, P)
    r = ref.double().repeat_interleave(ref_div, 0)
    sh = torch.tensor(S360, dtype=torch.long)
    if ref_dim == 2:
        wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
        loc = r[:, :, None, :, None, :] + offsets.double() / wh[None, None, None, :, None, :]
    else:
        loc = r[:, :, None, :, None, :2] + offsets.double() / P * r[:, :, None, :, None, 2:] * 0.5
    lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    want = O.msda_forward(value.double().numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), attn.numpy(), nthreads=8)
    # fp32 location arithmetic moves a tap by ~1e-7 of the map; 5e-5 of the output scale covers it
    close(outs[700], want, 5e-5)
    close(outs[710], want, 5e-5)


@pytest.mark.parametrize("tiled", TILED)
def test_graph_capture_and_repeatability(tiled):
    """No allocation, no synchronisation inside the call; same bits on every replay."""
    sh, lsi, value, loc, attn = encoder_case(S360, 2, seed=2)
    dv, ds, di, dl, da = (t.to(DEV) for t in (value, sh, lsi, loc, attn))
    _lib.set_kernel_variant(tiled)
    first = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first)
