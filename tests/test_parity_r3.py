"""Parity holes named by the round-2 review ("Next round" item 1):
(a) encoder-720p (BASELINE config 3's largest call: Lq = S = 19 560) forward + all three gradients, ALL rows,
    B = 2 and B = 5, model-like locations around the pixel-centre reference points, against the oracle on the
    default path (the `gv_query_splits` unit counts differ from 360p);
(b) bf16 forward / backward at the encoder-360p and encoder-720p shapes within the north star's 1e-2;
(c) the fused prologue's backward (2-d and 4-d reference points, reference_batch_div > 1) against fp64
    autograd through an INDEPENDENT composition: the reference module's expressions
    (projects/IDOL/idol/models/ops/modules/ms_deform_attn.py:99-108) followed by the grid_sample statement of
    the reference fallback (oracle/msda_torch_fallback.py, itself pinned to the reference goldens) -- not
    against the unfused HIP op.
Reference call sites: projects/IDOL/idol/models/deformable_transformer.py:249-261 (encoder reference points),
projects/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py:65-73 (the offsets' initial bias)."""
import math

import numpy as np
import pytest
import torch

from oracle import msda_oracle as O

S360 = [(48, 80), (24, 40), (12, 20), (6, 10)]
S720 = [(92, 160), (46, 80), (23, 40), (12, 20)]
DEV = "cuda:0"


def encoder_case(shapes, B, seed, M=8, P=4, noise=1.0):
    """An encoder call: the queries are the pixels of the pyramid, reference point = pixel centre
    (deformable_transformer.py:249-261 with valid ratios 1), sample = reference + (head direction x (k+1) +
    N(0, noise)) pixels of the sampled level (ms_deform_attn.py:65-73).  Samples of border pixels leave the map
    (zero padding); a few queries are thrown far away so that taps also land in distant units."""
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
    ref = torch.cat(ref, 0)                                                    # [S, 2]
    th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]                                   # [M, 2]
    k = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, 1, 1, P, 1)
    off = d.view(1, 1, M, 1, 1, 2) * k + noise * torch.randn(B, S, M, L, P, 2, generator=g)
    far = torch.rand(B, S, 1, 1, 1, 1, generator=g) < 0.01                     # 1 % of the queries sample anywhere
    off = torch.where(far, 40.0 * torch.randn(B, S, M, L, P, 2, generator=g), off)
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).float().view(1, 1, 1, L, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + off / wh).contiguous()
    value = torch.randn(B, S, M, 32, generator=g)
    attn = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).contiguous()
    go = torch.randn(B, S, M * 32, generator=g)
    return sh, lsi, value, loc, attn, go


def scale(x):
    return max(1e-30, float(np.abs(x).max()))


def off_the_pixel_grid(loc, sh, eps=1e-4):
    wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double().view(1, 1, 1, -1, 1, 2)
    px = loc.double() * wh - 0.5
    return ((px - px.round()).abs() > eps).all(-1, keepdim=True).numpy()


def oracle_all(value, sh, lsi, loc, attn, go, nthreads=16):
    args = (value.double().numpy(), sh.numpy(), lsi.numpy(), loc.double().numpy(), attn.double().numpy())
    want = O.msda_forward(*args, nthreads=nthreads)
    rv, rl, ra = O.msda_backward(*args, go.double().numpy(), nthreads=nthreads)
    return want, rv, rl, ra


# ------------------------------------------------------------------------------- (a) encoder 720p, fp32
@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 5])
def test_encoder_720p_all_rows_all_gradients_fp32(B):
    import MultiScaleDeformableAttention as MSDA
    sh, lsi, value, loc, attn, go = encoder_case(S720, B, seed=41 + B)
    dv, ds, di, dl, da, dg = (t.to(DEV) for t in (value, sh, lsi, loc, attn, go))
    out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dv, ds, di, dl, da, dg, 64)
    torch.cuda.synchronize()
    want, rv, rl, ra = oracle_all(value, sh, lsi, loc, attn, go)
    # Tolerances: the far queries of this generator sample ~100 pixels away from their reference point, where an fp32
    # pixel coordinate carries ~6e-6 px of rounding (|px| * 2^-24); times the largest difference between neighbouring
    # values (~8 for N(0,1) rows) that is ~5e-5 absolute = 2e-5 of the output's scale -- the first run had ONE of
    # 25 M elements at 1.04e-5.  3e-5 / 4e-5 of scale here; 1e-5 / 2e-5 hold on the near-reference cases
    # (test_parity_gaps.py).  The reference's own bar is rtol 1e-2 / atol 1e-3 (ops/test.py:56).
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=0, atol=3e-5 * scale(want))
    np.testing.assert_allclose(gv.double().cpu().numpy(), rv, rtol=0, atol=4e-5 * scale(rv))
    ok = off_the_pixel_grid(loc, sh)
    assert ok.mean() > 0.999
    np.testing.assert_allclose(gl.double().cpu().numpy() * ok, rl * ok, rtol=0, atol=4e-5 * scale(rl))
    np.testing.assert_allclose(ga.double().cpu().numpy(), ra, rtol=0, atol=4e-5 * scale(ra))
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-2, atol=1e-3)          # the reference's own bar, ops/test.py:56


@pytest.mark.gpu
def test_encoder_360p_pixel_grid_queries_fp32():
    """the same generator at 360p, B = 5 (the shape the bench's encoder_360p_M case times)"""
    import MultiScaleDeformableAttention as MSDA
    sh, lsi, value, loc, attn, go = encoder_case(S360, 5, seed=77)
    dv, ds, di, dl, da, dg = (t.to(DEV) for t in (value, sh, lsi, loc, attn, go))
    out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dv, ds, di, dl, da, dg, 64)
    torch.cuda.synchronize()
    want, rv, rl, ra = oracle_all(value, sh, lsi, loc, attn, go)
    ok = off_the_pixel_grid(loc, sh)
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=0, atol=3e-5 * scale(want))
    np.testing.assert_allclose(gv.double().cpu().numpy(), rv, rtol=0, atol=4e-5 * scale(rv))
    np.testing.assert_allclose(gl.double().cpu().numpy() * ok, rl * ok, rtol=0, atol=4e-5 * scale(rl))
    np.testing.assert_allclose(ga.double().cpu().numpy(), ra, rtol=0, atol=4e-5 * scale(ra))


# ------------------------------------------------------------------------------- (b) bf16 at the encoder shapes
@pytest.mark.gpu
@pytest.mark.parametrize("name,shapes,B", [("encoder_360p", S360, 5), ("encoder_720p", S720, 2)])
@pytest.mark.parametrize("loc_bf16", [False, True])
def test_bf16_forward_backward_encoder_shapes(name, shapes, B, loc_bf16):
    """value / grad_out bf16, locations fp32 (autocast) or bf16; 1e-2 of each tensor's scale against the fp64
    oracle fed with the SAME rounded inputs (the reference has no 16-bit path, ms_deform_attn_cuda.cu:64)."""
    import MultiScaleDeformableAttention as MSDA
    sh, lsi, value, loc, attn, go = encoder_case(shapes, B, seed=53)
    ld = torch.bfloat16 if loc_bf16 else torch.float32
    v16, g16 = value.bfloat16(), go.bfloat16()
    l_in, a_in = loc.to(ld), attn.to(ld)
    out = MSDA.ms_deform_attn_forward(v16.to(DEV), sh.to(DEV), lsi.to(DEV), l_in.to(DEV), a_in.to(DEV), 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v16.to(DEV), sh.to(DEV), lsi.to(DEV), l_in.to(DEV), a_in.to(DEV),
                                              g16.to(DEV), 64)
    torch.cuda.synchronize()
    assert out.dtype == gv.dtype == torch.bfloat16 and gl.dtype == ga.dtype == ld
    want, rv, rl, ra = oracle_all(v16, sh, lsi, l_in, a_in, g16)
    ok = off_the_pixel_grid(l_in, sh, eps=1e-3) if not loc_bf16 else None
    for nm, got, ref in (("out", out, want), ("grad_value", gv, rv), ("grad_loc", gl, rl), ("grad_attn", ga, ra)):
        g = got.double().cpu().numpy()
        if nm == "grad_loc":
            if loc_bf16:
                # a bf16 location has 8 bits of mantissa: many samples sit exactly ON a pixel boundary, where the
                # derivative is one-sided; compare where the kernel's fp32 and the oracle's fp64 floor agree
                wh = torch.stack([sh[:, 1], sh[:, 0]], -1).double().view(1, 1, 1, -1, 1, 2)
                px = l_in.double() * wh - 0.5
                m = ((px - px.round()).abs() > 1e-3).all(-1, keepdim=True).numpy()
                g, ref = g * m, ref * m
            else:
                g, ref = g * ok, ref * ok
        np.testing.assert_allclose(g, ref, rtol=0, atol=1e-2 * scale(ref), err_msg=f"{name} {nm}")


# ------------------------------------------------------------------------------- (c) fused backward, independent
SMALL = [(12, 20), (6, 10), (3, 5), (2, 3)]


def compose64(value, offsets, logits, ref, ref_div, shapes):
    """the reference module's expressions in fp64 (IDOL ops/modules/ms_deform_attn.py:99-108)"""
    B, Lq, M, L, P, _ = offsets.shape
    attn = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    r = ref.repeat_interleave(ref_div, 0)
    sh = torch.tensor(shapes, dtype=torch.float64)
    if ref.shape[-1] == 2:
        normalizer = torch.stack([sh[..., 1], sh[..., 0]], -1)
        loc = r[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
    else:
        loc = r[:, :, None, :, None, :2] + offsets / P * r[:, :, None, :, None, 2:] * 0.5
    return loc, attn


@pytest.mark.gpu
@pytest.mark.parametrize("shapes,B,Lq,ref_dim,ref_div,ref_grad", [
    (SMALL, 2, 37, 2, 1, True),          # 2-d references with their own gradient (IDOL encoder / first decoder layer)
    (SMALL, 4, 300, 2, 2, False),        # frames of a clip sharing one reference row (SeqFormer encoder)
    (SMALL, 3, 50, 4, 1, False),         # 4-d references (box-refined decoder layers)
    (SMALL, 6, 1200, 4, 3, False),       # 4-d, shared rows, > 1024 queries (the query-split path of grad_value)
    (S360, 2, 5100, 2, 2, False),        # encoder-360p call of a two-frame clip
    ([(7, 130), (1, 70), (30, 3), (2, 2)], 2, 1100, 2, 1, True),      # odd levels through the tile-fed grad_value path
    ([(9, 66), (3, 200), (1, 1), (12, 20)], 3, 1030, 4, 3, False),    # (blocks, flat blocks, a single pixel, bands)
])
def test_fused_backward_against_fp64_autograd_of_an_independent_composition(shapes, B, Lq, ref_dim, ref_div, ref_grad):
    from oracle.msda_torch_fallback import msda_grid_sample
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors
    g = torch.Generator().manual_seed(1000 * B + Lq + ref_dim)
    L, M, P = len(shapes), 8, 4
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, 32, generator=g)
    offsets = torch.randn(B, Lq, M, L, P, 2, generator=g) * (2.0 if ref_dim == 2 else 1.0)
    logits = torch.randn(B, Lq, M, L * P, generator=g) * 2
    ref = torch.rand(B // ref_div, Lq, L, ref_dim, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = 0.05 + 0.4 * ref[..., 2:]
    ref[0, 0, :, :2] = 1.2                                   # a query whose samples all fall outside the map
    gout = torch.randn(B, Lq, M * 32, generator=g)

    shapes_t, lsi = level_tensors(shapes, DEV)
    leaves = [value.to(DEV).requires_grad_(True), offsets.to(DEV).requires_grad_(True),
              logits.to(DEV).requires_grad_(True), ref.to(DEV).requires_grad_(ref_grad)]
    out = MSDeformAttnFusedFunction.apply(leaves[0], shapes_t, lsi, leaves[1], leaves[2], leaves[3])
    out.backward(gout.to(DEV))
    torch.cuda.synchronize()

    v64, o64, l64, r64 = (t.double().requires_grad_(True) for t in (value, offsets, logits, ref))
    loc, attn = compose64(v64, o64, l64, r64, ref_div, shapes)
    want = msda_grid_sample(v64, shapes, loc, attn)
    want.backward(gout.double())
    np.testing.assert_allclose(out.detach().double().cpu().numpy(), want.detach().numpy(), rtol=0,
                               atol=3e-5 * scale(want.detach().numpy()))
    # fp32 vs fp64 floor() can differ for a sample within ~1e-4 px of a pixel boundary; the offsets' gradient is
    # discontinuous there (see test_parity_gaps.off_the_pixel_grid); everything else is continuous
    ok = torch.from_numpy(off_the_pixel_grid(loc.detach(), torch.tensor(shapes), eps=2e-4))
    pairs = [("value", leaves[0].grad, v64.grad, None), ("offsets", leaves[1].grad, o64.grad, ok),
             ("logits", leaves[2].grad, l64.grad, None)]
    if ref_grad:
        # the reference-point gradient sums the location gradients of a level's points over heads: mask the
        # (query, level) rows that contain a boundary sample
        row_ok = ok.all(dim=2, keepdim=False).all(dim=3, keepdim=False)[..., 0]       # [B, Lq, L]
        pairs.append(("reference", leaves[3].grad, r64.grad, row_ok.unsqueeze(-1)))
    for name, got, ref_g, mask in pairs:
        got = got.double().cpu()
        if mask is not None:
            assert float(mask.double().mean()) > (0.995 if name == "offsets" else 0.9)
            got, ref_g = got * mask, ref_g * mask
        tol = 5e-5 * (float(ref_g.abs().max()) + 1e-30)
        assert float((got - ref_g).abs().max()) <= tol, name


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq", [(3, 80), (2, 1200)])      # 1 200 queries: the coarse levels are query-split (fp32 split image)
def test_fused_bf16_backward_against_fp64_composition(B, Lq):
    """bf16 value with fp32 Linear outputs (autocast): 1e-2 of scale against the same independent fp64 composition,
    fed with the rounded value / grad_out."""
    from oracle.msda_torch_fallback import msda_grid_sample
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors
    shapes, ref_dim, ref_div = SMALL, 4, 1
    g = torch.Generator().manual_seed(5)
    L, M, P = 4, 8, 4
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, 32, generator=g).bfloat16()
    offsets = torch.randn(B, Lq, M, L, P, 2, generator=g)
    logits = torch.randn(B, Lq, M, L * P, generator=g) * 2
    ref = torch.rand(B, Lq, L, ref_dim, generator=g)
    ref[..., 2:] = 0.05 + 0.4 * ref[..., 2:]
    gout = torch.randn(B, Lq, M * 32, generator=g).bfloat16()
    shapes_t, lsi = level_tensors(shapes, DEV)
    leaves = [value.to(DEV).requires_grad_(True), offsets.to(DEV).requires_grad_(True),
              logits.to(DEV).requires_grad_(True), ref.to(DEV)]
    out = MSDeformAttnFusedFunction.apply(leaves[0], shapes_t, lsi, leaves[1], leaves[2], leaves[3])
    assert out.dtype == torch.bfloat16
    out.backward(gout.to(DEV))
    torch.cuda.synchronize()
    v64, o64, l64 = (t.double().requires_grad_(True) for t in (value, offsets, logits))
    loc, attn = compose64(v64, o64, l64, ref.double(), ref_div, shapes)
    want = msda_grid_sample(v64, shapes, loc, attn)
    want.backward(gout.double())
    for name, got, ref_g in (("out", out.detach(), want.detach()), ("value", leaves[0].grad, v64.grad),
                             ("offsets", leaves[1].grad, o64.grad), ("logits", leaves[2].grad, l64.grad)):
        assert float((got.double().cpu() - ref_g).abs().max()) <= 1e-2 * (float(ref_g.abs().max()) + 1e-30), name
