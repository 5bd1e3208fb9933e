"""MSDeformAttn with the module's prologue fused in: the fused kernels against
(a) the fp64 oracle fed with numpy-computed softmax / locations and (b) autograd through the unfused
composition the reference module spells out (ops/modules/ms_deform_attn.py:99-112)."""
import numpy as np
import pytest
import torch

from oracle import msda_oracle as O

SHAPES = [(12, 20), (6, 10), (3, 5), (2, 3)]


def make(B, Lq, ref_dim, ref_div, seed, dtype=torch.float32, M=8, P=4):
    g = torch.Generator().manual_seed(seed)
    L = len(SHAPES)
    S = sum(h * w for h, w in SHAPES)
    value = torch.randn(B, S, M, 32, generator=g)
    offsets = torch.randn(B, Lq, M, L, P, 2, generator=g) * (2.0 if ref_dim == 2 else 1.0)
    logits = torch.randn(B, Lq, M, L * P, generator=g) * 2
    ref = torch.rand(B // ref_div, Lq, L, ref_dim, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = 0.05 + 0.4 * ref[..., 2:]
    ref[0, 0, :, :2] = 1.2            # a query whose samples fall outside the map
    gout = torch.randn(B, Lq, M * 32, generator=g)
    return [t.to(dtype) for t in (value, offsets, logits, ref)] + [gout.to(dtype)]


def compose(value, offsets, logits, ref, ref_div, shapes_t):
    """the reference module's expressions (IDOL ops/modules/ms_deform_attn.py:99-108)"""
    B, Lq, M, L, P, _ = offsets.shape
    attn = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    r = ref.repeat_interleave(ref_div, 0)
    if ref.shape[-1] == 2:
        normalizer = torch.stack([shapes_t[..., 1], shapes_t[..., 0]], -1)
        loc = r[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
    else:
        loc = r[:, :, None, :, None, :2] + offsets / P * r[:, :, None, :, None, 2:] * 0.5
    return loc, attn


def level_tensors(device):
    from vnext_amd.ops.functions import level_tensors as lt
    return lt(SHAPES, device)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,ref_dim,ref_div", [(2, 37, 2, 1), (4, 300, 2, 2), (3, 50, 4, 1), (6, 1200, 4, 3),
                                                   (1, 1, 2, 1)])
def test_fused_forward_against_oracle_and_composition(B, Lq, ref_dim, ref_div):
    from vnext_amd.ops.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction
    value, offsets, logits, ref, _ = make(B, Lq, ref_dim, ref_div, seed=B * 100 + Lq)
    dev = "cuda:0"
    shapes_t, lsi = level_tensors(dev)
    out = MSDeformAttnFusedFunction.apply(value.to(dev), shapes_t, lsi, offsets.to(dev), logits.to(dev), ref.to(dev))
    loc, attn = compose(value.to(dev), offsets.to(dev), logits.to(dev), ref.to(dev), ref_div, shapes_t)
    unfused = MSDeformAttnFunction.apply(value.to(dev), shapes_t, lsi, loc.contiguous(), attn.contiguous(), 64)
    scale = float(unfused.abs().max())
    assert float((out - unfused).abs().max()) <= 2e-5 * scale
    # fp64 oracle on numpy-computed softmax / locations
    loc64, attn64 = compose(value.double(), offsets.double(), logits.double(), ref.double(), ref_div,
                            torch.tensor(SHAPES, dtype=torch.long))
    want = O.msda_forward(value.double().numpy(), np.array(SHAPES, dtype=np.int64), lsi.cpu().numpy(),
                          loc64.numpy(), attn64.numpy())
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=0, atol=3e-5 * float(np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,ref_dim,ref_div,ref_grad", [(2, 37, 2, 1, True), (4, 300, 2, 2, False), (3, 50, 4, 1, False),
                                                            (5, 300, 4, 1, False)])
def test_fused_backward_against_autograd_of_the_composition(B, Lq, ref_dim, ref_div, ref_grad):
    from vnext_amd.ops.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction
    dev = "cuda:0"
    value, offsets, logits, ref, gout = [t.to(dev) for t in make(B, Lq, ref_dim, ref_div, seed=7 * B + Lq)]
    shapes_t, lsi = level_tensors(dev)

    def leaves():
        ts = [value.clone().requires_grad_(True), offsets.clone().requires_grad_(True),
              logits.clone().requires_grad_(True), ref.clone().requires_grad_(ref_grad)]
        return ts
    v, o, lg, r = leaves()
    MSDeformAttnFusedFunction.apply(v, shapes_t, lsi, o, lg, r).backward(gout)
    v2, o2, lg2, r2 = leaves()
    loc, attn = compose(v2, o2, lg2, r2, ref_div, shapes_t)
    MSDeformAttnFunction.apply(v2, shapes_t, lsi, loc.contiguous(), attn.contiguous(), 64).backward(gout)
    pairs = [("value", v.grad, v2.grad), ("offsets", o.grad, o2.grad), ("logits", lg.grad, lg2.grad)]
    if ref_grad:
        pairs.append(("reference", r.grad, r2.grad))
    for name, got, want in pairs:
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) <= 5e-5 * scale, name


@pytest.mark.gpu
def test_fused_bf16_value_with_fp32_queries():
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction
    dev = "cuda:0"
    value, offsets, logits, ref, gout = [t.to(dev) for t in make(3, 80, 4, 1, seed=5)]
    shapes_t, lsi = level_tensors(dev)
    want = MSDeformAttnFusedFunction.apply(value, shapes_t, lsi, offsets, logits, ref)
    got = MSDeformAttnFusedFunction.apply(value.bfloat16(), shapes_t, lsi, offsets, logits, ref)
    assert got.dtype == torch.bfloat16
    assert float((got.float() - want).abs().max()) <= 2e-2 * float(want.abs().max())


@pytest.mark.gpu
def test_unsupported_cases_are_reported_not_guessed():
    from vnext_amd import msda_ext
    dev = "cuda:0"
    value, offsets, logits, ref, _ = [t.to(dev) for t in make(2, 9, 2, 1, seed=1)]
    shapes_t, lsi = level_tensors(dev)
    assert msda_ext.fused_supported(value, shapes_t, offsets, logits, ref, lsi)
    assert not msda_ext.fused_supported(value, shapes_t, offsets, logits, ref, lsi.clone())          # not tagged as packed
    # a tagged pair built for ANOTHER pyramid must not be trusted (grad_value would stay uninitialised)
    from vnext_amd.ops.functions import level_tensors as lt
    other_shapes, other_lsi = lt([(5, 7), (3, 3), (2, 2), (1, 1)], dev)
    assert not msda_ext.fused_supported(value, other_shapes, offsets, logits, ref, other_lsi)
    with pytest.raises(RuntimeError, match="packed levels"):
        msda_ext.ms_deform_attn_fused_backward(value, other_shapes, other_lsi, offsets, logits, ref,
                                               torch.zeros(value.shape[0], offsets.shape[1], 256, device=dev))
    assert not msda_ext.fused_supported(value.double(), shapes_t, offsets, logits, ref, lsi)
    assert not msda_ext.fused_supported(value[..., :16].contiguous(), shapes_t, offsets, logits, ref, lsi)
    with pytest.raises(RuntimeError, match="built for 32-channel heads"):
        msda_ext.ms_deform_attn_fused_forward(value[..., :16].contiguous(), shapes_t, lsi, offsets, logits, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["idol", "seqformer_encode", "seqformer_decode"])
def test_modules_with_the_fused_prologue_equal_the_reference_form(kind):
    """return_samples=False switches a module to the fused kernels: same output and parameter
    gradients as the reference-form module (which materialises locations and weights)."""
    from vnext_amd.ops.modules import MSDeformAttnIDOL, MSDeformAttnSeqFormer
    dev = "cuda:0"
    torch.manual_seed(3)
    N, T, Lq, C = 2, 3, 40, 256
    S = sum(h * w for h, w in SHAPES)
    shapes_t, lsi = level_tensors(dev)
    if kind == "idol":
        mod = MSDeformAttnIDOL(C, 4, 8, 4).to(dev)
        args = lambda: (torch.randn(N, Lq, C, device=dev), torch.rand(N, Lq, 4, 2, device=dev),  # noqa: E731
                        torch.randn(N, S, C, device=dev), shapes_t, lsi, None)
    elif kind == "seqformer_encode":
        mod = MSDeformAttnSeqFormer(C, 4, 8, 4, "encode").to(dev)
        args = lambda: (torch.randn(N, T, Lq, C, device=dev), None, torch.rand(N, Lq, 4, 2, device=dev),  # noqa: E731
                        torch.randn(N, T, S, C, device=dev), shapes_t, lsi, None)
    else:
        mod = MSDeformAttnSeqFormer(C, 4, 8, 4, "decode").to(dev)
        args = lambda: (torch.randn(N, Lq, C, device=dev), torch.randn(N, T, Lq, C, device=dev),  # noqa: E731
                        0.1 + 0.5 * torch.rand(N, T, Lq, 4, 4, device=dev), torch.randn(N, T, S, C, device=dev),
                        shapes_t, lsi, None)
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.02)
        mod.attention_weights.weight.normal_(0, 0.05)

    def run(fused):
        mod.return_samples = not fused
        mod.zero_grad()
        torch.manual_seed(9)
        out = mod(*args())
        first = out[0] if isinstance(out, tuple) else out
        first.square().sum().backward()
        return first.detach(), {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}, out
    ref_out, ref_grads, _ = run(False)
    got_out, got_grads, raw = run(True)
    if kind != "seqformer_encode":
        assert raw[-1] is None and raw[-2] is None          # nothing materialised
    assert float((got_out - ref_out).abs().max()) <= 2e-5 * float(ref_out.abs().max())
    assert set(got_grads) == set(ref_grads)
    for n in ref_grads:
        assert float((got_grads[n] - ref_grads[n]).abs().max()) <= 2e-4 * (float(ref_grads[n].abs().max()) + 1e-12), n


@pytest.mark.gpu
@pytest.mark.parametrize("res", ["360p", "720p"])
def test_fused_at_the_encoder_shape_of_the_baseline_configs(res):
    """BASELINE configs 2-5: Lq = S, B = 5 (T folded), references shared by the frames.  Fused vs
    the unfused composition on the full tensors, and the adjoint identity
    <f(value), g> == <value, grad_value(g)> (the op is linear in `value`)."""
    from vnext_amd.ops.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction, level_tensors as lt
    shapes = {"360p": [(48, 80), (24, 40), (12, 20), (6, 10)], "720p": [(92, 160), (46, 80), (23, 40), (12, 20)]}[res]
    dev = "cuda:0"
    S = sum(h * w for h, w in shapes)
    B, T, M, L, P = 5, 5, 8, 4, 4
    g = torch.Generator(device=dev).manual_seed(17)
    value = torch.randn(B, S, M, 32, device=dev, generator=g)
    offsets = 2.0 * torch.randn(B, S, M, L, P, 2, device=dev, generator=g)
    logits = torch.randn(B, S, M, L * P, device=dev, generator=g)
    centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                         .flip(-1).reshape(-1, 2) for h, w in shapes])            # pixel centres (x, y)
    ref = centres[None, :, None, :].expand(B // T, S, L, 2).contiguous()
    shapes_t, lsi = lt(shapes, dev)
    out = MSDeformAttnFusedFunction.apply(value, shapes_t, lsi, offsets, logits, ref)
    attn = torch.softmax(logits, -1).view(B, S, M, L, P)
    normalizer = torch.stack([shapes_t[..., 1], shapes_t[..., 0]], -1)
    loc = ref.repeat_interleave(T, 0)[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
    want = MSDeformAttnFunction.apply(value, shapes_t, lsi, loc.contiguous(), attn.contiguous(), 64)
    assert float((out - want).abs().max()) <= 3e-5 * float(want.abs().max())
    gout = torch.randn(out.shape, device=dev, generator=g)
    v = value.clone().requires_grad_(True)
    MSDeformAttnFusedFunction.apply(v, shapes_t, lsi, offsets, logits, ref).backward(gout)
    lhs = float((out.double() * gout.double()).sum())
    rhs = float((value.double() * v.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("ref_dim,ref_grad", [(2, True), (4, False)])
def test_fused_backward_with_the_coarse_levels_staged_in_lds(ref_dim, ref_grad):
    """msda_bwd_slab_kernel<FUSED> (development variant 734; measured in round 6, not the product path: msda_d32.hip,
    fused_dispatch) against the gather form the product takes, on an encoder-sized call: every gradient incl. the
    reference points', ragged query count."""
    from vnext_amd import _lib
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors as lt
    shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    dev = "cuda:0"
    S = sum(h * w for h, w in shapes)
    B, M, L, P, Lq = 2, 8, 4, 4, S - 3
    g = torch.Generator(device=dev).manual_seed(29 + ref_dim)
    value = torch.randn(B, S, M, 32, device=dev, generator=g)
    offsets = 2.0 * torch.randn(B, Lq, M, L, P, 2, device=dev, generator=g)
    logits = torch.randn(B, Lq, M, L * P, device=dev, generator=g)
    ref = torch.rand(B, Lq, L, ref_dim, device=dev, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = 0.05 + 0.1 * ref[..., 2:]
    gout = torch.randn(B, Lq, M * 32, device=dev, generator=g)
    shapes_t, lsi = lt(shapes, dev)

    def grads(variant):
        v, o, lg = value.clone().requires_grad_(True), offsets.clone().requires_grad_(True), logits.clone().requires_grad_(True)
        r = ref.clone().requires_grad_(ref_grad)
        _lib.set_kernel_variant(variant)
        try:
            MSDeformAttnFusedFunction.apply(v, shapes_t, lsi, o, lg, r).backward(gout)
            torch.cuda.synchronize()
        finally:
            _lib.set_kernel_variant(0)
        return [v.grad, o.grad, lg.grad] + ([r.grad] if ref_grad else [])
    for name, got, want in zip(("value", "offsets", "logits", "reference"), grads(734), grads(0)):
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) <= 2e-5 * scale, name


@pytest.mark.gpu
@pytest.mark.parametrize("Lq,ref_dim,ref_div,ref_grad", [(300, 4, 1, False), (300, 2, 1, True), (5100, 2, 2, False), (2300, 4, 1, False)])
def test_bf16_offsets_and_logits_beside_fp32_reference_points(Lq, ref_dim, ref_div, ref_grad):
    """what torch.autocast(bfloat16) leaves at the op: bf16 value, bf16 Linear outputs, fp32 reference points
    (VNX_MSDA_REF_F32, round 6).  Against the same call with the SAME offsets / logits promoted to fp32 (what the module did
    until round 6): the forward differs by the output's rounding at most, the gradients of the 16-bit inputs by theirs."""
    from vnext_amd.ops.functions import MSDeformAttnFusedFunction, level_tensors as lt
    shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    dev = "cuda:0"
    S = sum(h * w for h, w in shapes)
    B, M, L, P = 4, 8, 4, 4
    g = torch.Generator(device=dev).manual_seed(Lq + ref_dim)
    value = torch.randn(B, S, M, 32, device=dev, generator=g).bfloat16()
    offsets = (2.0 * torch.randn(B, Lq, M, L, P, 2, device=dev, generator=g)).bfloat16()
    logits = torch.randn(B, Lq, M, L * P, device=dev, generator=g).bfloat16()
    ref = torch.rand(B // ref_div, Lq, L, ref_dim, device=dev, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = 0.05 + 0.1 * ref[..., 2:]
    gout = torch.randn(B, Lq, M * 32, device=dev, generator=g).bfloat16()
    shapes_t, lsi = lt(shapes, dev)

    def run(promote):
        v = value.clone().requires_grad_(True)
        o = (offsets.float() if promote else offsets.clone()).requires_grad_(True)
        lg = (logits.float() if promote else logits.clone()).requires_grad_(True)
        r = ref.clone().requires_grad_(ref_grad)
        out = MSDeformAttnFusedFunction.apply(v, shapes_t, lsi, o, lg, r)
        out.backward(gout)
        torch.cuda.synchronize()
        return [out, v.grad, o.grad, lg.grad] + ([r.grad] if ref_grad else [])
    got, want = run(False), run(True)
    assert got[2].dtype == torch.bfloat16 and got[3].dtype == torch.bfloat16
    for name, a, b in zip(("out", "grad_value", "grad_offsets", "grad_logits", "grad_reference"), got, want):
        scale = float(b.float().abs().max()) + 1e-12
        tol = 2e-5 if name == "grad_reference" else 1e-2      # fp32 atomics over heads vs one bf16 rounding
        assert float((a.float() - b.float()).abs().max()) <= tol * scale, name
