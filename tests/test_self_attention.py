"""query_self_attention / query_self_attention_block (vnext_amd/ops/self_attention.py, vnext_amd/csrc/self_attn.hip) against the
expression they replace: the decoder layers' `self_attn(q, k, tgt)[0]` with q = k = tgt + query_pos on an
nn.MultiheadAttention, closed by `norm2(tgt + dropout2(.))` (projects/SeqFormer/seqformer/models/deformable_transformer.py:
286-323 and the `_box` twin; IDOL's decoder layer)."""
import math

import pytest
import torch

from vnext_amd.ops import self_attention as SA
from vnext_amd.ops.self_attention import _QuerySelfAttention, query_self_attention, query_self_attention_block

DEV = "cuda:0"


def _mha(C=256, H=8, p=0.1, seed=0, train=False, device="cpu", dtype=torch.float32):
    torch.manual_seed(seed)
    m = torch.nn.MultiheadAttention(C, H, dropout=p)
    with torch.no_grad():
        m.in_proj_bias.copy_(0.3 * torch.randn(3 * C))
        m.out_proj.bias.copy_(0.2 * torch.randn(C))
        m.in_proj_weight.mul_(2.0)
    m.train(train)
    return m.to(device=device, dtype=dtype)


def _reference(x, pos, mha):
    """The reference expression in the module's own precision, seq-first as the reference calls it."""
    if pos is None:
        qk = x
    else:
        t = x.shape[0] // pos.shape[0]
        qk = x + (pos if t == 1 else pos.unsqueeze(1).expand(-1, t, -1, -1).reshape(x.shape))
    return mha(qk.transpose(0, 1), qk.transpose(0, 1), x.transpose(0, 1))[0].transpose(0, 1)


def test_cpu_takes_the_reference_expression():
    mha = _mha()
    x, pos = torch.randn(2, 9, 256), torch.randn(2, 9, 256)
    assert not SA.fused_applies(x, pos, mha)
    assert torch.equal(query_self_attention(x, pos, mha), _reference(x, pos, mha))
    norm, drop = torch.nn.LayerNorm(256), torch.nn.Dropout(0.0)
    assert torch.equal(query_self_attention_block(x, pos, mha, drop, norm), norm(x + _reference(x, pos, mha)))
    # frames sharing one position block: [N * T, Q, C] queries against [N, Q, C] positions
    xb = torch.randn(6, 9, 256)
    want = _reference(xb, pos, mha)
    torch.testing.assert_close(query_self_attention(xb, pos, mha), want)


def _compare(B, Q, C, H, n_pos, seed):
    """p = 0: forward and every gradient against the module evaluated in fp64."""
    mha = _mha(C, H, p=0.1, seed=seed, train=False, device=DEV)
    ref = _mha(C, H, p=0.1, seed=seed, train=False, device=DEV, dtype=torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, Q, C, generator=g).to(DEV).requires_grad_(True)
    pos = None if n_pos == 0 else torch.randn(n_pos, Q, C, generator=g).to(DEV).requires_grad_(True)
    assert SA.fused_applies(x, pos, mha)
    y = query_self_attention(x, pos, mha)
    xd = x.detach().double().requires_grad_(True)
    pd = None if pos is None else pos.detach().double().requires_grad_(True)
    want = _reference(xd, pd, ref)
    tol = 3e-6 * float(want.detach().abs().max()) * math.sqrt(C / 32)
    torch.testing.assert_close(y.double(), want, rtol=0, atol=tol)
    go = torch.randn(B, Q, C, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    pairs = [(x.grad, xd.grad), (mha.in_proj_weight.grad, ref.in_proj_weight.grad), (mha.in_proj_bias.grad, ref.in_proj_bias.grad),
             (mha.out_proj.weight.grad, ref.out_proj.weight.grad), (mha.out_proj.bias.grad, ref.out_proj.bias.grad)]
    if pos is not None:
        pairs.append((pos.grad, pd.grad))
    for got, r in pairs:
        # (+ 1e-5: with one query the q / k / position gradients are exactly zero and fp32 leaves rounding noise there)
        torch.testing.assert_close(got.double(), r, rtol=0, atol=2e-5 * float(r.abs().max()) + 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Q,C,H,n_pos", [
    (2, 300, 256, 8, 2),          # the mask / class queries of two clips
    (10, 300, 256, 8, 2),         # the box queries: five frames of a clip share its positions
    (10, 300, 256, 8, 10),
    (1, 300, 256, 8, 0),          # no positions
    (3, 1, 256, 8, 3), (2, 7, 256, 8, 1), (2, 32, 256, 8, 2), (2, 33, 256, 8, 2), (1, 64, 256, 8, 1), (2, 65, 256, 8, 2),
    (1, 129, 256, 8, 1),
    (2, 100, 128, 4, 2), (1, 50, 32, 1, 1), (2, 40, 512, 16, 1),
])
def test_without_dropout_it_is_the_modules_attention(B, Q, C, H, n_pos):
    _compare(B, Q, C, H, n_pos, seed=B * 1000 + Q)


@pytest.mark.gpu
def test_block_equals_norm_of_residual_and_attention():
    mha = _mha(device=DEV)
    torch.manual_seed(3)
    norm = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(256, device=DEV))
        norm.bias.copy_(0.2 * torch.randn(256, device=DEV))
    drop = torch.nn.Dropout(0.1).eval()
    x = torch.randn(10, 300, 256, device=DEV, requires_grad=True)
    pos = torch.randn(2, 300, 256, device=DEV, requires_grad=True)
    y = query_self_attention_block(x, pos, mha, drop, norm)
    go = torch.randn_like(y)
    y.backward(go)
    got = [t.grad.clone() for t in (x, pos, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                                    norm.weight, norm.bias)]
    for t in (x, pos, *mha.parameters(), *norm.parameters()):
        t.grad = None
    want = norm(x + _reference(x, pos, mha))
    torch.testing.assert_close(y, want, rtol=0, atol=3e-5)
    want.backward(go)
    ref = [t.grad for t in (x, pos, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                            norm.weight, norm.bias)]
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=0, atol=3e-5 * float(b.abs().max()) + 1e-6)


def _core(x, pos, mha, p, seed, t=1):
    return _QuerySelfAttention.apply(x, pos, mha.in_proj_weight, mha.in_proj_bias, mha.num_heads, p, seed, None, t)


@pytest.mark.gpu
def test_dropout_mask_read_back_then_forward_and_backward_follow_it():
    """Q <= 32: with Wq = Wk = 0 (uniform probabilities) and v_j = e_j the context IS the kept / dropped pattern of a row,
    so the mask of a (seed, batch, head) can be read off the output; the same seed with real weights must then equal
    torch's attention evaluated with exactly that mask, forward and backward."""
    B, Q, C, H, p, seed = 2, 32, 64, 2, 0.25, 1234567
    probe = _mha(C, H, p=p, device=DEV)
    with torch.no_grad():
        probe.in_proj_weight.zero_()
        probe.in_proj_bias.zero_()
        for h in range(H):                                  # v = x Wv^T: row j of x = e_j in every head's 32 channels
            probe.in_proj_weight[2 * C + h * 32:2 * C + (h + 1) * 32, :32] = torch.eye(32, device=DEV)
    x_probe = torch.zeros(B, Q, C, device=DEV)
    x_probe[:, torch.arange(Q), torch.arange(Q)] = 1.0
    ctx = _core(x_probe, None, probe, p, seed).detach()                          # [B, Q, C]: ctx[b, i, h*32 + j] = M / ((1-p) Q)
    mask = (ctx.view(B, Q, H, 32).permute(0, 2, 1, 3)[..., :Q] * Q * (1 - p)).round()      # [B, H, Q(i), Q(j)]
    assert set(mask.unique().tolist()) <= {0.0, 1.0}
    rate = 1.0 - float(mask.mean())
    assert abs(rate - p) < 4 * math.sqrt(p * (1 - p) / mask.numel()) + 0.01, rate
    assert not torch.equal(mask[0, 0], mask[0, 1]) and not torch.equal(mask[0], mask[1])
    assert torch.equal(ctx, _core(x_probe, None, probe, p, seed))                # the same seed: the same mask
    assert not torch.equal(ctx, _core(x_probe, None, probe, p, seed + 1))

    mha = _mha(C, H, p=p, seed=5, device=DEV)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, Q, C, generator=g).to(DEV).requires_grad_(True)
    pos = torch.randn(B, Q, C, generator=g).to(DEV).requires_grad_(True)
    y = _core(x, pos, mha, p, seed)
    go = torch.randn(B, Q, C, generator=g).to(DEV)
    y.backward(go)
    got = [t.grad.clone() for t in (x, pos, mha.in_proj_weight, mha.in_proj_bias)]
    for t in (x, pos, mha.in_proj_weight, mha.in_proj_bias):
        t.grad = None
    # torch, fp64, with the mask read above
    xd, pd = x.detach().double().requires_grad_(True), pos.detach().double().requires_grad_(True)
    w, bias = mha.in_proj_weight.detach().double().requires_grad_(True), mha.in_proj_bias.detach().double().requires_grad_(True)
    qk = (xd + pd) @ w[:2 * C].t() + bias[:2 * C]
    v = xd @ w[2 * C:].t() + bias[2 * C:]
    q, k = qk[..., :C], qk[..., C:]
    heads = lambda z: z.view(B, Q, H, 32).permute(0, 2, 1, 3)
    a = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(32), -1) * mask.double() / (1 - p)
    want = (a @ heads(v)).permute(0, 2, 1, 3).reshape(B, Q, C)
    torch.testing.assert_close(y.double(), want, rtol=0, atol=3e-6 * float(want.abs().max()))
    want.backward(go.double())
    for a_, b_ in zip(got, (xd.grad, pd.grad, w.grad, bias.grad)):
        torch.testing.assert_close(a_.double(), b_, rtol=0, atol=2e-5 * float(b_.abs().max()))


@pytest.mark.gpu
def test_training_mode_drops_and_keeps_the_expectation():
    """300 queries, p = 0.1, training: outputs differ from call to call, their mean over many masks approaches the p = 0
    result, and the backward of a call is the derivative of THAT call (finite differences with the seed held fixed)."""
    mha = _mha(p=0.1, train=True, device=DEV)
    x = torch.randn(2, 300, 256, device=DEV)
    pos = torch.randn(2, 300, 256, device=DEV)
    base = _core(x, pos, mha, 0.0, 1)
    a, b = query_self_attention(x, pos, mha), query_self_attention(x, pos, mha)
    assert not torch.equal(a, b)
    acc = torch.zeros_like(base)
    for s in range(64):
        acc += _core(x, pos, mha, 0.1, 1000 + s)
    err = float((acc / 64 - base).abs().mean()) / float(base.abs().mean())
    assert err < 0.08, err
    # directional derivative at a fixed seed
    small = _mha(64, 2, p=0.2, seed=2, train=True, device=DEV)
    xs = torch.randn(1, 24, 64, device=DEV, requires_grad=True)
    ps = torch.randn(1, 24, 64, device=DEV)
    r = torch.randn(1, 24, 64, device=DEV)
    (_core(xs, ps, small, 0.2, 77) * r).sum().backward()
    dx = torch.randn_like(xs)
    eps = 1e-2
    with torch.no_grad():
        fd = ((_core(xs + eps * dx, ps, small, 0.2, 77) * r).sum() - (_core(xs - eps * dx, ps, small, 0.2, 77) * r).sum()) / (2 * eps)
    an = float((xs.grad * dx).sum())
    assert abs(float(fd) - an) <= 2e-2 * max(1.0, abs(an)), (float(fd), an)


@pytest.mark.gpu
def test_bad_arguments_are_reported():
    from vnext_amd import _lib
    lib = _lib.lib()
    t = torch.zeros(4, 3 * 64, device=DEV)
    o = torch.zeros(4, 64, device=DEV)
    l = torch.zeros(8, device=DEV)
    st = lib.vnx_query_self_attention_forward(_lib.VNX_F32, t.data_ptr(), None, o.data_ptr(), l.data_ptr(), 1, 4, 4, 16, 192, 0.0, 0,
                                              None, _lib.current_stream(t))
    assert st == 2 and "32 channels" in lib.vnx_last_error().decode()
    st = lib.vnx_query_self_attention_forward(_lib.VNX_F32, t.data_ptr(), None, o.data_ptr(), l.data_ptr(), 1, 4, 2, 32, 100, 0.0, 0,
                                              None, _lib.current_stream(t))
    assert st == 1
    st = lib.vnx_query_self_attention_forward(_lib.VNX_F32, None, None, o.data_ptr(), l.data_ptr(), 1, 4, 2, 32, 192, 0.0, 0,
                                              None, _lib.current_stream(t))
    assert st == 1


@pytest.mark.gpu
def test_captured_training_replays_draw_fresh_masks_and_backward_follows():
    """As for the other fused dropout sites (tests/test_fused_norm.py): inside a `step_scope` a captured forward + backward of
    the attention draws a fresh mask on every replay (a device-side step seed bumped by a captured add), and the backward of
    a replay recomputes the mask of ITS forward.  Q = 32 with the mask-revealing weights of the read-back test above."""
    from vnext_amd.ops.fused_norm import step_scope
    B, Q, C, H, p = 2, 32, 64, 2, 0.25
    probe = _mha(C, H, p=p, train=True, device=DEV)
    with torch.no_grad():
        probe.in_proj_weight.zero_()
        probe.in_proj_bias.zero_()
        for h in range(H):
            probe.in_proj_weight[2 * C + h * 32:2 * C + (h + 1) * 32, :32] = torch.eye(32, device=DEV)
        probe.out_proj.weight.copy_(torch.eye(C, device=DEV))
        probe.out_proj.bias.zero_()
    x0 = torch.zeros(B, Q, C, device=DEV)
    x0[:, torch.arange(Q), torch.arange(Q)] = 1.0

    class Site(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mha = probe

        def forward(self, xx):
            with step_scope(xx.device):
                return query_self_attention(xx, None, self.mha)

    site = Site().train()
    graphed = torch.cuda.make_graphed_callables(site, (x0.clone().requires_grad_(True),))
    go = torch.randn(B, Q, C, device=DEV)
    outs, grads = [], []
    for _ in range(3):
        xi = x0.clone().requires_grad_(True)
        y = graphed(xi)
        y.backward(go)
        torch.cuda.synchronize()
        outs.append(y.detach().clone())
        grads.append(xi.grad.clone())
    masks = [(o.view(B, Q, H, 32).permute(0, 2, 1, 3)[..., :Q] * Q * (1 - p)).round() for o in outs]
    for m in masks:
        assert set(m.unique().tolist()) <= {0.0, 1.0} and abs(1.0 - float(m.mean()) - p) < 0.05
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    # the backward of replay i used mask i: with uniform probabilities and identity projections
    #   out[b, i, h, :] = sum_j M_ij v_j / ((1 - p) Q),  v = x Wv^T  =>  grad_x = Wv^T-projected sum_i M_ij go_i / ((1 - p) Q)
    for m, g in zip(masks, grads):
        go_h = go.view(B, Q, H, 32).permute(0, 2, 1, 3)                                     # [B, H, Q(i), 32]
        dv = (m.transpose(-1, -2) / ((1 - p) * Q)) @ go_h                                 # [B, H, Q(j), 32]
        want = dv.sum(1)                                                                   # every head's Wv block reads x[:, :, :32]
        torch.testing.assert_close(g[..., :32], want, rtol=0, atol=2e-6)
        assert float(g[..., 32:].abs().max()) == 0.0
