"""ffn_block (vnext_amd/ops/fused_ffn.py, vnext_amd/csrc/ffn_act.hip + add_norm.hip) against the expression it replaces --
`norm(x + dropout_out(linear2(dropout_mid(relu(linear1(x))))))`, the feed-forward half of every layer of the reference's
deformable transformers (projects/SeqFormer/seqformer/models/deformable_transformer.py:226-236,330-345) -- and its two
fused passes on their own."""
import pytest
import torch
import torch.nn.functional as F

from vnext_amd.ops.fused_ffn import _BiasReluDropout, ffn_block, fused_applies, linear_masked
from vnext_amd.ops.fused_norm import add_dropout_norm

DEV = "cuda:0"


def _block(p, d_ffn=1024, seed=0, train=True):
    torch.manual_seed(seed)
    l1, l2, norm = torch.nn.Linear(256, d_ffn), torch.nn.Linear(d_ffn, 256), torch.nn.LayerNorm(256)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(256)); norm.bias.copy_(0.2 * torch.randn(256))
        l1.bias.copy_(0.3 * torch.randn(d_ffn)); l2.bias.copy_(0.3 * torch.randn(256))
    d_mid, d_out = torch.nn.Dropout(p), torch.nn.Dropout(p)
    for m in (d_mid, d_out):
        m.train(train)
    return l1, d_mid, l2, d_out, norm


def test_cpu_takes_the_reference_expression():
    l1, d_mid, l2, d_out, norm = _block(0.0)
    x = torch.randn(3, 7, 256)
    assert not fused_applies(x, l1, l2, norm, F.relu)
    want = norm(x + d_out(l2(d_mid(F.relu(l1(x))))))
    assert torch.equal(ffn_block(x, l1, F.relu, d_mid, l2, d_out, norm), want)
    assert torch.equal(ffn_block(x, l1, F.gelu, d_mid, l2, d_out, norm), norm(x + l2(F.gelu(l1(x)))))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,d_ffn", [((1, 256), 1024), ((5, 300, 256), 1024), ((2, 5, 5100, 256), 1024), ((777, 256), 36),
                                         ((64, 256), 4096)])
def test_without_dropout_it_is_the_reference_block_forward_and_every_gradient(shape, d_ffn):
    l1, d_mid, l2, d_out, norm = (m.to(DEV) for m in _block(0.1, d_ffn, train=False))      # eval mode: p = 0
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
    assert fused_applies(x, l1, l2, norm, F.relu)
    y = ffn_block(x, l1, F.relu, d_mid, l2, d_out, norm)
    dbl = [torch.nn.Linear(256, d_ffn), torch.nn.Linear(d_ffn, 256), torch.nn.LayerNorm(256)]
    for a, b in zip(dbl, (l1, l2, norm)):
        a.to(DEV).double().load_state_dict({k: v.double() for k, v in b.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    want = dbl[2](xd + dbl[1](F.relu(dbl[0](xd))))
    torch.testing.assert_close(y.double(), want, rtol=0, atol=5e-6 * float(want.detach().abs().max()))
    go = torch.randn(shape, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    # the ReLU's derivative jumps at 0: a hidden unit whose pre-activation is within rounding of 0 may fall on different
    # sides in fp32 and fp64 (a handful of the 52 M units of the encoder shape do), and then the whole row of grad_x
    # differs legitimately -- compare grad_x on the rows without such a unit; the parameter gradients are sums over all
    # rows, where one flipped unit is far below the tolerance
    with torch.no_grad():
        safe = (dbl[0](xd).abs() > 1e-5).all(-1, keepdim=True)
    assert float(safe.double().mean()) > 0.98
    torch.testing.assert_close(x.grad.double() * safe, xd.grad * safe, rtol=0, atol=2e-5 * float(xd.grad.abs().max()))
    pairs = [(p.grad, q.grad) for a, b in zip((l1, l2, norm), dbl) for p, q in zip(a.parameters(), b.parameters())]
    assert len(pairs) == 6
    n_unsafe = int((~(dbl[0](xd).detach().abs() > 1e-5)).sum())          # hidden units within rounding of the ReLU's kink
    for got, ref in pairs:
        assert got is not None
        err = (got.double() - ref).abs() / float(ref.abs().max())
        # one flipped unit moves one row of linear1's weight gradient (and one element of its bias gradient) by one
        # sample's worth: allow that many rows to be off by a sample, everything else must be tight
        assert float(err.max()) < (2e-2 if n_unsafe else 5e-5)
        assert int((err > 5e-5).sum()) <= 256 * n_unsafe


@pytest.mark.gpu
def test_bias_relu_dropout_pass_mask_scaling_and_backward():
    """The middle pass on its own with p = 0.25: where relu(h + b) is positive the output is either 0 (dropped) or the
    value / (1 - p); the drop rate is p; the backward's mask is the forward's (g_h = y > 0 ? g / (1 - p) : 0) and its
    bias gradient the column sums of g_h -- bit-identical from run to run."""
    p, rows, cols = 0.25, 3001, 1024
    g = torch.Generator().manual_seed(4)
    h0 = torch.randn(rows, cols, generator=g).to(DEV)
    b = (0.5 * torch.randn(cols, generator=g)).to(DEV).requires_grad_(True)
    h = h0.clone().requires_grad_(True)
    a = _BiasReluDropout.apply(h * 1.0, b, p, 1234567, None)       # (h * 1.0: the op works in place on a fresh tensor)
    pre = torch.relu(h0 + b.detach())
    pos = pre > 0
    kept = a > 0
    assert bool((kept <= pos).all())
    rate = 1.0 - float(kept[pos].float().mean())
    assert abs(rate - p) < 0.01
    torch.testing.assert_close(a.detach(), torch.where(kept, pre / (1 - p), torch.zeros_like(pre)), rtol=1e-6, atol=0)
    go = torch.randn(rows, cols, generator=g).to(DEV)
    a.backward(go)
    want_h = torch.where(kept, go / (1 - p), torch.zeros_like(go))
    torch.testing.assert_close(h.grad, want_h, rtol=1e-6, atol=0)
    torch.testing.assert_close(b.grad.double(), want_h.double().sum(0), rtol=0, atol=2e-5 * float(want_h.double().sum(0).abs().max()))
    first = b.grad.clone()
    h2 = h0.clone().requires_grad_(True); b.grad = None
    _BiasReluDropout.apply(h2 * 1.0, b, p, 1234567, None).backward(go)
    assert torch.equal(b.grad, first) and torch.equal(h2.grad, h.grad)          # same seed -> same mask, fixed-order sums
    h3 = h0.clone()
    a3 = _BiasReluDropout.apply(h3, b.detach(), p, 7654321, None)
    assert not torch.equal(a3 > 0, kept)                                        # another seed -> another mask


@pytest.mark.gpu
def test_a_nan_activation_lets_its_gradient_through_like_aten():
    """threshold_backward is `x <= 0 ? 0 : g`: a NaN activation is not <= 0, so its gradient passes.  The fused backward
    tested `y > 0` until round 4 and zeroed it -- a diverging run looked healthy one layer further down."""
    g = torch.Generator().manual_seed(9)
    h0 = torch.randn(64, 256, generator=g).to(DEV)
    h0[3, 17] = float("nan")
    h0[40, 200] = float("inf")
    b = torch.zeros(256, device=DEV, requires_grad=True)
    h = h0.clone().requires_grad_(True)
    _BiasReluDropout.apply(h * 1.0, b, 0.0, 1, None).backward(torch.ones(64, 256, device=DEV))
    ref = h0.clone().requires_grad_(True)
    torch.relu(ref).backward(torch.ones(64, 256, device=DEV))
    assert torch.equal(h.grad, ref.grad)
    assert float(h.grad[3, 17]) == 1.0 and float(h.grad[40, 200]) == 1.0


@pytest.mark.gpu
def test_add_norm_with_a_folded_bias_returns_its_gradient():
    """add_dropout_norm(x, r, ..., r_bias=b) = norm(x + dropout(r + b)); p = 0 against fp64 autograd, and p > 0 against
    the same call with the bias added by torch beforehand and the same seed (same mask): outputs and gradients equal,
    grad_b = column sums of grad_r."""
    torch.manual_seed(0)
    norm = torch.nn.LayerNorm(256).to(DEV)
    rows = 2049
    g = torch.Generator().manual_seed(6)
    x = torch.randn(rows, 256, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(rows, 256, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(rows, 256, generator=g).to(DEV)
    drop = torch.nn.Dropout(0.0)
    y = add_dropout_norm(x, r, drop, norm, r_bias=b)
    y.backward(go)
    xd, rd, bd = (t.detach().double().requires_grad_(True) for t in (x, r, b))
    nd = torch.nn.LayerNorm(256).to(DEV).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    want = nd(xd + rd + bd)
    want.backward(go.double())
    torch.testing.assert_close(y.double(), want, rtol=0, atol=2e-6 * float(want.detach().abs().max()))
    for got, ref in ((x.grad, xd.grad), (r.grad, rd.grad), (b.grad, bd.grad), (norm.weight.grad, nd.weight.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=0, atol=5e-6 * float(ref.abs().max()))
    # p > 0: same seed, bias folded vs bias added up front
    drop = torch.nn.Dropout(0.2).train()
    grads = []
    for folded in (True, False):
        for t in (x, r, b):
            t.grad = None
        out = add_dropout_norm(x, r, drop, norm, seed=99, r_bias=b) if folded else add_dropout_norm(x, r + b, drop, norm, seed=99)
        out.backward(go)
        grads.append((out.detach().clone(), x.grad.clone(), r.grad.clone(), b.grad.clone()))
    for got, ref in zip(*grads):
        torch.testing.assert_close(got, ref, rtol=0, atol=1e-5 * float(ref.abs().max()))
    torch.testing.assert_close(grads[0][3].double(), grads[0][2].double().sum(0), rtol=0, atol=1e-5 * float(grads[0][2].double().sum(0).abs().max()))


@pytest.mark.gpu
def test_training_mode_block_drops_at_both_sites_and_stays_finite():
    l1, d_mid, l2, d_out, norm = (m.to(DEV) for m in _block(0.1, train=True))
    x = torch.randn(4, 300, 256, device=DEV, requires_grad=True)
    y1 = ffn_block(x, l1, F.relu, d_mid, l2, d_out, norm)
    y2 = ffn_block(x, l1, F.relu, d_mid, l2, d_out, norm)
    assert not torch.equal(y1, y2)                       # fresh masks per call
    (y1.square().mean() + y2.square().mean()).backward()
    for t in (x.grad, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad, norm.weight.grad, norm.bias.grad):
        assert t is not None and bool(torch.isfinite(t).all()) and float(t.abs().sum()) > 0
    # the mean of many stochastic passes approaches the deterministic block (inverted dropout is unbiased before the norm)
    for m in (d_mid, d_out):
        m.eval()
    base = ffn_block(x, l1, F.relu, d_mid, l2, d_out, norm)
    assert float((y1 - base).abs().mean()) < 0.5


def test_linear_masked_on_cpu_is_the_expression():
    lin = torch.nn.Linear(256, 256)
    x = torch.randn(2, 3, 50, 256)
    mask = torch.rand(2, 3, 50) < 0.2
    assert torch.equal(linear_masked(x, lin, mask), lin(x).masked_fill(mask[..., None], 0.0))
    assert torch.equal(linear_masked(x, lin, None), lin(x))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 5, 5100), (3, 77)])
def test_linear_masked_forward_and_gradients(shape):
    """value = value_proj(x).masked_fill(mask[..., None], 0) (ms_deform_attn.py:94-96): forward, grad_x, grad_W and the
    bias gradient that comes out of the masking pass, against fp64 autograd through the expression itself."""
    torch.manual_seed(1)
    lin = torch.nn.Linear(256, 256).to(DEV)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(*shape, 256, generator=g).to(DEV).requires_grad_(True)
    mask = (torch.rand(*shape, generator=g) < 0.15).to(DEV)
    mask[..., -3:] = True                                   # a padded tail, as the 384-row frames of a 360p clip have
    y = linear_masked(x, lin, mask)
    assert bool((y[mask] == 0).all())
    go = torch.randn(*shape, 256, generator=g).to(DEV)
    y.backward(go)
    ld = torch.nn.Linear(256, 256).to(DEV).double()
    ld.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    want = ld(xd).masked_fill(mask[..., None], 0.0)
    want.backward(go.double())
    torch.testing.assert_close(y.double(), want, rtol=0, atol=3e-6 * float(want.detach().abs().max()))
    for got, ref in ((x.grad, xd.grad), (lin.weight.grad, ld.weight.grad), (lin.bias.grad, ld.bias.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=0, atol=2e-5 * float(ref.abs().max()))
    with torch.no_grad():                                    # inference: same values, nothing recorded
        torch.testing.assert_close(linear_masked(x.detach(), lin, mask), y.detach(), rtol=0, atol=0)
