"""add_dropout_norm (vnext_amd/ops/fused_norm.py, vnext_amd/csrc/add_norm.hip) against the expression it replaces,
`norm(x + dropout(x2))` of the reference's transformer layers (deformable_transformer.py:201-236,286-385)."""
import pytest
import torch

from vnext_amd.ops.fused_norm import add_dropout_norm, fused_applies

DEV = "cuda:0"


def _modules(p, seed=0, train=True):
    torch.manual_seed(seed)
    norm = torch.nn.LayerNorm(256)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(256))
        norm.bias.copy_(0.2 * torch.randn(256))
    drop = torch.nn.Dropout(p)
    drop.train(train)
    return drop, norm


def test_cpu_and_other_widths_take_the_reference_expression():
    drop, norm = _modules(0.0)
    x, r = torch.randn(3, 7, 256), torch.randn(3, 7, 256)
    assert not fused_applies(x, r, norm)
    assert torch.equal(add_dropout_norm(x, r, drop, norm), norm(x + drop(r)))
    n2 = torch.nn.LayerNorm(64)
    assert not fused_applies(torch.randn(2, 64), torch.randn(2, 64), n2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 256), (5, 300, 256), (2, 5, 5100, 256), (4097, 256)])
def test_without_dropout_it_is_the_layer_norm_of_the_sum(shape):
    drop, norm = _modules(0.1, train=False)            # eval mode: p = 0
    norm = norm.to(DEV)
    g = torch.Generator().manual_seed(1)
    x = (3 * torch.randn(shape, generator=g) + 1).to(DEV).requires_grad_(True)
    r = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
    assert fused_applies(x, r, norm)
    y = add_dropout_norm(x, r, drop, norm)
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    nd = torch.nn.LayerNorm(256).to(DEV).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    want = nd(xd + rd)
    torch.testing.assert_close(y.double(), want, rtol=0, atol=2e-6 * float(want.detach().abs().max()))
    go = torch.randn(shape, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    for got, ref in ((x.grad, xd.grad), (r.grad, rd.grad), (norm.weight.grad, nd.weight.grad), (norm.bias.grad, nd.bias.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=0, atol=3e-6 * float(ref.abs().max()))
    norm.weight.grad = norm.bias.grad = None


@pytest.mark.gpu
def test_dropout_mask_statistics_scaling_and_backward_consistency():
    """With x = 0, gamma = 1, beta = 0 nothing but the mask decides where z = dropout(r) is zero: read the mask off z (the
    tensor saved for the backward), check its rate and the 1/(1-p) scaling, then hold forward AND backward to torch's
    expression evaluated with exactly that mask."""
    p = 0.1
    drop, norm = _modules(p)
    norm = norm.to(DEV)
    g = torch.Generator().manual_seed(3)
    rows = 20000
    x = torch.randn(rows, 256, generator=g).to(DEV).requires_grad_(True)
    r = (torch.rand(rows, 256, generator=g) + 0.5).to(DEV).requires_grad_(True)       # never zero: zeros are drops
    seed = 0x1234567890ABCDEF
    y = add_dropout_norm(x, r, drop, norm, seed=seed)
    z = y.grad_fn.saved_tensors[0].clone()
    kept = (z - x.detach()) != 0
    assert abs(float((~kept).float().mean()) - p) < 2e-3                              # 5 M samples: sigma = 1.3e-4
    assert abs(float((~kept).float().mean(0).max()) - p) < 0.02                       # no column is special
    torch.testing.assert_close((z - x.detach())[kept], (r.detach() / (1 - p))[kept], rtol=1e-6, atol=1e-6)
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    nd = torch.nn.LayerNorm(256).to(DEV).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    want = nd(xd + rd * kept.double() / (1 - p))
    torch.testing.assert_close(y.double(), want, rtol=0, atol=2e-6 * float(want.detach().abs().max()))
    go = torch.randn(rows, 256, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    for got, ref in ((x.grad, xd.grad), (r.grad, rd.grad), (norm.weight.grad, nd.weight.grad), (norm.bias.grad, nd.bias.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=0, atol=3e-6 * float(ref.abs().max()))
    # a different seed gives a different mask, the same seed the same one
    z = z.clone()
    y2 = add_dropout_norm(x, r, drop, norm, seed=seed + 1)
    y3 = add_dropout_norm(x, r, drop, norm, seed=seed)
    assert torch.equal(y3.grad_fn.saved_tensors[0], z) and not torch.equal(y2.grad_fn.saved_tensors[0], z)
    # the parameter gradients are summed in a fixed order: bit-identical on a second run
    gw = norm.weight.grad.clone()
    norm.weight.grad = None
    x.grad = r.grad = None
    add_dropout_norm(x, r, drop, norm, seed=seed).backward(go)
    assert torch.equal(norm.weight.grad, gw)


@pytest.mark.gpu
def test_captured_training_replays_draw_fresh_masks_and_backward_follows():
    """a host seed is baked into a captured hipGraph.  Inside `step_scope` the kernels mix in a device-side
    step seed that a captured `add_` bumps on every replay: masks differ from replay to replay, and the backward of a
    replay (captured too, as make_graphed_callables does) recomputes the mask of ITS forward."""
    from vnext_amd.ops.fused_norm import step_scope
    p = 0.25
    drop, norm = _modules(p)
    norm = norm.to(DEV)
    with torch.no_grad():
        norm.weight.fill_(1.0); norm.bias.zero_()
    rows = 512
    g = torch.Generator().manual_seed(5)
    x = torch.zeros(rows, 256, device=DEV)
    r = (torch.rand(rows, 256, generator=g) + 0.5).to(DEV)
    go = torch.randn(rows, 256, generator=g).to(DEV)

    class Site(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.drop, self.norm = drop, norm

        def forward(self, xx, rr):
            with step_scope(xx.device):
                y = add_dropout_norm(xx, rr, self.drop, self.norm)
            return y

    site = Site().train()
    xs, rs = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    graphed = torch.cuda.make_graphed_callables(site, (xs, rs))
    masks, gmasks = [], []
    for _ in range(3):
        xi, ri = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        y = graphed(xi, ri)
        y.backward(go)
        torch.cuda.synchronize()
        # x = 0: z = dropout(r), so y's row statistics depend on the mask; read the mask off grad_r instead:
        # grad_r = keep * scale * dz with dz != 0 almost surely
        gmasks.append((ri.grad != 0).clone())
        # and off the forward: dropped elements have z = 0 -> y = (0 - mean) * rstd, identical within a row
        masks.append(y.detach().clone())
    for i in range(3):
        rate = 1.0 - float(gmasks[i].float().mean())
        assert abs(rate - p) < 0.02
    assert not torch.equal(gmasks[0], gmasks[1]) and not torch.equal(gmasks[1], gmasks[2])
    assert not torch.equal(masks[0], masks[1])
    # forward and backward of one replay agree on the mask: a dropped element of row i has y == min-like constant
    # c_i = -mean_i * rstd_i; recompute z's zero pattern from y and compare with the backward's pattern
    for y, gm in zip(masks, gmasks):
        kept = gm
        z = torch.where(kept, r / (1 - p), torch.zeros_like(r))
        want = torch.nn.functional.layer_norm(z, (256,))
        torch.testing.assert_close(y, want, rtol=0, atol=3e-5)


@pytest.mark.gpu
def test_a_second_scope_entry_before_the_backward_does_not_change_the_mask_under_it():
    """the backward used to read the step-seed word as it stood when the BACKWARD ran; a second `step_scope`
    entry between a forward and its backward (two trunk forwards, one backward) bumped it and the backward silently
    recomputed another mask.  Every scope entry now leaves a snapshot that its sites' forward and backward both read."""
    from vnext_amd.ops.fused_norm import step_scope
    p = 0.25
    drop, norm = _modules(p)
    norm = norm.to(DEV)
    with torch.no_grad():
        norm.weight.fill_(1.0); norm.bias.zero_()
    rows = 256
    g = torch.Generator().manual_seed(9)
    x = torch.zeros(rows, 256, device=DEV)
    r = (torch.rand(rows, 256, generator=g) + 0.5).to(DEV)
    go = torch.randn(rows, 256, generator=g).to(DEV)
    r1, r2 = r.clone().requires_grad_(True), r.clone().requires_grad_(True)
    ys = [torch.empty_like(x), torch.empty_like(x)]

    def step():
        r1.grad = r2.grad = None
        with step_scope(DEV):
            y1 = add_dropout_norm(x, r1, drop, norm)
        with step_scope(DEV):                       # bumps the step seed again before y1's backward has run
            y2 = add_dropout_norm(x, r2, drop, norm)
        ((y1 + y2) * go).sum().backward()
        ys[0].copy_(y1.detach()); ys[1].copy_(y2.detach())

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()                                   # eager warm-up: creates the step seed outside the capture
    torch.cuda.current_stream().wait_stream(s)
    r1.grad = r2.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    seen = []
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for y, rr in zip(ys, (r1, r2)):
            kept = rr.grad != 0                      # the mask the BACKWARD used
            z = torch.where(kept, r / (1 - p), torch.zeros_like(r))
            torch.testing.assert_close(y, torch.nn.functional.layer_norm(z, (256,)), rtol=0, atol=3e-5)   # = the forward's
            seen.append(kept.clone())
    assert not torch.equal(seen[0], seen[1])         # the two sites of one replay drew different masks
    assert not torch.equal(seen[0], seen[2])         # and so did the two replays


@pytest.mark.gpu
def test_capture_without_a_step_scope_falls_back_to_the_graph_safe_expression():
    p = 0.25
    drop, norm = _modules(p)
    norm = norm.to(DEV)
    x = torch.zeros(256, 256, device=DEV)
    r = torch.ones(256, 256, device=DEV)
    out = torch.empty_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            out.copy_(add_dropout_norm(x, r, drop, norm))
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = add_dropout_norm(x, r, drop, norm)       # no step_scope: nn.Dropout's Philox offset advances per replay
        assert y.grad_fn is None or "AddDropoutLayerNorm" not in type(y.grad_fn).__name__
        out.copy_(y)
    graph.replay(); torch.cuda.synchronize(); a = out.clone()
    graph.replay(); torch.cuda.synchronize(); b = out.clone()
    assert not torch.equal(a, b)
