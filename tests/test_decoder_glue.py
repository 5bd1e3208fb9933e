"""refined_boxes / time_weighted_sum (vnext_amd/ops/decoder_glue.py, vnext_amd/csrc/decoder_glue.hip) against the expressions
they replace in the reference's decoder (projects/SeqFormer/seqformer/models/deformable_transformer.py:366-380 with
util/misc.py:493-497, and :305-312)."""
import pytest
import torch

from vnext_amd.ops import decoder_glue as G

DEV = "cuda:0"


def _boxes_reference(delta, ref):
    if ref.shape[-1] == 4:
        moved = delta + G.inverse_sigmoid(ref)
    else:
        moved = torch.cat([delta[..., :2] + G.inverse_sigmoid(ref), delta[..., 2:]], -1)
    return moved.sigmoid()


def test_cpu_takes_the_reference_expressions():
    d, r = torch.randn(2, 3, 7, 4), torch.rand(2, 3, 7, 2)
    assert torch.equal(G.refined_boxes(d, r), _boxes_reference(d, r))
    x, z = torch.randn(2, 5, 7, 16), torch.randn(2, 5, 7, 1)
    assert torch.equal(G.time_weighted_sum(x, z), (x * torch.softmax(z, 1)).sum(1))


@pytest.mark.gpu
@pytest.mark.parametrize("comps", [2, 4])
@pytest.mark.parametrize("shape", [(2, 5, 300), (1,), (3, 257)])
def test_refined_boxes_forward_and_both_gradients(comps, shape):
    g = torch.Generator().manual_seed(comps * 10 + len(shape))
    delta = (2 * torch.randn(*shape, 4, generator=g)).to(DEV).requires_grad_(True)
    ref = torch.rand(*shape, comps, generator=g)
    flat = ref.view(-1)
    # the clamps' corners: outside [0, 1], on its ends, around eps and 1 - eps
    edge = torch.tensor([-0.3, 0.0, 1.0, 1.2, 1e-5, 0.5e-5, 2e-5, 1 - 1e-5, 1 - 0.5e-5, 1 - 2e-5, 1e-7, 0.5])
    flat[:min(len(edge), flat.numel())] = edge[:flat.numel()]
    ref = ref.to(DEV).requires_grad_(True)
    y = G.refined_boxes(delta, ref)
    dd, rd = delta.detach().double().requires_grad_(True), ref.detach().double().requires_grad_(True)
    want = _boxes_reference(dd, rd)
    torch.testing.assert_close(y.double(), want, rtol=0, atol=3e-7)
    go = torch.randn(*shape, 4, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    torch.testing.assert_close(delta.grad.double(), dd.grad, rtol=0, atol=1e-6)
    # grad_reference = grad_delta x the logit's slope, which reaches 1e5 at the clamps, where fp32 knows y (1 - y) of a saturated
    # sigmoid to a few per cent only: hold the difference against the slope (and skip the elements fp32 puts on the other
    # side of a clamp's corner than fp64 does)
    x64 = rd.detach()
    inside = (x64 >= 0) & (x64 <= 1)
    slope = torch.where(inside & (x64 >= 1e-5), 1 / x64.clamp(min=1e-30), torch.zeros_like(x64)) \
        + torch.where(inside & (1 - x64 >= 1e-5), 1 / (1 - x64).clamp(min=1e-30), torch.zeros_like(x64))
    corner = ((x64 - 1e-5).abs() < 1e-11) | ((1 - x64 - 1e-5).abs() < 2e-7)
    err = (ref.grad.double() - rd.grad).abs()
    assert bool((err[~corner] <= 2e-6 * (1 + slope[~corner])).all()), float((err[~corner] / (1 + slope[~corner])).max())
    assert torch.equal(ref.grad[~inside.to(DEV)], torch.zeros_like(ref.grad[~inside.to(DEV)]))


@pytest.mark.gpu
def test_refined_boxes_without_a_reference_gradient_skips_it():
    delta = torch.randn(10, 300, 4, device=DEV, requires_grad=True)
    ref = torch.rand(10, 300, 4, device=DEV)
    y = G.refined_boxes(delta, ref)
    y.sum().backward()
    want = _boxes_reference(delta.detach().double(), ref.double())
    torch.testing.assert_close(delta.grad.double(), want * (1 - want), rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,T,Q,C", [(2, 5, 300, 256), (1, 1, 7, 256), (3, 16, 5, 64), (2, 3, 33, 8), (1, 5, 300, 512)])
def test_time_weighted_sum_forward_and_gradients(N, T, Q, C):
    g = torch.Generator().manual_seed(N * 100 + T)
    x = torch.randn(N, T, Q, C, generator=g).to(DEV).requires_grad_(True)
    z = (3 * torch.randn(N, T, Q, 1, generator=g)).to(DEV).requires_grad_(True)
    y = G.time_weighted_sum(x, z)
    xd, zd = x.detach().double().requires_grad_(True), z.detach().double().requires_grad_(True)
    want = (xd * torch.softmax(zd, 1)).sum(1)
    torch.testing.assert_close(y.double(), want, rtol=0, atol=2e-6 * float(want.detach().abs().max()))
    go = torch.randn(N, Q, C, generator=g).to(DEV)
    y.backward(go)
    want.backward(go.double())
    torch.testing.assert_close(x.grad.double(), xd.grad, rtol=0, atol=2e-6 * float(xd.grad.abs().max()))
    torch.testing.assert_close(z.grad.double(), zd.grad, rtol=0, atol=5e-6 * float(zd.grad.abs().max()) + 1e-7)
    assert z.grad.shape == z.shape


@pytest.mark.gpu
def test_bad_arguments_are_reported():
    from vnext_amd import _lib
    lib = _lib.lib()
    t = torch.zeros(64, device=DEV)
    s = _lib.current_stream(t)
    assert lib.vnx_refine_boxes_forward(_lib.VNX_F32, t.data_ptr(), t.data_ptr(), t.data_ptr(), 4, 3, 1e-5, s) == 1
    assert lib.vnx_refine_boxes_forward(_lib.VNX_F32, None, t.data_ptr(), t.data_ptr(), 4, 4, 1e-5, s) == 1
    assert lib.vnx_time_weighted_sum_forward(_lib.VNX_F32, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 17, 1, 4, s) == 1
    assert lib.vnx_time_weighted_sum_forward(_lib.VNX_F32, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 2, 1, 6, s) == 1
