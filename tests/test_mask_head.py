"""Dynamic mask head: the numpy oracle against outputs of the reference's own
function bodies (CPU), and the fused HIP kernel against both (GPU)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import heads_oracle as H

CASES = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "heads_mask_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"heads_mask_{name}.npz")))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_functions(name):
    g = load(name)
    out = H.dynamic_mask_head(g["feats"], g["ref"], g["params"], list(g["num_insts"]))
    np.testing.assert_allclose(out, g["out"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", CASES)
def test_oracle_backward_matches_reference_autograd(name):
    g = load(name)
    gf, gr, gp = H.dynamic_mask_head_backward(g["feats"], g["ref"], g["params"], list(g["num_insts"]), g["grad_out"])
    np.testing.assert_allclose(gf, g["grad_feats"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gp, g["grad_params"], rtol=1e-11, atol=1e-9)
    # the reference's `.float()` on the relative coordinates rounds this one gradient to fp32
    np.testing.assert_allclose(gr, g["grad_ref"], rtol=1e-6, atol=1e-6)


def test_upsampling_adjoint_is_the_transpose():
    rng = np.random.default_rng(0)
    x, g = rng.standard_normal((3, 4, 7)), rng.standard_normal((3, 8, 14))
    np.testing.assert_allclose((H.aligned_bilinear_x2(x) * g).sum(), (x * H.aligned_bilinear_x2_adjoint(g)).sum(),
                               rtol=1e-12)


def test_upsampling_closed_form_is_the_pad_interpolate_crop_chain():
    x = torch.randn(2, 1, 5, 9, dtype=torch.float64)
    t = torch.nn.functional.pad(x, (0, 1, 0, 1), mode="replicate")
    t = torch.nn.functional.interpolate(t, size=(11, 19), mode="bilinear", align_corners=True)
    t = torch.nn.functional.pad(t, (1, 0, 1, 0), mode="replicate")[:, :, :10, :18]
    np.testing.assert_allclose(H.aligned_bilinear_x2(x.numpy()), t.numpy(), rtol=1e-13, atol=1e-13)


def test_cpu_tensors_are_rejected():
    from vnext_amd.heads import dynamic_mask_with_coords
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        dynamic_mask_with_coords(torch.zeros(1, 8, 2, 2), torch.zeros(1, 1, 2), torch.zeros(1, 1, 169), [1], 8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernel_matches_reference_outputs(name):
    from vnext_amd.heads import dynamic_mask_with_coords
    g = load(name)
    dev = "cuda:0"
    with torch.no_grad():
        out = dynamic_mask_with_coords(torch.from_numpy(g["feats"]).float().to(dev),
                                       torch.from_numpy(g["ref"]).float().to(dev)[None],
                                       torch.from_numpy(g["params"]).float().to(dev)[None],
                                       [int(n) for n in g["num_insts"]], 8)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1,) + g["out"].shape
    # fp32 inputs: compare with the oracle run on the same fp32-rounded inputs
    ref = H.dynamic_mask_head(g["feats"].astype(np.float32).astype(np.float64),
                              g["ref"].astype(np.float32).astype(np.float64),
                              g["params"].astype(np.float32).astype(np.float64), list(g["num_insts"]))
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(out[0].double().cpu().numpy(), ref, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(out[0].double().cpu().numpy(), g["out"], rtol=0, atol=1e-4 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("H_,W_,n", [(48, 80, 300), (92, 160, 37), (1, 130, 3), (17, 1, 2), (33, 47, 5), (7, 95, 4), (5, 31, 3), (6, 64, 3),
                                     # the runs kernel's edges: rows wider than a chunk (the kept logits overlap their new
                                     # place), many rows per register, one row per run (n = 1), one run per instance (n = 2 100)
                                     (3, 400, 2), (2, 500, 3), (300, 3, 2), (48, 80, 1), (48, 80, 2100)])
def test_kernel_at_frame_sizes(H_, W_, n):
    """360p / 720p frame sizes (BASELINE configs): a sample of instances against the oracle,
    and linearity in the last layer's bias (adds a constant to every logit)."""
    from vnext_amd.heads import dynamic_mask_with_coords
    gen = torch.Generator().manual_seed(H_ * 1000 + W_)
    feats = torch.randn(2, 8, H_, W_, generator=gen)
    counts = [n, max(n // 2, 1)]
    n_all = sum(counts)
    ref = torch.rand(1, n_all, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])
    params = 0.3 * torch.randn(1, n_all, 169, generator=gen)
    with torch.no_grad():
        out = dynamic_mask_with_coords(feats.cuda(), ref.cuda(), params.cuda(), counts, 8)
        p2 = params.clone()
        p2[..., 168] += 1.5
        out2 = dynamic_mask_with_coords(feats.cuda(), ref.cuda(), p2.cuda(), counts, 8)
    assert out.shape == (1, n_all, 2 * H_, 2 * W_)
    assert float((out2 - out - 1.5).abs().max()) <= 1e-4 * max(1.0, float(out.abs().max()))
    pick = sorted(set([0, n - 1, n, n_all - 1]))
    img_of = [0] * counts[0] + [1] * counts[1]
    for j in pick:
        one = H.dynamic_mask_head(feats[img_of[j]:img_of[j] + 1].double().numpy(), ref[0, j:j + 1].double().numpy(),
                                  params[0, j:j + 1].double().numpy(), [1])
        scale = max(1e-6, float(np.abs(one).max()))
        np.testing.assert_allclose(out[0, j].double().cpu().numpy(), one[0], rtol=0, atol=2e-5 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("H_,W_,n", [(48, 80, 40), (92, 160, 9), (5, 31, 3), (3, 400, 2), (300, 3, 2), (11, 64, 700)])
def test_single_frame_call_every_instance_and_forced_run_counts(H_, W_, n):
    """One frame per call (the inference path: the kernel is told so and skips the instance -> frame lookup): every
    instance against the oracle; and the development build's forced configurations -- one run per instance, several, one
    row per run, the strip kernel of rounds 1 - 3 -- give the same bits (the arithmetic of a pixel is the same sequence
    of operations however the frame is cut)."""
    from vnext_amd import _lib
    from vnext_amd.heads import dynamic_mask_with_coords
    gen = torch.Generator().manual_seed(H_ * 31 + W_)
    feats = torch.randn(1, 8, H_, W_, generator=gen)
    ref = torch.rand(1, n, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])
    params = 0.3 * torch.randn(1, n, 169, generator=gen)
    with torch.no_grad():
        out = dynamic_mask_with_coords(feats.cuda(), ref.cuda(), params.cuda(), [n], 8)
    torch.cuda.synchronize()
    for j in range(0, n, max(1, n // 12)):
        one = H.dynamic_mask_head(feats.double().numpy(), ref[0, j:j + 1].double().numpy(), params[0, j:j + 1].double().numpy(), [1])
        scale = max(1e-6, float(np.abs(one).max()))
        np.testing.assert_allclose(out[0, j].double().cpu().numpy(), one[0], rtol=0, atol=2e-5 * scale)
    try:
        for v in (701, 702, 703, 707, 798, 799):
            _lib.set_kernel_variant(v)
            with torch.no_grad():
                other = dynamic_mask_with_coords(feats.cuda(), ref.cuda(), params.cuda(), [n], 8)
            assert torch.equal(other, out), v
    finally:
        _lib.set_kernel_variant(0)


def _grads_on_gpu(feats, ref, params, counts, gout):
    from vnext_amd.heads import dynamic_mask_with_coords
    dev = "cuda:0"
    f = feats.float().to(dev).requires_grad_(True)
    r = ref.float().to(dev)[None].requires_grad_(True)
    p = params.float().to(dev)[None].requires_grad_(True)
    out = dynamic_mask_with_coords(f, r, p, counts, 8)
    gf, gr, gp = torch.autograd.grad(out, (f, r, p), gout.float().to(dev)[None])
    torch.cuda.synchronize()
    return gf.double().cpu().numpy(), gr[0].double().cpu().numpy(), gp[0].double().cpu().numpy()


def _close(got, want, tol):
    scale = max(1e-9, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_backward_kernel_matches_reference_autograd(name):
    g = load(name)
    counts = [int(n) for n in g["num_insts"]]
    gf, gr, gp = _grads_on_gpu(torch.from_numpy(g["feats"]), torch.from_numpy(g["ref"]),
                               torch.from_numpy(g["params"]), counts, torch.from_numpy(g["grad_out"]))
    # against the oracle on the same fp32-rounded inputs (tight), then the reference's fp64 run
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    of, orf, op = H.dynamic_mask_head_backward(f32(g["feats"]), f32(g["ref"]), f32(g["params"]), counts, f32(g["grad_out"]))
    for got, want, ref64 in ((gf, of, g["grad_feats"]), (gr, orf, g["grad_ref"]), (gp, op, g["grad_params"])):
        _close(got, want, 2e-5)
        _close(got, ref64, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("H_,W_,counts", [(48, 80, [7, 0, 12]), (92, 160, [3, 5]), (1, 130, [2]), (17, 1, [2, 1]),
                                          (9, 63, [1]), (8, 64, [2])])
def test_backward_kernel_at_frame_sizes(H_, W_, counts):
    """Training-like instance counts at the 360p / 720p mask-feature sizes plus strip-edge shapes,
    against the fp64 oracle; and linearity in grad_out."""
    gen = torch.Generator().manual_seed(H_ * 977 + W_)
    n_all = sum(counts)
    feats = torch.randn(len(counts), 8, H_, W_, generator=gen)
    ref = torch.rand(n_all, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])
    params = 0.3 * torch.randn(n_all, 169, generator=gen)
    gout = torch.randn(n_all, 2 * H_, 2 * W_, generator=gen)
    gf, gr, gp = _grads_on_gpu(feats, ref, params, counts, gout)
    of, orf, op = H.dynamic_mask_head_backward(feats.double().numpy(), ref.double().numpy(), params.double().numpy(),
                                               counts, gout.double().numpy())
    _close(gf, of, 2e-5)
    _close(gr, orf, 2e-5)
    _close(gp, op, 2e-5)
    gf2, gr2, gp2 = _grads_on_gpu(feats, ref, params, counts, -2.0 * gout)
    _close(gf2, -2.0 * gf, 1e-5)
    _close(gp2, -2.0 * gp, 1e-5)
    assert gf.shape == feats.shape and not np.any(gf[1]) if counts[1:2] == [0] else True


@pytest.mark.gpu
def test_no_instances_gives_empty_output_and_zero_feature_gradient():
    from vnext_amd.heads import dynamic_mask_with_coords
    f = torch.randn(1, 8, 4, 6, device="cuda:0", requires_grad=True)
    out = dynamic_mask_with_coords(f, torch.zeros(1, 0, 2, device="cuda:0"), torch.zeros(1, 0, 169, device="cuda:0"), [0], 8)
    assert out.shape == (1, 0, 8, 12)


@pytest.mark.gpu
def test_backward_at_the_training_shape_of_a_360p_clip():
    """forward_mask_head_train at BASELINE config 2: T = 5 frames of 48x80 features, the matched
    instances of 6 decoder layers in one launch (here 6 x 5 x 4 = 120 instances in arbitrary
    image order through the flat-instance surface) -- every gradient against the fp64 oracle."""
    from vnext_amd.heads import dynamic_mask_head
    gen = torch.Generator().manual_seed(123)
    T_, H_, W_, n = 5, 48, 80, 120
    feats = torch.randn(T_, 8, H_, W_, generator=gen)
    pts = torch.rand(n, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])
    params = 0.3 * torch.randn(n, 169, generator=gen)
    image = torch.randint(0, T_, (n,), generator=gen, dtype=torch.int32)
    gout = torch.randn(n, 2 * H_, 2 * W_, generator=gen)
    f = feats.cuda().requires_grad_(True); p = pts.cuda().requires_grad_(True); w = params.cuda().requires_grad_(True)
    out = dynamic_mask_head(f, p, w, image.cuda(), 8)
    out.backward(gout.cuda())
    # oracle: instances grouped by image (its layout), then scattered back
    order = torch.argsort(image, stable=True)
    counts = torch.bincount(image, minlength=T_).tolist()
    of, orf, op = H.dynamic_mask_head_backward(feats.double().numpy(), pts[order].double().numpy(),
                                               params[order].double().numpy(), counts, gout[order].double().numpy())
    _close(f.grad.double().cpu().numpy(), of, 3e-5)
    _close(p.grad[order].double().cpu().numpy(), orf, 3e-5)
    _close(w.grad[order].double().cpu().numpy(), op, 3e-5)
    one = H.dynamic_mask_head(feats.double().numpy(), pts[order].double().numpy(), params[order].double().numpy(), counts)
    _close(out[order].double().detach().cpu().numpy(), one, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [1, 2, 3])
def test_backward_accepts_a_4_byte_aligned_grad_feats_pointer(shift):
    """the zero-fill launch of the backward wrote grad_feats with 16-byte stores on the strength of "it comes
    from the allocator".  The C ABI takes any caller's pointer: a view with a storage offset is only 4-byte aligned.
    Through the C ABI with grad_feats `shift` floats into a buffer: same gradients, nothing written outside the view."""
    from vnext_amd import _lib
    gen = torch.Generator().manual_seed(31 + shift)
    counts, H_, W_ = [3, 2], 9, 21
    n_all = sum(counts)
    feats = torch.randn(len(counts), 8, H_, W_, generator=gen).cuda()
    ref = (torch.rand(n_all, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])).cuda()
    params = (0.3 * torch.randn(n_all, 169, generator=gen)).cuda()
    gout = torch.randn(n_all, 2 * H_, 2 * W_, generator=gen).cuda()
    image = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)).cuda()
    guard = 7.0
    buf = torch.full((feats.numel() + 8,), guard, device="cuda:0")
    gfeats = buf[shift:shift + feats.numel()]
    assert gfeats.data_ptr() % 16 == 4 * shift
    gref, gparams = torch.empty_like(ref), torch.empty_like(params)
    _lib.check(_lib.lib().vnx_dynamic_mask_head_backward(
        _lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), image.data_ptr(), gout.data_ptr(),
        gfeats.data_ptr(), gref.data_ptr(), gparams.data_ptr(), len(counts), 8, H_, W_, n_all, 169, 8,
        torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    of, orf, op = H.dynamic_mask_head_backward(feats.double().cpu().numpy(), ref.double().cpu().numpy(),
                                               params.double().cpu().numpy(), counts, gout.double().cpu().numpy())
    _close(gfeats.double().cpu().numpy().reshape(of.shape), of, 2e-5)
    _close(gparams.double().cpu().numpy(), op, 2e-5)
    assert bool((buf[:shift] == guard).all()) and bool((buf[shift + feats.numel():] == guard).all())


@pytest.mark.gpu
def test_training_pair_zero_fills_in_the_forward_and_a_second_backward_still_works():
    """ABI 15: `vnx_dynamic_mask_head_forward_train` zero-fills the backward's three gradient buffers in the forward's own launch
    and `vnx_dynamic_mask_head_backward_zeroed` accumulates into them without a zero-fill of its own.  Through the C ABI on
    buffers filled with garbage, at a misaligned grad_feats pointer: the same output as the plain forward, the oracle's
    gradients, nothing written outside the views.  Through autograd: the buffers serve ONE backward -- a second backward over the
    retained graph takes the self-zeroing entry point and gives the same gradients (the mask head's sums over instances of a frame
    are atomic: equal to the oracle's tolerance, not bit for bit)."""
    from vnext_amd import _lib
    from vnext_amd.heads import dynamic_mask_head
    gen = torch.Generator().manual_seed(77)
    counts, H_, W_ = [4, 0, 3], 13, 37
    n_all = sum(counts)
    feats = torch.randn(len(counts), 8, H_, W_, generator=gen).cuda()
    ref = (torch.rand(n_all, 2, generator=gen) * torch.tensor([W_ * 8.0, H_ * 8.0])).cuda()
    params = (0.3 * torch.randn(n_all, 169, generator=gen)).cuda()
    gout = torch.randn(n_all, 2 * H_, 2 * W_, generator=gen).cuda()
    image = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    guard = 7.0
    buf = torch.full((feats.numel() + 8,), guard, device="cuda:0")
    gfeats = buf[1:1 + feats.numel()]
    gref, gparams = torch.full_like(ref, guard), torch.full_like(params, guard)
    out_plain, out_train = torch.empty(n_all, 2 * H_, 2 * W_, device="cuda:0"), torch.empty(n_all, 2 * H_, 2 * W_, device="cuda:0")
    dims = (len(counts), 8, H_, W_, n_all, 169, 8, stream)
    _lib.check(_lib.lib().vnx_dynamic_mask_head_forward(_lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(),
                                                        image.data_ptr(), out_plain.data_ptr(), *dims))
    _lib.check(_lib.lib().vnx_dynamic_mask_head_forward_train(_lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(),
                                                              image.data_ptr(), out_train.data_ptr(), gfeats.data_ptr(),
                                                              gref.data_ptr(), gparams.data_ptr(), *dims))
    torch.cuda.synchronize()
    assert torch.equal(out_plain, out_train)
    assert not gfeats.any() and not gref.any() and not gparams.any()
    assert float(buf[0]) == guard and bool((buf[1 + feats.numel():] == guard).all())
    _lib.check(_lib.lib().vnx_dynamic_mask_head_backward_zeroed(
        _lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), image.data_ptr(), gout.data_ptr(),
        gfeats.data_ptr(), gref.data_ptr(), gparams.data_ptr(), *dims))
    torch.cuda.synchronize()
    of, orf, op = H.dynamic_mask_head_backward(feats.double().cpu().numpy(), ref.double().cpu().numpy(),
                                               params.double().cpu().numpy(), counts, gout.double().cpu().numpy())
    _close(gfeats.double().cpu().numpy().reshape(of.shape), of, 2e-5)
    _close(gref.double().cpu().numpy(), orf, 2e-5)
    _close(gparams.double().cpu().numpy(), op, 2e-5)
    assert float(buf[0]) == guard and bool((buf[1 + feats.numel():] == guard).all())
    # no instances at all: the training forward still leaves zeros (its own zero-fill launch)
    g0 = torch.full((2, 8, 3, 5), guard, device="cuda:0")
    _lib.check(_lib.lib().vnx_dynamic_mask_head_forward_train(_lib.VNX_F32, g0.data_ptr(), None, None, None, None, g0.data_ptr(), None,
                                                              None, 2, 8, 3, 5, 0, 169, 8, stream))
    torch.cuda.synchronize()
    assert not g0.any()
    # autograd: first backward = the forward-zeroed buffers, second (retained graph) = fresh buffers, self-zeroing entry point
    f, p, w = feats.clone().requires_grad_(True), ref.clone().requires_grad_(True), params.clone().requires_grad_(True)
    out = dynamic_mask_head(f, p, w, image, 8)
    first = torch.autograd.grad(out, (f, p, w), gout, retain_graph=True)
    second = torch.autograd.grad(out, (f, p, w), gout)
    for a, b, want in zip(first, second, (of, orf, op)):
        _close(a.double().cpu().numpy().reshape(want.shape), want, 2e-5)
        _close(b.double().cpu().numpy().reshape(want.shape), want, 2e-5)
    with torch.no_grad():      # no backward will follow: the plain forward, no buffers
        assert torch.equal(dynamic_mask_head(feats, ref, params, image, 8), out_plain)
