"""IDOL reid head: numpy oracle vs fixtures produced by the reference's own code
(CPU); HIP similarity / bi-softmax / batched loss vs the same fixtures (GPU)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import heads_oracle as H

MATCH = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "reid_match_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


@pytest.mark.parametrize("name", MATCH)
def test_oracle_matches_tracker_expressions(name):
    g = load(f"reid_match_{name}")
    np.testing.assert_allclose(H.similarity(g["embeds"], g["memo"]), g["longrang"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(H.similarity(g["embeds"], g["memo"], cosine=True), g["cosine"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(H.bisoftmax(g["longrang"]), g["bisoftmax"], rtol=1e-12, atol=1e-14)


def test_closed_form_of_the_pairwise_logsumexp_matches_reference_loss():
    """log(1 + sum_{n,p} exp(s_n - s_p)) == log(1 + sum_n e^{s_n} * sum_p e^{-s_p}), per instance."""
    g = load("reid_loss")
    dot = H.similarity(g["ref"], g["key"])
    cos = H.similarity(g["ref"], g["key"], cosine=True)
    total, aux = 0.0, 0.0
    for i in range(dot.shape[1]):
        sn, sp = dot[g["neg"][:, i], i], dot[g["pos"][:, i], i]
        total += np.log1p(np.exp(sn).sum() * np.exp(-sp).sum())
        sel = g["aux"][:, i]
        aux += ((cos[sel, i] - g["pos"][sel, i].astype(float)) ** 2).mean()
    n = int(g["n_items"])
    np.testing.assert_allclose(total / n, g["loss_reid"], rtol=1e-12)
    np.testing.assert_allclose(aux / n, g["loss_reid_aux"], rtol=1e-12)


def test_cpu_tensors_are_rejected():
    from vnext_amd.heads import similarity
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        similarity(torch.zeros(2, 8), torch.zeros(3, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", MATCH)
def test_match_scores_on_gpu(name):
    from vnext_amd.heads import match_scores
    g = load(f"reid_match_{name}")
    e, m = torch.from_numpy(g["embeds"]).float().cuda(), torch.from_numpy(g["memo"]).float().cuda()
    ref_dot = H.similarity(e.cpu().numpy(), m.cpu().numpy())
    for metric in ("longrang", "cosine", "bisoftmax", "softmax"):
        got = match_scores(e, m, metric).double().cpu().numpy()
        want = g[metric]
        scale = float(np.abs(want).max())
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * max(scale, 1e-3), err_msg=metric)
    got = match_scores(e, m, "longrang").double().cpu().numpy()
    np.testing.assert_allclose(got, ref_dot, rtol=0, atol=1e-6 * float(np.abs(ref_dot).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,C", [(1, 1, 256), (16, 16, 256), (300, 300, 256), (33, 47, 64), (5, 9, 37), (20, 3, 10)])
def test_similarity_shapes_and_transpose(n, k, C):
    """Non-symmetric random inputs (a transposed tile would show), ragged tiles, odd channel counts."""
    from vnext_amd.heads import similarity
    gen = torch.Generator().manual_seed(n * 1000 + k)
    a, b = torch.randn(n, C, generator=gen), torch.randn(k, C, generator=gen)
    for normalize in (False, True):
        got = similarity(a.cuda(), b.cuda(), normalize).double().cpu().numpy()
        want = H.similarity(a.numpy(), b.numpy(), cosine=normalize)
        assert got.shape == (n, k)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * float(np.abs(want).max()))


@pytest.mark.gpu
def test_similarity_gradients():
    from vnext_amd.heads import similarity
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(37, 256, generator=gen).cuda().requires_grad_(True)
    b = torch.randn(21, 256, generator=gen).cuda().requires_grad_(True)
    w = torch.randn(37, 21, generator=gen).cuda()
    for normalize in (False, True):
        a.grad = b.grad = None
        (similarity(a, b, normalize) * w).sum().backward()
        a2 = a.detach().double().cpu().requires_grad_(True)
        b2 = b.detach().double().cpu().requires_grad_(True)
        x, y = (torch.nn.functional.normalize(a2, dim=1), torch.nn.functional.normalize(b2, dim=1)) if normalize else (a2, b2)
        ((x @ y.t()) * w.double().cpu()).sum().backward()
        for got, want in ((a.grad, a2.grad), (b.grad, b2.grad)):
            np.testing.assert_allclose(got.double().cpu().numpy(), want.numpy(), rtol=0,
                                       atol=1e-5 * float(want.abs().max()))


@pytest.mark.gpu
def test_batched_loss_matches_reference_loss_reid():
    from vnext_amd.heads import loss_reid
    g = load("reid_loss")
    t = lambda k, dt=torch.float32: torch.from_numpy(g[k]).to(dt).cuda()  # noqa: E731
    ref, key = t("ref").requires_grad_(True), t("key").requires_grad_(True)
    contrast, aux = loss_reid(ref, key, t("pos", torch.bool), t("neg", torch.bool), t("aux", torch.bool))
    n = int(g["n_items"])
    np.testing.assert_allclose(float(contrast) / n, g["loss_reid"], rtol=2e-5)
    np.testing.assert_allclose(float(aux) / n, g["loss_reid_aux"], rtol=2e-5)
    (contrast + aux).backward()
    assert torch.isfinite(ref.grad).all() and torch.isfinite(key.grad).all() and float(ref.grad.abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1, 1), (50, 50), (7, 300), (110, 111), (111, 111), (300, 300),
                                 (6, 2048), (5, 2048), (3, 4096), (2, 4096), (2048, 6)])
def test_bisoftmax_staged_and_global_forms(n, k):
    """vnx_reid_bisoftmax (tracker.py:232-235: the mean of the row softmax and the column softmax) on both sides of the
    12 288-element limit below which the matrix is staged in LDS, against float64 softmaxes.  The few-rows x full-memory-
    bank shapes (k = DeviceTracker capacity) pass the element limit but would need more than 64 KiB of dynamic LDS in the
    staged form (6 x 2 048: 65 584 B): they must take the global-memory form instead of failing to launch."""
    from vnext_amd.heads.reid import bisoftmax
    g = torch.Generator().manual_seed(n * 1000 + k)
    s = 3.0 * torch.randn(n, k, generator=g)
    got = bisoftmax(s.cuda()).double().cpu()
    want = 0.5 * (s.double().softmax(1) + s.double().softmax(0))
    torch.testing.assert_close(got, want, rtol=0, atol=2e-6)
