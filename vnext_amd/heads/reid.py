"""IDOL re-identification head on the HIP kernels.

* `similarity(a, b, normalize)`: S = a . b^T ([n,C] x [k,C]) on the matrix cores (exact fp32),
  optionally with both operands L2-normalised in the kernel; differentiable (the backward is the
  same kernel on transposed operands).  Replaces `torch.mm(embeds, memo_embeds.t())`
  (projects/IDOL/idol/models/tracker.py:229-244) and the per-instance
  `einsum('nc,kc->nk')` launches of projects/IDOL/idol/models/pos_neg_select.py:47,58-62.
* `match_scores(embeds, memo_embeds, metric)`: the tracker's association scores for
  metric in {'bisoftmax', 'softmax', 'cosine', 'longrang'} (tracker.py:228-244).
* `loss_reid(...)`: the contrastive + auxiliary cosine losses of
  projects/IDOL/idol/models/deformable_detr.py:418-454, batched: one similarity launch per image
  instead of one per instance, and the pair-wise logsumexp in closed form,
  log(1 + sum_neg e^{s_n} * sum_pos e^{-s_p}), which is the same number.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _launch_similarity(a, b, normalize):
    n, C = a.shape
    k = b.shape[0]
    out = torch.empty((n, k), dtype=torch.float32, device=a.device)
    if n == 0 or k == 0:
        return out
    with torch.cuda.device(a.device):
        st = _lib.lib().vnx_reid_similarity(
            _lib.VNX_F32, a.data_ptr(), b.data_ptr(), out.data_ptr(), n, k, C, a.stride(0), b.stride(0),
            k, 1 if normalize else 0, torch.cuda.current_stream(a.device).cuda_stream)
    _lib.check(st)
    return out


def _prep(x):
    if not x.is_cuda:
        raise RuntimeError("reid similarity: no CPU implementation (HIP library only)")
    if x.dim() != 2:
        raise RuntimeError("reid similarity: expected [rows, channels] matrices")
    x = x.float()
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    if x.stride(0) % 4 != 0:  # odd channel count: pad the rows to a multiple of 4 floats
        pad = (-x.shape[1]) % 4
        x = torch.nn.functional.pad(x, (0, pad))[:, :x.shape[1]]
    return x


class _DotSimilarity(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _prep(a), _prep(b)
        ctx.save_for_backward(a, b)
        return _launch_similarity(a, b, False)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        a, b = ctx.saved_tensors
        grad = grad.contiguous().float()
        ga = gb = None
        if ctx.needs_input_grad[0]:   # dS/da = G . b   = NT(G [n,k], b^T [C,k])
            ga = _launch_similarity(_prep(grad), _prep(b.t().contiguous()), False)
        if ctx.needs_input_grad[1]:   # dS/db = G^T . a = NT(G^T [k,n], a^T [C,n])
            gb = _launch_similarity(_prep(grad.t().contiguous()), _prep(a.t().contiguous()), False)
        return ga, gb


def similarity(a, b, normalize: bool = False):
    """[n,C] x [k,C] -> [n,k];  normalize=True gives the cosine similarity."""
    if normalize and torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        # differentiable path: normalise with torch (elementwise), contract with the kernel
        return _DotSimilarity.apply(torch.nn.functional.normalize(a.float(), dim=1),
                                    torch.nn.functional.normalize(b.float(), dim=1))
    if normalize:
        return _launch_similarity(_prep(a), _prep(b), True)
    return _DotSimilarity.apply(a, b)


def bisoftmax(sim):
    sim = sim.float().contiguous()
    n, k = sim.shape
    out = torch.empty_like(sim)
    if n == 0 or k == 0:
        return out
    with torch.cuda.device(sim.device):
        st = _lib.lib().vnx_reid_bisoftmax(_lib.VNX_F32, sim.data_ptr(), out.data_ptr(), n, k, k, k,
                                           torch.cuda.current_stream(sim.device).cuda_stream)
    _lib.check(st)
    return out


def match_scores(embeds, memo_embeds, metric: str = "bisoftmax"):
    """Association scores of `IDOL_Tracker.match` (tracker.py:228-244); inference only."""
    with torch.no_grad():
        if metric == "cosine":
            return similarity(embeds, memo_embeds, normalize=True)
        feats = similarity(embeds, memo_embeds)
        if metric == "longrang":
            return feats
        if metric == "bisoftmax":
            return bisoftmax(feats)
        if metric == "softmax":
            return feats.softmax(dim=1)
    raise NotImplementedError(metric)


def loss_reid(ref_embeds, key_embeds, pos_mask, neg_mask, aux_mask):
    """One image's contribution to the two reid losses (deformable_detr.py:427-448), summed over
    its instances.

    ref_embeds [R, C]   embeddings of the reference frame's queries
    key_embeds [I, C]   embedding of the key-frame query matched to each valid instance
    pos_mask, neg_mask [R, I] bool   contrastive positives / negatives per instance
    aux_mask [R, I] bool             the samples kept for the cosine loss (all positives + the
                                     negatives the reference draws with random.sample, :49-56)
    -> (sum_i log(1 + sum_{n,p} exp(s_n - s_p)),  sum_i mean_{aux_i} (cos - label)^2)
    """
    dot = similarity(ref_embeds, key_embeds)                    # [R, I]   ("contrast", :47)
    cos = similarity(ref_embeds, key_embeds, normalize=True)    # [R, I]   ("aux_consin", :58-62)
    neg_inf = torch.finfo(dot.dtype).min
    # log sum_n e^{s_n} and log sum_p e^{-s_p}; an empty set gives -inf -> the pair term vanishes
    lse_neg = torch.logsumexp(dot.masked_fill(~neg_mask, float("-inf")), dim=0)
    lse_pos = torch.logsumexp((-dot).masked_fill(~pos_mask, float("-inf")), dim=0)
    pair = (lse_neg + lse_pos).clamp_min(neg_inf)
    contrast = torch.nn.functional.softplus(pair)               # log(1 + e^{pair})
    label = pos_mask.to(cos.dtype)
    cnt = aux_mask.sum(0).clamp_min(1)
    aux = (((cos - label) ** 2) * aux_mask).sum(0) / cnt
    return contrast.sum(), aux.sum()
