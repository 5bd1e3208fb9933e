"""The decoder layers' self-attention over the object queries, glue fused (vnext_amd/csrc/self_attn.hip).

Reference (projects/SeqFormer/seqformer/models/deformable_transformer.py:286-323, the `_box` twin :300-321, and IDOL's decoder
layer, projects/IDOL/idol/models/deformable_transformer.py):

    q = k = self.with_pos_embed(tgt, query_pos)
    tgt2 = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1))[0].transpose(0, 1)
    tgt = tgt + self.dropout2(tgt2)
    tgt = self.norm2(tgt)

with `self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)`.  ATen runs that as: the position add, three
projections on transposed copies (each a GEMM + a bias add), a scale, a batched GEMM, softmax, dropout, a batched GEMM, a
copy back, the output projection and a mean of the attention weights over the heads that `[0]` throws away -- 18 launches
forward and about 27 backward per call, twelve calls per SeqFormer training step -- before the residual / dropout / LayerNorm
chain this package already fuses.  `query_self_attention_block` computes the same function with

    xp  = x + pos                                   (one add; pos broadcasts over the frames of the box queries)
    qkv = [xp Wqk^T | x Wv^T]                        (two library GEMMs into one [rows, 3 C] buffer, no bias)
    o   = softmax-dropout attention, all heads      ONE launch (vnx_query_self_attention_forward: bias, scale, scores,
                                                    online softmax, dropout, context; probabilities never stored)
    r   = o Wo^T                                     (library GEMM, no bias)
    y   = LayerNorm(x + dropout(r + bo))            ONE pass (fused_norm, r_bias)

and a backward of one attention launch + the GEMMs' (the in-projection's bias gradient is one column sum, the output
projection's falls out of the LayerNorm backward): 6 launches forward, 11 backward.

Same module (`nn.MultiheadAttention`: same parameters, same state dict).  The dropout masks are this package's hash masks
(fused_norm.py), not torch's Philox stream; with p = 0 (eval) the block equals the reference expression to fp32 rounding
(the online softmax sums in a different order).  Everywhere the kernels do not apply (CPU, autocast, other widths, masks,
capture without a step_scope) the block IS the reference expression, evaluated by torch.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .. import _lib
from . import fused_norm
from .fused_norm import add_dropout_norm, dropout_site

HEAD_DIM = 32
ENABLE = os.environ.get("VNX_FUSED_SELF_ATTN", "1") != "0"      # A/B switch: off = the reference expression, by torch


class _QuerySelfAttention(torch.autograd.Function):
    """x [B, Q, C] (contiguous), pos [B // t, Q, C] or None, in_proj weight [3C, C] / bias [3C] -> the attention context
    [B, Q, C] (before the output projection).  t: frames that share one pos row block (box queries: B = N * t)."""

    # Under torch.autocast the block runs in fp32 (custom_fwd casts what arrives and switches autocast off inside): x is the
    # fp32 residual stream anyway, the 300-query GEMMs are small, and one launch each way replaces the 18 + 27 of the eager
    # nn.MultiheadAttention -- which is what the block fell to under autocast until round 6.
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, pos, w_in, b_in, heads, p, seed, seed_tensor, t):
        lib = _lib.lib()
        B, Q, C = x.shape
        x2 = x.view(B * Q, C)
        if pos is None:
            xp2 = x2
        elif t == 1:
            xp2 = (x + pos).view(B * Q, C)
        else:
            xp2 = (x.view(B // t, t, Q, C) + pos.unsqueeze(1)).view(B * Q, C)
        qkv = torch.empty(B * Q, 3 * C, dtype=x.dtype, device=x.device)
        torch.mm(xp2, w_in[:2 * C].t(), out=qkv[:, :2 * C])
        torch.mm(x2, w_in[2 * C:].t(), out=qkv[:, 2 * C:])
        out = torch.empty(B, Q, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(B * heads * Q, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.vnx_query_self_attention_forward(
                _lib.VNX_F32, qkv.data_ptr(), b_in.data_ptr() if b_in is not None else None, out.data_ptr(), lse.data_ptr(),
                B, Q, heads, C // heads, 3 * C, float(p), int(seed),
                seed_tensor.data_ptr() if seed_tensor is not None else None, _lib.current_stream(x)))
        ctx.save_for_backward(x, xp2 if pos is not None else None, w_in, b_in, qkv, out, lse)
        ctx.heads, ctx.p, ctx.seed, ctx.seed_tensor, ctx.t, ctx.has_pos = heads, float(p), int(seed), seed_tensor, t, pos is not None
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        lib = _lib.lib()
        x, xp2, w_in, b_in, qkv, out, lse = ctx.saved_tensors
        if grad_out.dtype != torch.float32:
            grad_out = grad_out.float()
        B, Q, C = x.shape
        x2 = x.view(B * Q, C)
        if xp2 is None:
            xp2 = x2
        grad_out = grad_out.contiguous()
        g = torch.empty_like(qkv)
        with torch.cuda.device(x.device):
            _lib.check(lib.vnx_query_self_attention_backward(
                _lib.VNX_F32, qkv.data_ptr(), b_in.data_ptr() if b_in is not None else None, out.data_ptr(), lse.data_ptr(),
                grad_out.data_ptr(), g.data_ptr(), B, Q, ctx.heads, C // ctx.heads, 3 * C, 3 * C, ctx.p, ctx.seed,
                ctx.seed_tensor.data_ptr() if ctx.seed_tensor is not None else None, _lib.current_stream(x)))
        grad_b = g.sum(0) if b_in is not None else None
        grad_w = torch.empty_like(w_in)
        torch.mm(g[:, :2 * C].t(), xp2, out=grad_w[:2 * C])
        torch.mm(g[:, 2 * C:].t(), x2, out=grad_w[2 * C:])
        grad_xp = torch.mm(g[:, :2 * C], w_in[:2 * C])                      # d / d (x + pos)
        grad_x = torch.addmm(grad_xp, g[:, 2 * C:], w_in[2 * C:]).view(B, Q, C)
        grad_pos = None
        if ctx.has_pos:
            grad_pos = grad_xp.view(B, Q, C) if ctx.t == 1 else grad_xp.view(B // ctx.t, ctx.t, Q, C).sum(1)
        return grad_x, grad_pos, grad_w, grad_b, None, None, None, None, None


def fused_applies(x, pos, mha) -> bool:
    C = x.shape[-1]
    return (ENABLE and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32
            and isinstance(mha, torch.nn.MultiheadAttention) and mha._qkv_same_embed_dim and mha.embed_dim == C
            and mha.head_dim == HEAD_DIM and mha.in_proj_weight.dtype == torch.float32 and mha.bias_k is None
            and mha.bias_v is None and not mha.add_zero_attn and not mha.batch_first
            and (pos is None or (pos.dtype == torch.float32 and pos.dim() == 3 and pos.shape[1:] == x.shape[1:]
                                 and pos.shape[0] > 0 and x.shape[0] % pos.shape[0] == 0))
            and x.shape[0] > 0 and x.shape[1] > 0)


def query_self_attention(x, pos, mha):
    """`mha(q, k, x)[0]` for q = k = x + pos, batch-first: x [B, Q, C], pos [B // t, Q, C] (each of its row blocks serves t
    consecutive batch elements) or None; `mha` an nn.MultiheadAttention (seq-first, as the reference builds it).  Returns
    the module's output INCLUDING its output projection and bias."""
    def reference():
        if pos is None:
            qk = x
        else:
            t = x.shape[0] // pos.shape[0]
            qk = x + (pos if t == 1 else pos.unsqueeze(1).expand(-1, t, -1, -1).reshape(x.shape))
        return mha(qk.transpose(0, 1), qk.transpose(0, 1), x.transpose(0, 1))[0].transpose(0, 1)
    if not fused_applies(x, pos, mha):
        return reference()
    p, seed_tensor, ok = _site(x, mha)
    if not ok:
        return reference()
    ctxt = _QuerySelfAttention.apply(x.contiguous(), None if pos is None else pos.contiguous(), mha.in_proj_weight,
                                     mha.in_proj_bias, mha.num_heads, p, fused_norm._next_seed(), seed_tensor,
                                     1 if pos is None else x.shape[0] // pos.shape[0])
    return F.linear(ctxt, mha.out_proj.weight, mha.out_proj.bias)


class _AttnDropout:      # what dropout_site() reads of an nn.Dropout, for the probability dropout inside nn.MultiheadAttention
    def __init__(self, mha):
        self.p, self.training = float(mha.dropout), mha.training


def _site(x, mha):
    return dropout_site(x, _AttnDropout(mha))


def query_self_attention_block(x, pos, mha, dropout, norm):
    """`norm(x + dropout(mha(x + pos, x + pos, x)[0]))`, batch-first (see the module docstring): the self-attention
    sub-layer of a decoder layer.  x [B, Q, C]; pos [B // t, Q, C] or None."""
    if not fused_applies(x, pos, mha) or not fused_norm.fused_applies(x, x, norm) or mha.out_proj.bias is None:
        return add_dropout_norm(x, query_self_attention(x, pos, mha), dropout, norm)
    p, seed_tensor, ok = _site(x, mha)
    if not ok:
        return add_dropout_norm(x, query_self_attention(x, pos, mha), dropout, norm)
    x = x.contiguous()
    ctxt = _QuerySelfAttention.apply(x, None if pos is None else pos.contiguous(), mha.in_proj_weight, mha.in_proj_bias,
                                     mha.num_heads, p, fused_norm._next_seed(), seed_tensor,
                                     1 if pos is None else x.shape[0] // pos.shape[0])
    r = F.linear(ctxt, mha.out_proj.weight)                   # no bias: it is added in the LayerNorm pass
    return add_dropout_norm(x, r, dropout, norm, r_bias=mha.out_proj.bias)
