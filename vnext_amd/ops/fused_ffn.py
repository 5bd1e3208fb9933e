"""The feed-forward block of a deformable-transformer layer with its glue fused (vnext_amd/csrc/ffn_act.hip, add_norm.hip).

Reference (projects/SeqFormer/seqformer/models/deformable_transformer.py:226-236 encoder layer, :330-345 decoder layer and
its `_box` twin; IDOL's layers are the same):

    src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))
    src = src + self.dropout3(src2)
    src = self.norm2(src)

ATen runs that as GEMM(+bias), relu, dropout (+ byte mask), GEMM(+bias) -- and in the backward masked_scale,
threshold_backward and one column-sum launch per bias -- around the residual / dropout / LayerNorm chain this package
already fuses (fused_norm.py).  `ffn_block` computes the same function with

    h = x W1^T                      (library GEMM, no bias)
    a = dropout(relu(h + b1))       ONE in-place pass (vnx_bias_relu_dropout_forward)
    r = a W2^T                      (library GEMM, no bias)
    y = LayerNorm(x + dropout(r + b2))   ONE pass (vnx_add_dropout_layernorm_forward with r_bias)

and a backward in which grad_b1 comes out of the in-place relu / dropout backward pass and grad_b2 out of the LayerNorm
backward's partial sums: 4 launches forward instead of 5, 9 backward instead of 11, and 2 + 3 passes over the [rows, d_ffn]
hidden tensor instead of 5.25 + 6.25 (209 MB per encoder layer of a two-clip training step).

Same modules, same parameters, same state dict.  The dropout masks are this package's hash masks (fused_norm.py), not
torch's Philox stream; in eval mode (p = 0) the block is bit-for-bit a GEMM / relu / GEMM / LayerNorm chain.  Everywhere
the fused kernels do not apply (CPU, autocast, other activations or widths, capture without a step_scope) `ffn_block` IS the
reference expression, evaluated by torch.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _lib
from . import fused_norm
from .fused_norm import add_dropout_norm, dropout_site

MAX_CHANNELS = 4096


class _BiasReluDropout(torch.autograd.Function):
    """a = dropout(relu(h + bias)) in place over h (h is the GEMM's fresh output: nobody else holds it); backward:
    grad_h (a new tensor: autograd's gradient buffers are not ours to overwrite) and grad_bias, in one pass."""

    @staticmethod
    def forward(ctx, h, bias, p, seed, seed_tensor):
        lib = _lib.lib()
        assert h.is_contiguous()
        rows, cols = h.numel() // h.shape[-1], h.shape[-1]
        with torch.cuda.device(h.device):
            _lib.check(lib.vnx_bias_relu_dropout_forward(
                _lib.VNX_F32, h.data_ptr(), bias.data_ptr() if bias is not None else None, rows, cols, float(p), int(seed),
                seed_tensor.data_ptr() if seed_tensor is not None else None, _lib.current_stream(h)))
        ctx.mark_dirty(h)
        ctx.save_for_backward(h)
        ctx.p, ctx.has_bias = float(p), bias is not None
        return h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        lib = _lib.lib()
        (y,) = ctx.saved_tensors
        grad = grad.contiguous()
        grad_h = torch.empty_like(grad)
        rows, cols = y.numel() // y.shape[-1], y.shape[-1]
        grad_bias = torch.empty(cols, dtype=torch.float32, device=y.device) if ctx.has_bias else None
        partial = torch.empty(lib.vnx_bias_relu_dropout_partial_bytes(cols), dtype=torch.uint8, device=y.device) \
            if ctx.has_bias else None
        with torch.cuda.device(y.device):
            _lib.check(lib.vnx_bias_relu_dropout_backward(
                _lib.VNX_F32, grad.data_ptr(), y.data_ptr(), grad_h.data_ptr(),
                grad_bias.data_ptr() if grad_bias is not None else None,
                partial.data_ptr() if partial is not None else None, rows, cols, ctx.p, _lib.current_stream(y)))
        return grad_h, grad_bias, None, None, None


def fused_applies(x, linear1, linear2, norm, activation) -> bool:
    return (activation is F.relu and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.shape[-1] == fused_norm.CHANNELS and linear1.in_features == fused_norm.CHANNELS
            and linear2.out_features == fused_norm.CHANNELS and linear1.out_features == linear2.in_features
            and linear1.out_features % 4 == 0 and linear1.out_features <= MAX_CHANNELS
            and linear1.weight.dtype == torch.float32 and linear2.weight.dtype == torch.float32
            and linear1.bias is not None and linear2.bias is not None
            and fused_norm.fused_applies(x, x, norm))


def ffn_block(x, linear1, activation, dropout_mid, linear2, dropout_out, norm):
    """`norm(x + dropout_out(linear2(dropout_mid(activation(linear1(x))))))` (see the module docstring)."""
    def reference():
        return add_dropout_norm(x, linear2(dropout_mid(activation(linear1(x)))), dropout_out, norm)
    if not fused_applies(x, linear1, linear2, norm, activation):
        return reference()
    p_mid, seed_mid, ok = dropout_site(x, dropout_mid)
    if not ok:
        return reference()
    h = F.linear(x, linear1.weight)                           # no bias: it is added in the activation pass
    a = _BiasReluDropout.apply(h, linear1.bias, p_mid, fused_norm._next_seed(), seed_mid)
    r = F.linear(a, linear2.weight)                           # no bias: it is added in the LayerNorm pass
    return add_dropout_norm(x, r, dropout_out, norm, r_bias=linear2.bias)
