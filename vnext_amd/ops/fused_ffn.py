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
the fused kernels do not apply (CPU, other dtypes, activations or widths, capture without a step_scope) `ffn_block` IS the
reference expression, evaluated by torch.  Under torch.autocast(bfloat16) the two GEMMs run in bf16 and the in-place passes
read and write their bf16 rows (round 6); the residual stream and the LayerNorm stay fp32, as in the eager chain.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .. import _lib
from . import fused_norm
from .fused_norm import add_dropout_norm, dropout_site

MAX_CHANNELS = 4096
# A/B switches (tools/time_fusions.py; VNX_FUSED_FFN=0 / VNX_FUSED_MASKED_LINEAR=0 in the environment): off = the reference
# expressions, evaluated by torch
ENABLE_FFN = os.environ.get("VNX_FUSED_FFN", "1") != "0"
ENABLE_MASKED_LINEAR = os.environ.get("VNX_FUSED_MASKED_LINEAR", "1") != "0"


def _code(dtype):
    return {torch.bfloat16: _lib.VNX_BF16, torch.float16: _lib.VNX_F16}.get(dtype, _lib.VNX_F32)


def autocast_once(t):
    """`t` in the autocast dtype when autocast is on and `t` is an fp32 CUDA tensor, else `t` itself.  Autocast casts the fp32
    INPUT of every GEMM at the call -- a tensor that feeds k Linears is cast k times forward, and autograd casts k gradients back
    and adds them in fp32.  A tensor with several consumers is cast ONCE with this instead (round 6: the decoder's `memory`
    feeds the value projection of all 12 cross-attention calls of a SeqFormer step -- 12 x 78 MB of casts forward, 12 x 78 MB of
    casts + 12 fp32 accumulations of 52 MB backward; a query feeds two Linears).  The cast is part of the graph: gradients meet
    in the 16-bit dtype and come back through one cast."""
    if torch.is_autocast_enabled() and t.is_cuda and t.dtype == torch.float32:
        return t.to(torch.get_autocast_dtype("cuda"))
    return t


def _gemm_dtype_ok(x, *weights) -> bool:
    """fp32 activations and weights: plain fp32 GEMMs, or -- under torch.autocast(bfloat16 / float16) -- 16-bit GEMMs whose
    outputs the in-place passes take as they are (ffn_act.hip / add_norm.hip read 16-bit rows and compute in fp32); the
    activation may already be in the autocast dtype (autocast_once)."""
    if not torch.is_autocast_enabled():
        return x.dtype == torch.float32 and all(w.dtype == torch.float32 for w in weights)
    adt = torch.get_autocast_dtype("cuda")
    # (a weight may already be in the autocast dtype: the layer's shadow copies, ops/shadow_weights.py)
    return (adt in (torch.bfloat16, torch.float16) and x.dtype in (torch.float32, adt)
            and all(w.dtype in (torch.float32, adt) for w in weights))


class _BiasReluDropout(torch.autograd.Function):
    """a = dropout(relu(h + bias)) (relu=True) or h + bias (relu=False, p = 0), rows flagged in row_zero written as zeros,
    in place over h (h is the GEMM's fresh output: nobody else holds it); backward: grad_h (a new tensor: autograd's
    gradient buffers are not ours to overwrite) and grad_bias, in one pass."""

    @staticmethod
    def forward(ctx, h, bias, p, seed, seed_tensor, relu=True, row_zero=None):
        lib = _lib.lib()
        assert h.is_contiguous() and (relu or p == 0.0)
        rows, cols = h.numel() // h.shape[-1], h.shape[-1]
        with torch.cuda.device(h.device):
            _lib.check(lib.vnx_bias_relu_dropout_forward(
                _code(h.dtype), h.data_ptr(), bias.data_ptr() if bias is not None else None,
                row_zero.data_ptr() if row_zero is not None else None, rows, cols, int(bool(relu)), float(p), int(seed),
                seed_tensor.data_ptr() if seed_tensor is not None else None, _lib.current_stream(h)))
        ctx.mark_dirty(h)
        ctx.save_for_backward(*((h,) if relu else ()), *((row_zero,) if row_zero is not None else ()))
        ctx.p, ctx.has_bias, ctx.relu, ctx.masked = float(p), bias is not None, bool(relu), row_zero is not None
        ctx.cols, ctx.dtype = cols, h.dtype
        return h

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        lib = _lib.lib()
        saved = list(ctx.saved_tensors)
        y = saved.pop(0) if ctx.relu else None
        row_zero = saved.pop(0) if ctx.masked else None
        grad = grad.contiguous()
        if grad.dtype != ctx.dtype:
            grad = grad.to(ctx.dtype)
        grad_h = torch.empty_like(grad)
        cols = ctx.cols
        rows = grad.numel() // cols
        grad_bias = torch.empty(cols, dtype=torch.float32, device=grad.device) if ctx.has_bias else None
        partial = torch.empty(lib.vnx_bias_relu_dropout_partial_bytes(cols), dtype=torch.uint8, device=grad.device) \
            if ctx.has_bias else None
        with torch.cuda.device(grad.device):
            _lib.check(lib.vnx_bias_relu_dropout_backward(
                _code(ctx.dtype), grad.data_ptr(), y.data_ptr() if y is not None else None,
                row_zero.data_ptr() if row_zero is not None else None, grad_h.data_ptr(),
                grad_bias.data_ptr() if grad_bias is not None else None,
                partial.data_ptr() if partial is not None else None, rows, cols, ctx.p, _lib.current_stream(grad)))
        return grad_h, grad_bias, None, None, None, None, None


def linear_masked(x, linear, row_mask):
    """`linear(x).masked_fill(row_mask[..., None], 0)` -- the value projection of MSDeformAttn with its padding mask
    (projects/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py:94-96) -- as one GEMM without bias + ONE in-place
    pass (bias add, padding rows zeroed); the backward zeroes the padding rows of the gradient and sums linear's bias
    gradient in the same pass.  ATen: GEMM, copy + fill, and copy + fill + a reduction launch backward.  row_mask: bool
    [..., rows] or None.  Falls back to the expression itself where the kernels do not apply."""
    def reference():
        out = linear(x)
        return out if row_mask is None else out.masked_fill(row_mask[..., None], float(0))
    ok = (x.is_cuda and _gemm_dtype_ok(x, linear.weight) and linear.bias is not None
          and linear.out_features % 4 == 0 and linear.out_features <= MAX_CHANNELS
          and (row_mask is None or (row_mask.dtype == torch.bool and row_mask.shape == x.shape[:-1])))
    if not ok or row_mask is None or not ENABLE_MASKED_LINEAR:       # without a mask the GEMM's own bias epilogue is the cheaper form
        return reference()
    h = F.linear(x, linear.weight)
    rz = row_mask.contiguous().view(torch.uint8) if row_mask is not None else None
    return _BiasReluDropout.apply(h, linear.bias, 0.0, 0, None, False, rz)


def fused_applies(x, linear1, linear2, norm, activation) -> bool:
    return (activation is F.relu and x.is_cuda and _gemm_dtype_ok(x, linear1.weight, linear2.weight)
            and x.shape[-1] == fused_norm.CHANNELS and linear1.in_features == fused_norm.CHANNELS
            and linear2.out_features == fused_norm.CHANNELS and linear1.out_features == linear2.in_features
            and linear1.out_features % 4 == 0 and linear1.out_features <= MAX_CHANNELS
            and linear1.bias is not None and linear2.bias is not None
            and fused_norm.fused_applies(x, x, norm))


def ffn_block(x, linear1, activation, dropout_mid, linear2, dropout_out, norm):
    """`norm(x + dropout_out(linear2(dropout_mid(activation(linear1(x))))))` (see the module docstring)."""
    def reference():
        return add_dropout_norm(x, linear2(dropout_mid(activation(linear1(x)))), dropout_out, norm)
    if not ENABLE_FFN or not fused_applies(x, linear1, linear2, norm, activation):
        return reference()
    p_mid, seed_mid, ok = dropout_site(x, dropout_mid)
    if not ok:
        return reference()
    h = F.linear(x, linear1.weight)                           # no bias: it is added in the activation pass
    a = _BiasReluDropout.apply(h, linear1.bias, p_mid, fused_norm._next_seed(), seed_mid)
    r = F.linear(a, linear2.weight)                           # no bias: it is added in the LayerNorm pass
    return add_dropout_norm(x, r, dropout_out, norm, r_bias=linear2.bias)
