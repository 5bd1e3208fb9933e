"""Two element-wise chains of a decoder layer, each one launch forward and one backward (vnext_amd/csrc/decoder_glue.hip).

`refined_boxes(delta, reference)` -- the iterative box refinement between decoder layers
(projects/SeqFormer/seqformer/models/deformable_transformer.py:366-380, IDOL's :350-365):

    sigmoid(delta + inverse_sigmoid(reference))                      reference [..., 4]
    sigmoid(cat(delta[..., :2] + inverse_sigmoid(reference), delta[..., 2:]))      reference [..., 2] (the first layer)

which ATen runs as three clamps, a subtraction, a division, a log, a slice + add + cat and a sigmoid.

`time_weighted_sum(x, logits)` -- SeqFormer's temporal weighting of an instance query's frame-level context (:305-312):

    (x * softmax(logits, 1)).sum(1)                                  x [N, T, Q, C], logits [N, T, Q, 1]

a softmax, a broadcast multiply and a reduction forward; two multiplies, two reductions and the softmax backward.

Both ARE those expressions (evaluated by torch) wherever the kernels do not apply: CPU, other dtypes.  Under torch.autocast
the 16-bit tensors a Linear hands over are promoted on the way in and the results are fp32 (round 6).
"""
from __future__ import annotations

import os

import torch

from .. import _lib

EPS = 1e-5
_FLOATS = (torch.float32, torch.bfloat16, torch.float16)      # 16-bit inputs only under autocast: promoted by custom_fwd, results fp32
ENABLE = os.environ.get("VNX_FUSED_DECODER_GLUE", "1") != "0"      # A/B switch: off = the reference expressions, by torch


def inverse_sigmoid(x, eps=EPS):
    """logit with both sides clamped away from 0 (projects/SeqFormer/seqformer/util/misc.py:493-497)"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class _RefineBoxes(torch.autograd.Function):
    # (autocast: the box MLP's bf16 output is promoted on the way in, the boxes come out fp32 -- custom_fwd)
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, delta, reference):
        lib = _lib.lib()
        delta, reference = delta.contiguous(), reference.contiguous()
        out = torch.empty_like(delta)
        rows = delta.numel() // 4
        with torch.cuda.device(delta.device):
            _lib.check(lib.vnx_refine_boxes_forward(_lib.VNX_F32, delta.data_ptr(), reference.data_ptr(), out.data_ptr(), rows,
                                                    reference.shape[-1], EPS, _lib.current_stream(delta)))
        ctx.save_for_backward(out, reference)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        lib = _lib.lib()
        out, reference = ctx.saved_tensors
        grad_out = grad_out.float().contiguous()
        grad_delta = torch.empty_like(out)
        grad_ref = torch.empty_like(reference) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(out.device):
            _lib.check(lib.vnx_refine_boxes_backward(
                _lib.VNX_F32, grad_out.data_ptr(), out.data_ptr(), reference.data_ptr(), grad_delta.data_ptr(),
                grad_ref.data_ptr() if grad_ref is not None else None, out.numel() // 4, reference.shape[-1], EPS,
                _lib.current_stream(out)))
        return grad_delta, grad_ref


def refined_boxes(delta, reference_points):
    """The layer's refined boxes (see the module docstring), WITH their graph: they are also the layer's box prediction (the
    reference's detector evaluates the same expression a second time for its loss, deformable_detr.py:195-213)."""
    if (ENABLE and delta.is_cuda and delta.dtype in _FLOATS and reference_points.dtype in _FLOATS
            and (delta.dtype == torch.float32 or torch.is_autocast_enabled())
            and delta.shape[-1] == 4 and reference_points.shape[-1] in (2, 4) and delta.shape[:-1] == reference_points.shape[:-1]):
        return _RefineBoxes.apply(delta, reference_points)
    if reference_points.shape[-1] == 4:
        moved = delta + inverse_sigmoid(reference_points)
    else:
        moved = torch.cat([delta[..., :2] + inverse_sigmoid(reference_points), delta[..., 2:]], -1)
    return moved.sigmoid()


class _TimeWeightedSum(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, logits):
        lib = _lib.lib()
        x, logits = x.contiguous(), logits.contiguous()
        N, T, Q, C = x.shape
        out = torch.empty(N, Q, C, dtype=x.dtype, device=x.device)
        weights = torch.empty(N, T, Q, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.vnx_time_weighted_sum_forward(_lib.VNX_F32, x.data_ptr(), logits.data_ptr(), out.data_ptr(),
                                                         weights.data_ptr(), N, T, Q, C, _lib.current_stream(x)))
        ctx.save_for_backward(x, weights)
        ctx.logits_shape = logits.shape
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        lib = _lib.lib()
        x, weights = ctx.saved_tensors
        N, T, Q, C = x.shape
        grad_out = grad_out.float().contiguous()
        grad_x = torch.empty_like(x)
        grad_logits = torch.empty(ctx.logits_shape, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.vnx_time_weighted_sum_backward(_lib.VNX_F32, grad_out.data_ptr(), x.data_ptr(), weights.data_ptr(),
                                                          grad_x.data_ptr(), grad_logits.data_ptr(), N, T, Q, C,
                                                          _lib.current_stream(x)))
        return grad_x, grad_logits


def time_weighted_sum(x, logits):
    """`(x * softmax(logits, 1)).sum(1)`: x [N, T, Q, C], logits [N, T, Q, 1] (or [N, T, Q]) -> [N, Q, C]."""
    if (ENABLE and x.is_cuda and x.dtype in _FLOATS and logits.dtype in _FLOATS and x.dim() == 4
            and ((x.dtype == torch.float32 and logits.dtype == torch.float32) or torch.is_autocast_enabled())
            and 1 <= x.shape[1] <= 16 and x.shape[-1] % 4 == 0 and logits.numel() == x.numel() // x.shape[-1]
            and tuple(logits.shape[:3]) == tuple(x.shape[:3]) and x.numel() > 0):
        return _TimeWeightedSum.apply(x, logits)
    if logits.dim() == 3:
        logits = logits.unsqueeze(-1)
    return (x * torch.softmax(logits, 1)).sum(1)
