"""`MSDeformAttnFunction`: autograd bridge to the HIP kernels.

Same contract as the reference class
(projects/SeqFormer/seqformer/models/ops/functions/ms_deform_attn_func.py:21-39):
positional signature (value, value_spatial_shapes, value_level_start_index,
sampling_locations, attention_weights, im2col_step), once-differentiable,
backward returns (grad_value, None, None, grad_sampling_loc, grad_attn_weight,
None) and makes grad_output contiguous first.

The reference file also carries `ms_deform_attn_core_pytorch` ("for debug and
test only", :42-62).  The product path has no such fallback; the CPU checker
lives under oracle/ and is never imported from here.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import msda_ext as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        # Callers that built level_start_index as the running sum of H*W (every caller in
        # the reference does) may tag the tensor; the backward then skips the general-path
        # launches.  Untagged tensors stay correct: the library checks on the device.
        ctx.levels_packed = bool(getattr(value_level_start_index, "_vnx_levels_packed", False))
        output = MSDA.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, shapes, level_start, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step, levels_packed=ctx.levels_packed)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


def mark_levels_packed(level_start_index):
    """Tag a level_start_index tensor built as cumsum(H*W) (see MSDeformAttnFunction.forward)."""
    level_start_index._vnx_levels_packed = True
    return level_start_index


def ms_deform_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                   attention_weights, im2col_step: int = 64):
    """Functional spelling of MSDeformAttnFunction.apply."""
    return MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index,
                                      sampling_locations, attention_weights, im2col_step)
