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
from ...msda_ext import packed_promise_holds


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        # Callers that built level_start_index as the running sum of H*W (every caller in
        # the reference does) may tag the tensor; the backward then skips the general-path
        # launches.  Untagged tensors -- or tagged ones whose sizes do not add up to THIS value's
        # length -- stay correct: the library checks on the device.
        ctx.levels_packed = packed_promise_holds(value_spatial_shapes, value_level_start_index, value.shape[1])
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


_LEVEL_TENSORS = {}


def level_tensors(shapes, device):
    """(spatial_shapes [L,2] int64, level_start_index [L] int64) on `device` for a tuple of
    (H, W) pairs, built once per (shapes, device) and reused: no host->device copy and no
    allocation on later calls (which also keeps callers hipGraph-capturable).  The tensors carry
    their host-side meaning -- `_vnx_hw` (the pairs) and `_vnx_levels_packed` -- so that callers
    can check lengths without reading the device."""
    key = (tuple((int(h), int(w)) for h, w in shapes), str(device))
    hit = _LEVEL_TENSORS.get(key)
    if hit is None:
        if len(_LEVEL_TENSORS) > 256:
            _LEVEL_TENSORS.clear()
        sizes = torch.as_tensor(key[0], dtype=torch.long, device=device)
        starts = torch.cat((sizes.new_zeros((1,)), sizes.prod(1).cumsum(0)[:-1]))
        sizes._vnx_hw = key[0]
        hit = _LEVEL_TENSORS[key] = (sizes, mark_levels_packed(starts))
    return hit


def check_flattened_length(spatial_shapes, length):
    """The reference's `assert (shapes[:,0] * shapes[:,1]).sum() == Len_in`
    (ops/modules/ms_deform_attn.py:91): on the host when the tensor came from `level_tensors`,
    else on the device like the reference (a synchronisation per call)."""
    hw = getattr(spatial_shapes, "_vnx_hw", None)
    if hw is not None:
        assert sum(h * w for h, w in hw) == length
    else:
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == length


class MSDeformAttnFusedFunction(Function):
    """MSDeformAttn with the module's prologue (softmax + location arithmetic) inside the kernels:
    apply(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
    reference_points) -> output.  See include/vnext_hip.h (vnx_msda_fused_*).  Gradients for
    value, the two Linear outputs and -- 2-d, per-batch references only -- the reference points."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_offsets, attention_logits,
                reference_points):
        output = MSDA.ms_deform_attn_fused_forward(value, value_spatial_shapes, value_level_start_index,
                                                   sampling_offsets, attention_logits, reference_points)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_offsets,
                              attention_logits, reference_points)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, offsets, logits, ref = ctx.saved_tensors
        want_ref = ctx.needs_input_grad[5]
        if want_ref and (ref.shape[-1] != 2 or ref.shape[0] != value.shape[0]):
            raise RuntimeError("MSDeformAttnFusedFunction: reference-point gradients exist for 2-d, per-batch "
                               "references only (4-d references are detached by every caller)")
        gv, goff, glog, gref = MSDA.ms_deform_attn_fused_backward(
            value, shapes, level_start, offsets, logits, ref, grad_output.contiguous(), want_reference_grad=want_ref)
        return gv, None, None, goff, glog, (gref.to(ref.dtype) if gref is not None else None)
