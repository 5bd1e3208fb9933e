"""The `MultiScaleDeformableAttention` extension surface, bound to libvnext_hip.so.

Mirrors the two functions the reference's compiled extension exports
(projects/SeqFormer/seqformer/models/ops/src/vision.cpp:13-16):

    ms_deform_attn_forward(value, spatial_shapes, level_start_index,
                           sampling_loc, attn_weight, im2col_step) -> Tensor[B, Lq, M*D]
    ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc,
                            attn_weight, grad_output, im2col_step) -> [gv, gloc, gattn]

with the argument checks and messages of ms_deform_attn.h:20-61 and
ms_deform_attn_cuda.cu:28-52,93-119.  Differences, all supersets:
  * bf16 / fp16 `value` is accepted (the reference dispatches float/double only,
    ms_deform_attn_cuda.cu:64); sampling_loc / attn_weight may then stay fp32.
  * the whole batch goes down in ONE launch; `im2col_step` is validated with the
    reference's rule (batch % min(batch, im2col_step) == 0) but no longer chunks.
  * outputs are torch.empty (the kernels write every element) instead of zeros.
  * a failed launch raises instead of printf (ms_deform_im2col_cuda.cuh:948-952).
There is no CPU path, as in the reference (ms_deform_attn.h:38,60).
"""
from __future__ import annotations

import torch

from . import _lib

_DT = {torch.float32: _lib.VNX_F32, torch.float64: _lib.VNX_F64,
       torch.bfloat16: _lib.VNX_BF16, torch.float16: _lib.VNX_F16}


def _check_inputs(tensors):
    # dispatcher first (ms_deform_attn.h:28,38): a non-GPU `value` has no implementation
    if not tensors[0][1].is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    for name, t in tensors:  # ms_deform_attn_cuda.cu:28-32
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
    for name, t in tensors:  # ms_deform_attn_cuda.cu:34-38
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
    dev = tensors[0][1].device
    for name, t in tensors:
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device}, value is on {dev}")


def _dims_and_types(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5 or spatial_shapes.dim() != 2:
        raise RuntimeError(
            "expected value [B,S,M,D], spatial_shapes [L,2], sampling_loc [B,Lq,M,L,P,2], "
            f"attn_weight [B,Lq,M,L,P]; got {tuple(value.shape)}, {tuple(spatial_shapes.shape)}, "
            f"{tuple(sampling_loc.shape)}, {tuple(attn_weight.shape)}")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 (Long) tensors")
    if tuple(sampling_loc.shape) != (batch, num_query, num_heads, num_levels, num_point, 2) or \
            tuple(attn_weight.shape) != (batch, num_query, num_heads, num_levels, num_point) or \
            spatial_shapes.shape[1] != 2 or level_start_index.numel() != num_levels:
        raise RuntimeError("inconsistent MSDeformAttn argument shapes")
    if value.dtype not in _DT:
        raise RuntimeError(f"ms_deform_attn: unsupported value dtype {value.dtype}")
    if sampling_loc.dtype != attn_weight.dtype or sampling_loc.dtype not in _DT:
        raise RuntimeError("sampling_loc and attn_weight must share a floating dtype")
    if sampling_loc.dtype != value.dtype and not (
            sampling_loc.dtype == torch.float32 and value.dtype in (torch.bfloat16, torch.float16)):
        raise RuntimeError(
            f"sampling_loc dtype {sampling_loc.dtype} must equal value dtype {value.dtype} "
            "(or be float32 with a 16-bit value)")
    # ms_deform_attn_cuda.cu:50-52
    step = min(batch, int(im2col_step))
    if batch > 0 and (step <= 0 or batch % step != 0):
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return (batch, spatial_size, num_heads, channels, num_levels, num_query, num_point)


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)])
    dims = _dims_and_types(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step)
    batch, _, num_heads, channels, _, num_query, _ = dims
    output = torch.empty((batch, num_query, num_heads * channels), dtype=value.dtype,
                         device=value.device)
    with torch.cuda.device(value.device):
        st = _lib.lib().vnx_msda_forward(
            _DT[value.dtype], _DT[sampling_loc.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            output.data_ptr(), *dims, _stream_ptr(value.device))
    _lib.check(st)
    return output


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step, levels_packed: bool = False):
    """`levels_packed=True` (not in the reference signature; optional) promises
    level_start_index is the running sum of H*W with total spatial_size -- what the
    reference's transformer always builds -- and lets the library skip the
    general-path launches.  Left False, the library checks the same thing on the
    device and stays correct for any layout."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)])
    dims = _dims_and_types(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step)
    batch, spatial_size, num_heads, channels, _, num_query, _ = dims
    if grad_output.dtype != value.dtype or grad_output.numel() != batch * num_query * num_heads * channels:
        raise RuntimeError("grad_output must be [B, Lq, M*D] with value's dtype")
    grad_value = torch.empty_like(value)
    grad_sampling_loc = torch.empty_like(sampling_loc)
    grad_attn_weight = torch.empty_like(attn_weight)
    l = _lib.lib()
    flags = _lib.MSDA_LEVELS_PACKED if levels_packed else 0
    ws_bytes = l.vnx_msda_backward_workspace_bytes(_DT[value.dtype], _DT[sampling_loc.dtype], *dims,
                                                   flags)
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device) if ws_bytes else None
    with torch.cuda.device(value.device):
        st = l.vnx_msda_backward(
            _DT[value.dtype], _DT[sampling_loc.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            grad_output.data_ptr(), grad_value.data_ptr(), grad_sampling_loc.data_ptr(),
            grad_attn_weight.data_ptr(), *dims, flags,
            workspace.data_ptr() if workspace is not None else None, ws_bytes,
            _stream_ptr(value.device))
    _lib.check(st)
    return [grad_value, grad_sampling_loc, grad_attn_weight]


# ---------------------------------------------------------------------------------- fused prologue
def packed_promise_holds(spatial_shapes, level_start_index, spatial_size) -> bool:
    """The record-fed grad_value kernel writes every row exactly once only when the levels tile
    [0, S) back to back; with the promise flag set the library does not re-check on the device.  So
    the promise is only passed down when it can be verified on the HOST: both tensors must carry the
    tags `level_tensors` leaves on them and the level sizes must add up to THIS value's length (a
    tagged pair built for another S would otherwise leave grad_value uninitialised)."""
    hw = getattr(spatial_shapes, "_vnx_hw", None)
    return bool(getattr(level_start_index, "_vnx_levels_packed", False)) and hw is not None and \
        sum(h * w for h, w in hw) == int(spatial_size)


def fused_supported(value, spatial_shapes, sampling_offsets, attention_logits, reference_points, level_start_index):
    """True when vnx_msda_fused_* can take these tensors (else: compose the unfused op)."""
    if not (value.is_cuda and value.dim() == 4 and value.shape[-1] == 32):
        return False
    if sampling_offsets.dim() != 6 or sampling_offsets.shape[3] * sampling_offsets.shape[4] != 16:
        return False
    if sampling_offsets.shape[0] != value.shape[0] or reference_points.shape[-1] not in (2, 4):
        return False
    if not packed_promise_holds(spatial_shapes, level_start_index, value.shape[1]):
        return False
    q = sampling_offsets.dtype
    if attention_logits.dtype != q:
        return False
    # reference points in the offsets' dtype, or fp32 beside bf16 offsets (what autocast leaves: VNX_MSDA_REF_F32)
    if reference_points.dtype != q and not (q == torch.bfloat16 and reference_points.dtype == torch.float32):
        return False
    return (value.dtype == torch.float32 and q == torch.float32) or \
        (value.dtype == torch.bfloat16 and q in (torch.float32, torch.bfloat16))


def _require_packed(spatial_shapes, level_start_index, spatial_size):
    if not packed_promise_holds(spatial_shapes, level_start_index, spatial_size):
        raise RuntimeError(
            "ms_deform_attn_fused: the fused kernels need packed levels whose sizes add up to value.shape[1], verifiable "
            "on the host -- pass the (spatial_shapes, level_start_index) pair of vnext_amd.ops.functions.level_tensors "
            "built for this feature pyramid (or use the unfused op)")


def _fused_dims(value, sampling_offsets, reference_points):
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_offsets.shape
    ref_dim = reference_points.shape[-1]
    if reference_points.shape[0] == 0 or B % reference_points.shape[0] != 0:
        raise RuntimeError("ms_deform_attn_fused: batch must be a multiple of the reference batch")
    if reference_points.dtype == torch.float32 and sampling_offsets.dtype != torch.float32:
        ref_dim |= _lib.MSDA_REF_F32      # fp32 reference points beside 16-bit offsets / logits (include/vnext_hip.h)
    return B, S, M, D, L, Lq, P, ref_dim, B // reference_points.shape[0]


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
                                 reference_points):
    """value [B,S,M,32]; sampling_offsets [B,Lq,M,L,P,2], attention_logits [B,Lq,M,L*P] (the two
    Linear outputs, pre-softmax); reference_points [B or B/T, Lq, L, 2|4]  ->  [B, Lq, M*32]"""
    tensors = [("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
               ("sampling_offsets", sampling_offsets), ("attention_logits", attention_logits),
               ("reference_points", reference_points)]
    _check_inputs(tensors)
    B, S, M, D, L, Lq, P, ref_dim, ref_div = _fused_dims(value, sampling_offsets, reference_points)
    _require_packed(spatial_shapes, level_start_index, S)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        st = _lib.lib().vnx_msda_fused_forward(
            _DT[value.dtype], _DT[sampling_offsets.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_offsets.data_ptr(), attention_logits.data_ptr(),
            reference_points.data_ptr(), out.data_ptr(), B, S, M, D, L, Lq, P, ref_dim, ref_div,
            torch.cuda.current_stream(value.device).cuda_stream)
    _lib.check(st)
    return out


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
                                  reference_points, grad_output, want_reference_grad=False):
    """-> [grad_value, grad_sampling_offsets, grad_attention_logits, grad_reference_points | None]"""
    tensors = [("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
               ("sampling_offsets", sampling_offsets), ("attention_logits", attention_logits),
               ("reference_points", reference_points), ("grad_output", grad_output)]
    _check_inputs(tensors)
    B, S, M, D, L, Lq, P, ref_dim, ref_div = _fused_dims(value, sampling_offsets, reference_points)
    _require_packed(spatial_shapes, level_start_index, S)
    gv = torch.empty_like(value)
    goff = torch.empty_like(sampling_offsets)
    glog = torch.empty_like(attention_logits)
    gref = torch.empty((B, Lq, L, 2), dtype=torch.float32, device=value.device) if want_reference_grad else None
    lib = _lib.lib()
    n = lib.vnx_msda_fused_backward_workspace_bytes(_DT[value.dtype], B, S, M, L, Lq, P)
    ws = torch.empty(max(n, 1), dtype=torch.uint8, device=value.device)
    with torch.cuda.device(value.device):
        st = lib.vnx_msda_fused_backward(
            _DT[value.dtype], _DT[sampling_offsets.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_offsets.data_ptr(), attention_logits.data_ptr(),
            reference_points.data_ptr(), grad_output.data_ptr(), gv.data_ptr(), goff.data_ptr(), glog.data_ptr(),
            gref.data_ptr() if gref is not None else None, B, S, M, D, L, Lq, P, ref_dim, ref_div,
            ws.data_ptr(), n, torch.cuda.current_stream(value.device).cuda_stream)
    _lib.check(st)
    return [gv, goff, glog, gref]
