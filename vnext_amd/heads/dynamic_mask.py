"""Dynamic mask head on the fused HIP kernel.

`dynamic_mask_with_coords` keeps the argument meaning of the reference method
`CondInst_segm.dynamic_mask_with_coords(mask_feats, reference_points, mask_head_params,
num_insts, mask_feat_stride, rel_coord=True)`
(projects/SeqFormer/seqformer/models/segmentation_condInst.py:425-493; IDOL :398-468) and
returns the same `[1, sum(num_insts), H/4, W/4]` logits.  The reference materialises a
`[1, n*10, H*W]` input and runs three grouped convolutions and two pads + an interpolate;
here one kernel reads the 8-channel features and the 169 parameters per instance and
writes the logits -- see vnext_amd/csrc/mask_head.hip.

Differentiable: the backward (training path, `forward_mask_head_train`, :354-401) is a second
fused kernel that recomputes the hidden layers and returns the gradients of the mask
features, the reference points and the parameters (`vnx_dynamic_mask_head_backward`).
No double backward (the reference's loss never needs one).
"""
from __future__ import annotations

import torch

from .. import _lib

NUM_PARAMS = 169          # (8+2)*8 + 8*8 + 8 weights, 8 + 8 + 1 biases
MASK_OUT_STRIDE = 4       # segmentation_condInst.py:40


_INDEX_CACHE = {}


def _instance_image_index(num_insts, device):
    if len(set(num_insts)) == 1:  # the inference case: built once per (count, images, device)
        n = int(num_insts[0])
        key = (n, len(num_insts), str(device))
        idx = _INDEX_CACHE.get(key)
        if idx is None:
            if len(_INDEX_CACHE) > 64:
                _INDEX_CACHE.clear()
            idx = (torch.arange(n * len(num_insts), device=device, dtype=torch.int32) // max(n, 1)).contiguous()
            _INDEX_CACHE[key] = idx
        return idx
    idx = torch.repeat_interleave(torch.arange(len(num_insts), dtype=torch.int32),
                                  torch.as_tensor(num_insts, dtype=torch.int64))
    return idx.to(device, non_blocking=True)


class _DynamicMaskHead(torch.autograd.Function):
    """feats [N,8,H,W], ref [n,2], params [n,169] (fp32, contiguous), inst_image [n] int32 -> [n,2H,2W]"""

    @staticmethod
    def forward(ctx, feats, ref, params, inst_image, stride):
        N, C, H, W = feats.shape
        n_all = ref.shape[0]
        out = torch.empty((n_all, 2 * H, 2 * W), dtype=torch.float32, device=feats.device)
        # A forward whose backward will follow (any differentiable input): the three gradient buffers are allocated NOW and
        # zero-filled by the forward's own launch (vnx_dynamic_mask_head_forward_train, ABI 15) -- the backward then
        # accumulates into them without a zero-fill launch of its own: one launch fewer per training step.
        ctx.grad_bufs = None
        with torch.cuda.device(feats.device):
            if any(ctx.needs_input_grad[:3]):
                bufs = (torch.empty_like(feats), torch.empty_like(ref), torch.empty_like(params))
                st = _lib.lib().vnx_dynamic_mask_head_forward_train(
                    _lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), inst_image.data_ptr(),
                    out.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                    N, C, H, W, n_all, params.shape[1], int(stride), torch.cuda.current_stream(feats.device).cuda_stream)
                ctx.grad_bufs = bufs
            else:
                st = _lib.lib().vnx_dynamic_mask_head_forward(
                    _lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), inst_image.data_ptr(),
                    out.data_ptr(), N, C, H, W, n_all, params.shape[1], int(stride),
                    torch.cuda.current_stream(feats.device).cuda_stream)
        _lib.check(st)
        ctx.save_for_backward(feats, ref, params, inst_image)
        ctx.stride = int(stride)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        feats, ref, params, inst_image = ctx.saved_tensors
        N, C, H, W = feats.shape
        n_all = ref.shape[0]
        grad_out = grad_out.to(torch.float32).contiguous()
        # the buffers the forward zero-filled -- handed to autograd as the gradients, so they serve ONE backward; a second
        # backward over a retained graph allocates fresh ones and takes the entry point that zero-fills for itself
        bufs, ctx.grad_bufs = ctx.grad_bufs, None
        if bufs is not None:
            gfeats, gref, gparams = bufs
            entry = _lib.lib().vnx_dynamic_mask_head_backward_zeroed
        else:
            gfeats, gref, gparams = torch.empty_like(feats), torch.empty_like(ref), torch.empty_like(params)
            entry = _lib.lib().vnx_dynamic_mask_head_backward
        with torch.cuda.device(feats.device):
            st = entry(
                _lib.VNX_F32, feats.data_ptr(), ref.data_ptr(), params.data_ptr(), inst_image.data_ptr(),
                grad_out.data_ptr(), gfeats.data_ptr(), gref.data_ptr(), gparams.data_ptr(),
                N, C, H, W, n_all, params.shape[1], ctx.stride,
                torch.cuda.current_stream(feats.device).cuda_stream)
        _lib.check(st)
        return gfeats, gref, gparams, None, None


def dynamic_mask_head(mask_feats, points, params, inst_image, mask_feat_stride=8):
    """Lower-level surface for callers that already hold a flat instance list:
    mask_feats [N, 8, H, W]; points [n, 2] (x, y image pixels); params [n, 169];
    inst_image [n] int32 (any order, any image) -> logits [n, 2H, 2W].  Differentiable."""
    if not mask_feats.is_cuda:
        raise RuntimeError("dynamic_mask_head: no CPU implementation (HIP library only)")
    if mask_feats.dtype != torch.float32:
        raise RuntimeError(f"dynamic mask head: float32 only for now (got {mask_feats.dtype})")
    if mask_feat_stride // MASK_OUT_STRIDE != 2 or mask_feat_stride % MASK_OUT_STRIDE:
        raise NotImplementedError("dynamic mask head is built for mask_feat_stride / mask_out_stride == 2")
    n = points.shape[0]
    if params.shape[0] != n or inst_image.shape[0] != n:
        raise RuntimeError("dynamic_mask_head: points, params and inst_image disagree on the instance count")
    H, W = mask_feats.shape[-2:]
    if n == 0:
        return torch.empty((0, 2 * H, 2 * W), dtype=torch.float32, device=mask_feats.device)
    return _DynamicMaskHead.apply(mask_feats.contiguous(), points.to(torch.float32).contiguous(),
                                  params.to(torch.float32).contiguous(),
                                  inst_image.to(torch.int32).contiguous(), int(mask_feat_stride))


def dynamic_mask_with_coords(mask_feats, reference_points, mask_head_params, num_insts,
                             mask_feat_stride, rel_coord=True):
    """mask_feats [N, 8, H, W]; reference_points [1, sum n, 2] (image pixels);
    mask_head_params [1, sum n, 169]; num_insts: instances per image.
    -> [1, sum n, 2H, 2W]"""
    if not mask_feats.is_cuda:
        raise RuntimeError("dynamic_mask_with_coords: no CPU implementation (HIP library only)")
    if not rel_coord:
        raise NotImplementedError("dynamic mask head is built with rel_coord=True (the reference's setting)")
    if mask_feat_stride % MASK_OUT_STRIDE != 0 or mask_feat_stride // MASK_OUT_STRIDE != 2:
        raise NotImplementedError("dynamic mask head is built for mask_feat_stride / mask_out_stride == 2")
    N, C, H, W = mask_feats.shape
    n_all = reference_points.shape[1]
    if sum(int(n) for n in num_insts) != n_all or len(num_insts) != N or mask_head_params.shape[1] != n_all:
        raise RuntimeError("dynamic_mask_with_coords: num_insts does not match the instance tensors")
    if mask_feats.dtype != torch.float32:
        raise RuntimeError(f"dynamic mask head: float32 only for now (got {mask_feats.dtype})")
    feats = mask_feats.contiguous()
    # the reference rounds the relative coordinates to fp32 (`.float()`, :447)
    ref = reference_points.reshape(n_all, 2).to(torch.float32).contiguous()
    params = mask_head_params.reshape(n_all, mask_head_params.shape[-1]).to(torch.float32).contiguous()
    if n_all == 0:
        return torch.empty((1, 0, 2 * H, 2 * W), dtype=torch.float32, device=feats.device)
    inst_image = _instance_image_index([int(n) for n in num_insts], feats.device)
    return _DynamicMaskHead.apply(feats, ref, params, inst_image, int(mask_feat_stride))[None]
