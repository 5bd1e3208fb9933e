"""oracle/msda_torch_fallback.py -- TEST INFRASTRUCTURE ONLY.

A PyTorch/CPU statement of the same forward through `F.grid_sample`, i.e. the
technique of the reference's debug fallback (projects/SeqFormer/seqformer/models/
ops/functions/ms_deform_attn_func.py:42-62: per level, bilinear grid_sample with
zero padding and align_corners=False on [B*M, D, H, W], then the attention-
weighted sum).  bench.py times it on the host cores as the "pure-PyTorch
fallback" figure the north star asks for next to the GPU number; tests use it
as a third checker.  Differentiable through autograd (that is the CPU backward
baseline).  Never imported by vnext_amd/.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B,S,M,D]; spatial_shapes list/tensor of (H,W); loc [B,Lq,M,L,P,2];
    attn [B,Lq,M,L,P] -> [B,Lq,M*D]"""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in spatial_shapes]
    # [B,S,M,D] -> [B*M, D, S] once; levels are then contiguous slices of the last axis
    planes = value.permute(0, 2, 3, 1).reshape(B * M, D, S)
    grid = (sampling_locations * 2.0 - 1.0).permute(0, 2, 1, 3, 4, 5).reshape(B * M, Lq, L, P, 2)
    weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(B * M, 1, Lq, L, P)
    acc = value.new_zeros((B * M, D, Lq))
    start = 0
    for lvl, (h, w) in enumerate(sizes):
        fmap = planes[:, :, start:start + h * w].reshape(B * M, D, h, w)
        start += h * w
        taps = F.grid_sample(fmap, grid[:, :, lvl], mode="bilinear", padding_mode="zeros",
                             align_corners=False)          # [B*M, D, Lq, P]
        acc = acc + (taps * weights[:, :, :, lvl]).sum(-1)
    return acc.reshape(B, M * D, Lq).transpose(1, 2).contiguous()
