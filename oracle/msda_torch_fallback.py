"""oracle/msda_torch_fallback.py -- TEST INFRASTRUCTURE ONLY.

A PyTorch/CPU statement of the same forward through `F.grid_sample`, i.e. the
technique of the reference's debug fallback (projects/SeqFormer/seqformer/models/
ops/functions/ms_deform_attn_func.py:42-62: per level, bilinear grid_sample with
zero padding and align_corners=False on [B*M, D, H, W], then the attention-
weighted sum).  bench.py times it on the host cores as the "pure-PyTorch
fallback" figure the north star asks for next to the GPU number; tests use it
as a third checker.  Differentiable through autograd (that is the CPU backward
baseline).  Never imported by vnext_amd/.

Two forms.  `msda_grid_sample` folds the T frames of a clip into the batch and
accumulates level by level (one call, no [B*M, D, Lq, L*P] stack): the fastest way
to run the technique on a CPU.  `msda_core_frames` is what the reference actually
executes when its fallback stands in for the extension: ONE call per frame
(ops/modules/ms_deform_attn.py:107-120, `for idx_f in range(nf)`), each call
sampling every level, stacking the L results into [N*M, D, Lq, L*P], weighting and
summing (ms_deform_attn_func.py:42-62) -- the structure SURVEY.md section 8(d)
prescribes for the CPU baseline, and the one bench.py reports as `cpu_baseline.value`.
oracle/time_reference_cpu.py times the reference's own function next to it where
/root/reference exists (profiles/rNN_cpu_reference_fn.json).
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


def msda_core_one_frame(value, spatial_shapes, sampling_locations, attention_weights):
    """One frame through the reference's fallback, same intermediate tensors (func.py:42-62): per level a
    [N*M, D, H, W] view of that level's pixels and a [N*M, Lq, P, 2] grid, grid_sample -> [N*M, D, Lq, P]; the L
    results stacked to [N*M, D, Lq, L, P] -> [N*M, D, Lq, L*P], times the weights [N*M, 1, Lq, L*P], summed."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in spatial_shapes]
    per_level = value.split([h * w for h, w in sizes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(sizes):
        fmap = per_level[lvl].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        grid = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
    weights = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * weights).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def msda_core_frames(value, spatial_shapes, sampling_locations, attention_weights):
    """The module's frame loop around it (ms_deform_attn.py:103-120): value [N,T,S,M,D], loc [N,T,Lq,M,L,P,2],
    attn [N,T,Lq,M,L,P] -> [N,T,Lq,M*D]; every frame's value slice made contiguous first, results concatenated."""
    T = value.shape[1]
    frames = [value[:, t].contiguous() for t in range(T)]
    outs = []
    for t in range(T):
        outs.append(msda_core_one_frame(frames[t], spatial_shapes, sampling_locations[:, t],
                                        attention_weights[:, t].contiguous()).unsqueeze(1))
    return torch.cat(outs, dim=1)
