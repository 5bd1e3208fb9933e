"""What the two deformable transformers (seqformer_transformer.py, idol_transformer.py) share around their layer stacks: the
flattened multi-level buffers the encoder reads, the reference points a decoder layer samples around, and the iterative box
refinement between decoder layers.  Behaviour as the reference's (projects/SeqFormer/seqformer/models/deformable_transformer.py
:78-129, 336-385; projects/IDOL/idol/models/deformable_transformer.py:135-198, 325-375), expressed once and without the
per-level lists + `cat` and the per-layer recomputation of constants."""
from __future__ import annotations

import torch

from ..ops.decoder_glue import inverse_sigmoid, refined_boxes  # noqa: F401  (re-exported: the models import them from here)
from ..ops.functions import level_tensors


def flatten_levels(srcs, masks, pos_embeds, level_embed):
    """Per level [..., C, H_l, W_l] features and position embeddings and [..., H_l, W_l] padding masks -> ONE buffer each:
    features [..., S, C], position + level embedding [..., S, C], mask [..., S], the levels back to back in level order (the
    packed layout the op's grad_value kernels rely on), plus the op's (spatial_shapes, level_start_index) pair and the list
    of level sizes.  Every level is written straight into its slice: no list of transposed views, no `cat`."""
    lead, channels = srcs[0].shape[:-3], srcs[0].shape[-3]
    sizes = [tuple(int(v) for v in s.shape[-2:]) for s in srcs]
    total = sum(h * w for h, w in sizes)
    feats = srcs[0].new_empty(*lead, total, channels)
    # (the sum's own dtype: under bf16 autocast the position embeddings may arrive in bf16 while the level embedding is an fp32
    #  parameter -- the promoted sum must not be rounded back into a bf16 buffer; the `cat`-based form kept it: ADVICE r5)
    pos = pos_embeds[0].new_empty(*lead, total, channels, dtype=torch.result_type(pos_embeds[0], level_embed))
    pad = masks[0].new_empty(*lead, total)
    at = 0
    for lvl, (h, w) in enumerate(sizes):
        rows = slice(at, at + h * w)
        feats[..., rows, :] = srcs[lvl].flatten(-2).transpose(-1, -2)
        pos[..., rows, :] = pos_embeds[lvl].flatten(-2).transpose(-1, -2) + level_embed[lvl]
        pad[..., rows] = masks[lvl].flatten(-2)
        at += h * w
    shapes_t, start_t = level_tensors(sizes, feats.device)      # cached device tensors, tagged packed, host-side sizes attached
    return feats, pad, pos, shapes_t, start_t, sizes


class ReferenceScaler:
    """reference points -> what a decoder layer samples around: points (2 components) or boxes (4) times the valid ratios of
    every level.  The two ratio tensors are constants of a forward pass: built once here, not once per layer."""

    def __init__(self, valid_ratios, extra_axes):
        # valid_ratios [N, L, 2]; reference points [N, (T,) Q, 2|4]: `extra_axes` singleton axes sit between N and L
        index = (slice(None),) + (None,) * extra_axes
        self.xy = valid_ratios[index]
        self.xywh = torch.cat([valid_ratios, valid_ratios], -1)[index]

    def __call__(self, reference_points):
        scale = self.xywh if reference_points.shape[-1] == 4 else self.xy
        return reference_points.unsqueeze(-2) * scale


def refine_reference(delta, reference_points):
    """`refined_boxes`, detached: what the reference feeds on to the next layer."""
    return refined_boxes(delta, reference_points).detach()
