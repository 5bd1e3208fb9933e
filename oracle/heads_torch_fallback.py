"""oracle/heads_torch_fallback.py -- TEST INFRASTRUCTURE ONLY.

Differentiable PyTorch restatement of the dynamic mask head with the flat-instance signature of
`vnext_amd.heads.dynamic_mask_head`, so the CPU-only tests (gloo data parallelism, host logic)
can step the model without a GPU.  Same op chain as the reference
(projects/SeqFormer/seqformer/models/segmentation_condInst.py:404-493, 614-678): relative
coordinates | features -> 10->8->8->1 per-instance 1x1 convs with ReLU -> aligned bilinear x2.
Never imported by the product path."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dynamic_mask_head_torch(mask_feats, points, params, inst_image, mask_feat_stride=8):
    N, C, H, W = mask_feats.shape
    n = points.shape[0]
    if n == 0:
        return mask_feats.new_zeros((0, 2 * H, 2 * W))
    s = mask_feat_stride
    xs = torch.arange(W, device=mask_feats.device, dtype=torch.float32) * s + s // 2
    ys = torch.arange(H, device=mask_feats.device, dtype=torch.float32) * s + s // 2
    relx = (points[:, 0, None, None] - xs[None, None, :]).expand(n, H, W)
    rely = (points[:, 1, None, None] - ys[None, :, None]).expand(n, H, W)
    x0 = torch.cat([relx[:, None], rely[:, None], mask_feats[inst_image.long()]], 1).flatten(2)     # [n, 10, HW]
    w0, w1, w2, b0, b1, b2 = params.split([(C + 2) * 8, 64, 8, 8, 8, 1], 1)
    x1 = F.relu(torch.bmm(w0.reshape(n, 8, C + 2), x0) + b0[:, :, None])
    x2 = F.relu(torch.bmm(w1.reshape(n, 8, 8), x1) + b1[:, :, None])
    y = (torch.bmm(w2.reshape(n, 1, 8), x2) + b2[:, :, None]).reshape(n, 1, H, W)
    y = F.pad(y, (0, 1, 0, 1), mode="replicate")
    y = F.interpolate(y, size=(2 * H + 1, 2 * W + 1), mode="bilinear", align_corners=True)
    y = F.pad(y, (1, 0, 1, 0), mode="replicate")
    return y[:, 0, :2 * H, :2 * W]
