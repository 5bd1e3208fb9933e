"""Clip-level data parallelism for the hot path (e).

The reference wraps the model in DistributedDataParallel(broadcast_buffers=False,
find_unused_parameters=True) (detectron2/engine/defaults.py:60-79,380), launches one process
per GPU with the NCCL backend (detectron2/engine/launch.py:27-126) and runs
`SimpleTrainer.run_step` (detectron2/engine/train_loop.py:258-294): loss dict -> sum ->
backward (bucketed gradient all-reduce overlapped with it) -> optimizer step; the optimizer is
AdamW with a 0.1 multiplier on the backbone and full-model gradient-norm clipping at 0.01
(projects/SeqFormer/train_net.py:85-119).

Here: the same step on RCCL (backend "nccl" on ROCm) with the host-side stalls removed --
static graph instead of the unused-parameter walk, gradients as bucket views, no per-step
`.item()` / gloo gather of the loss dict (train_loop.py:296-345) -- and synthetic clips made
on the device, one shard of clips per rank (rank r takes clips r, r+W, ...:
TrainingSampler semantics, projects/SeqFormer/seqformer/data/build.py:18-32).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def init_distributed(backend: str | None = None):
    """One process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard_indices(num_items: int, rank: int, world: int):
    """Indices of this rank's clips: r, r+W, r+2W, ..."""
    return list(range(rank, num_items, world))


def synthetic_clips(num_clips: int, num_frames: int, height: int, width: int, device, seed: int = 0,
                    num_instances: int = 3, num_classes: int = 40):
    """Random frames plus `num_instances` synthetic tracks per clip in the dataset mapper's layout
    (per frame: gt_classes, gt_boxes xyxy pixels, gt_masks, gt_ids, image_size): a rectangle that
    drifts from frame to frame, its mask the filled rectangle."""
    g = torch.Generator(device=device).manual_seed(seed)
    clips = []
    for _ in range(num_clips):
        frames = [torch.rand(3, height, width, device=device, generator=g) * 255.0 for _ in range(num_frames)]
        clip = {"image": frames, "height": height, "width": width}
        if num_instances > 0:
            n = num_instances
            ctr = 0.25 + 0.5 * torch.rand(n, 2, device=device, generator=g)
            size = 0.1 + 0.25 * torch.rand(n, 2, device=device, generator=g)
            cls = torch.randint(0, num_classes, (n,), device=device, generator=g)
            wh = torch.tensor([width, height], device=device, dtype=torch.float32)
            ys = torch.arange(height, device=device)[None, :, None]
            xs = torch.arange(width, device=device)[None, None, :]
            inst = []
            for t in range(num_frames):
                c = (ctr + 0.02 * t).clamp(0.05, 0.95)
                lo, hi = ((c - size / 2).clamp(0, 1) * wh), ((c + size / 2).clamp(0, 1) * wh)
                masks = (xs >= lo[:, 0, None, None]) & (xs < hi[:, 0, None, None]) & \
                        (ys >= lo[:, 1, None, None]) & (ys < hi[:, 1, None, None])
                inst.append({"gt_classes": cls, "gt_boxes": torch.cat([lo, hi], 1), "gt_masks": masks,
                             "gt_ids": torch.arange(n, device=device), "image_size": (height, width)})
            clip["instances"] = inst
        clips.append(clip)
    return clips


def wrap_ddp(model, local_rank: int | None = None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    on_gpu = next(model.parameters()).is_cuda
    return DistributedDataParallel(
        model, device_ids=[local_rank] if on_gpu else None, broadcast_buffers=False,
        find_unused_parameters=False, static_graph=True, gradient_as_bucket_view=True)


def build_optimizer(model, base_lr=2e-4, backbone_multiplier=0.1, weight_decay=1e-4):
    """AdamW, backbone at base_lr * multiplier (train_net.py:85-113)."""
    backbone, rest = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (backbone if "backbone" in name else rest).append(p)
    on_gpu = bool(rest) and rest[0].is_cuda
    return torch.optim.AdamW([{"params": rest, "lr": base_lr},
                              {"params": backbone, "lr": base_lr * backbone_multiplier}],
                             lr=base_lr, weight_decay=weight_decay,
                             fused=True if on_gpu else None)   # one multi-tensor kernel chain per group


def train_step(model, optimizer, clips, clip_max_norm: float = 0.01):
    """SimpleTrainer.run_step without the host synchronisations."""
    loss_dict = model(clips)
    losses = sum(loss_dict.values())
    optimizer.zero_grad(set_to_none=True)
    losses.backward()
    params = [p for g in optimizer.param_groups for p in g["params"]]
    torch.nn.utils.clip_grad_norm_(params, clip_max_norm)   # full-model clipping, train_net.py:115-119
    optimizer.step()
    return losses.detach()
