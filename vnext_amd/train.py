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


# Gradient buckets.  xGMI is point to point (7 links x ~153 GB/s per GPU): a ring all-reduce of b bytes over 8 GPUs
# moves 2 * 7/8 * b over every link, ~11.4 us per MB, on top of ~30 us of launch + synchronisation per collective.
# SeqFormer-R50 has 188 MB of fp32 gradients per step and ~45 ms of backward to hide them under: 48 MB buckets =
# 4 collectives of ~0.55 ms each (DDP's default 25 MB would make 8; the FIRST bucket also closes later with larger
# buckets, but only the LAST one is exposed, and its size is what remains after the others -- the head's ~20 MB).
# VNX_DDP_BUCKET_MB overrides it for sweeps on hardware.
DEFAULT_BUCKET_MB = 48


def ddp_bucket_mb() -> int:
    return int(os.environ.get("VNX_DDP_BUCKET_MB", DEFAULT_BUCKET_MB))


class CommTimer:
    """DDP communication hook that times the gradient all-reduces: every bucket's span (hook call -> collective done,
    device events on GPU, wall clock on CPU) and the part of the communication nothing can hide -- from the moment the
    LAST bucket is ready (the backward has no gradient left to compute) to the end of the last all-reduce.
    `report()` after a synchronisation -> {"buckets", "allreduce_ms_sum", "exposed_allreduce_ms"} of the last step."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.spans = []          # [(start, end)] of the current step: events or floats
        self._last = None

    def begin_step(self):
        self.spans = []

    def hook(self, state, bucket):
        import time
        buf = bucket.buffer()
        group = self.group if self.group is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        on_gpu = buf.is_cuda
        if on_gpu:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        else:
            start, end = [time.perf_counter()], [None]
        buf.div_(world)
        fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
        span = (start, end)
        self.spans.append(span)

        def done(f):
            if on_gpu:
                end.record()          # the future's callback runs ordered after the collective on its stream
            else:
                end[0] = time.perf_counter()
            return f.value()[0]
        return fut.then(done)

    def report(self):
        if not self.spans:
            return None
        first = self.spans[0][0]
        if isinstance(first, list):
            sums = sum((e[0] - s[0]) * 1e3 for s, e in self.spans if e[0] is not None)
            exposed = (self.spans[-1][1][0] - self.spans[-1][0][0]) * 1e3 if self.spans[-1][1][0] is not None else None
        else:
            sums = sum(s.elapsed_time(e) for s, e in self.spans)
            exposed = self.spans[-1][0].elapsed_time(self.spans[-1][1])
        return {"buckets": len(self.spans), "allreduce_ms_sum": round(sums, 3),
                "exposed_allreduce_ms": None if exposed is None else round(exposed, 3)}


def wrap_ddp(model, local_rank: int | None = None, bucket_cap_mb: int | None = None, comm_timer: CommTimer | None = None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    on_gpu = next(model.parameters()).is_cuda
    ddp = DistributedDataParallel(
        model, device_ids=[local_rank] if on_gpu else None, broadcast_buffers=False,
        find_unused_parameters=False, static_graph=True, gradient_as_bucket_view=True,
        bucket_cap_mb=bucket_cap_mb if bucket_cap_mb is not None else ddp_bucket_mb())
    if comm_timer is not None:
        ddp.register_comm_hook(None, comm_timer.hook)
    return ddp


def set_rank_affinity(local_rank: int, local_world: int):
    """Pin this rank's host threads to the cores next to its GPU: the cores of the GPU's NUMA node (sysfs, through the
    device's PCI address), divided among the ranks whose GPUs share that node; all cores divided evenly when the
    topology cannot be read.  8 Python processes each issuing ~3 600 launches per step on one host are the first
    thing that bites at 8 GPUs (the reference leaves placement to the OS: detectron2/engine/launch.py:67-80).
    -> {"numa_node", "cpus"} for the bench line, or None when the platform has no sched_setaffinity."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))

    def node_of(idx):
        try:
            p = torch.cuda.get_device_properties(idx)
            bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                return int(f.read().strip())
        except Exception:
            return -1

    def cpus_of(node):
        try:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                out = []
                for part in f.read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    out.extend(range(int(lo), int(hi or lo) + 1))
                return [c for c in out if c in allowed]
        except Exception:
            return []

    node = node_of(local_rank) if torch.cuda.is_available() else -1
    pool, share, slot = [], local_world, local_rank
    if node >= 0:
        pool = cpus_of(node)
        mates = [r for r in range(local_world) if node_of(r) == node]
        if pool and local_rank in mates:
            share, slot = len(mates), mates.index(local_rank)
    if not pool:
        pool, node = allowed, -1
    per = max(1, len(pool) // max(1, share))
    mine = pool[slot * per:(slot + 1) * per] or pool
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
    except OSError:
        return None
    return {"numa_node": node, "cpus": len(mine)}


def enable_tuned_gemms(path=None):
    """The rocBLAS / hipBLASLt solutions recorded offline for the models' GEMM shapes on MI355X (vnext_amd/tuning):
    SeqFormer-R50 training step 76.5 -> 66.6 ms.  Call once per process before training / inference; a no-op (with the
    reason in the returned dict) on other stacks."""
    from . import tuning
    return tuning.enable(path)


def enable_conv_search(db_dir=None) -> dict:
    """MIOpen's convolution algorithms by measurement instead of by heuristic, answered from the find-db recorded on MI355X
    (vnext_amd/tuning: `enable_conv_search`; SeqFormer-R50 step 59.0 -> 55.4 ms fp32, 54.0 -> 46.0 ms bf16).  Like
    `enable_channels_last`: once per process, BEFORE its first convolution."""
    from . import tuning
    return tuning.enable_conv_search(db_dir)


def enable_channels_last() -> dict:
    """Opt in to the channels-last ResNet trunk for TRAINING steps (vnext_amd/models/seqformer.py: MIOpen's fp32 backward
    convolutions are NHWC kernels either way; given NCHW tensors it transposes around each: SeqFormer step 66.2 -> 64.3 ms).
    Two process-wide settings, which is why this is an explicit call and not an import side effect:
      * PYTORCH_MIOPEN_SUGGEST_NHWC=1 -- PyTorch hands MIOpen an NHWC problem only with it, and reads it ONCE: call this
        before the process runs its first convolution (a later call still flips the trunk's switch, but PyTorch has read
        the variable by then and the inputs are converted for nothing -- this function cannot see that; it is the caller's
        ordering to keep);
      * the trunk's switch `models.seqformer.CHANNELS_LAST`.
    VNX_CHANNELS_LAST=0 or an explicit PYTORCH_MIOPEN_SUGGEST_NHWC=0 opt out.  -> {"enabled", "why"} for the bench line."""
    from .models import seqformer
    if os.environ.get("VNX_CHANNELS_LAST", "1") == "0":
        return {"enabled": False, "why": "VNX_CHANNELS_LAST=0"}
    if not torch.cuda.is_available():
        return {"enabled": False, "why": "no GPU"}
    if os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC", "1") != "1":
        return {"enabled": False, "why": "PYTORCH_MIOPEN_SUGGEST_NHWC is set to something else"}
    os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"
    seqformer.CHANNELS_LAST = True
    return {"enabled": True, "why": "PYTORCH_MIOPEN_SUGGEST_NHWC=1 set before the first convolution; training trunk in channels-last"}


def capture_training_graphs(model, clips, autocast_dtype=None) -> dict:
    """Opt in to the graph-replayed training trunk (`SeqFormer.graph_training` / `IDOL.graph_training`: backbone, transformer
    and heads -- forward AND backward -- as two hipGraphs per input shape) and capture it NOW, on `clips`: one forward +
    backward whose gradients are dropped.  For fixed-size inputs only (every new frame size is a new capture and a new static
    memory pool; Detectron2's multi-scale augmentation wants the eager trunk).

    Call it BEFORE `wrap_ddp`: captured inside DistributedDataParallel.forward, `torch.cuda.make_graphed_callables` dies in
    hipStreamEndCapture on this stack (round 6, one-rank RCCL group: segmentation fault; with the process group alone it
    captures fine).  Captured first, the replayed backward hands the trunk's gradients to the parameters' accumulators like any
    autograd node and DDP's bucket hooks fire there (tests/test_model_ddp.py) -- all of the trunk's buckets when the replayed
    backward returns, i.e. the all-reduce no longer overlaps the trunk's backward.

    What it buys is the host's time, and only where the host is the bound (MI355X, one GPU, the eager figure taken BEFORE the
    capture: a captured graph's private memory pool slows the eager steps that follow it by ~5 ms, which is how a first A/B
    of this misread the fp32 legs): IDOL 720p pair bf16 57.7 -> 49.6 ms, SeqFormer 720p clip bf16 68.9 -> 65.5, two 360p clips
    bf16 ~54 -> ~51; the fp32 SeqFormer steps are bound by their kernels and a replay of ~2 800 graph nodes is SLOWER than
    launching them (59.2 -> 62.5 ms, 720p 92.6 -> 95.5).  So: opt-in, for bf16 / small-batch steps.
    -> {"enabled", "why"} for the bench line."""
    if not hasattr(model, "graph_training"):
        return {"enabled": False, "why": "%s has no graph_training switch" % type(model).__name__}
    if not torch.cuda.is_available() or not next(model.parameters()).is_cuda:
        return {"enabled": False, "why": "no GPU"}
    if dist.is_available() and dist.is_initialized() and isinstance(model, DistributedDataParallel):
        raise ValueError("capture_training_graphs: pass the bare model, before wrap_ddp")
    model.graph_training = True
    was_training = model.training
    model.train()
    with torch.autocast("cuda", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
        losses = sum(model(clips).values())
    losses.backward()
    model.zero_grad(set_to_none=True)
    model.train(was_training)
    torch.cuda.synchronize()
    return {"enabled": True, "why": "training trunk captured (forward + backward hipGraphs) for %d input shape(s)"
                                    % len(getattr(model, "_train_trunks", {}))}


def release_training_graphs(model) -> None:
    """Back to the eager trunk, the captured graphs destroyed NOW (they sit in a reference cycle with the model: `del` alone leaves
    them to the collector).  While a captured training graph is alive, every EAGER step of the process is slower -- 58.6 -> 64.6
    ms on the SeqFormer step, back to 59.1 once the graphs are gone (MI355X, round 6) -- so a process that measures or runs
    both forms releases the graphs in between."""
    import gc
    if hasattr(model, "graph_training"):
        model.graph_training = False
    getattr(model, "_train_trunks", {}).clear()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


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


def train_step(model, optimizer, clips, clip_max_norm: float = 0.01, comm_timer: CommTimer | None = None):
    """SimpleTrainer.run_step without the host synchronisations."""
    if comm_timer is not None:
        comm_timer.begin_step()
    loss_dict = model(clips)
    losses = sum(loss_dict.values())
    optimizer.zero_grad(set_to_none=True)
    losses.backward()
    params = [p for g in optimizer.param_groups for p in g["params"]]
    torch.nn.utils.clip_grad_norm_(params, clip_max_norm)   # full-model clipping, train_net.py:115-119
    optimizer.step()
    return losses.detach()
