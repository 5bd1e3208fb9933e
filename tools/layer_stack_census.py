"""Kernel launches and device time of the 6 + 6 layer stack alone (DeformableTransformer: encoder + decoder, forward and
backward of a sum loss) at the SeqFormer training shape (two T = 5 360p clips), by operator; and of the encoder alone
(development tool: the glue INSIDE the layers, apart from backbone / criterion / optimizer)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import tuning  # noqa: E402
from vnext_amd.models.seqformer import sine_position  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

dev = "cuda:0"
tuning.enable()
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
tr = model.detr.detr.transformer
query = model.detr.detr.query_embed.weight
N, T, C = 2, 5, 256
shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
srcs = [torch.randn(N, T, C, h, w, device=dev, requires_grad=True) for h, w in shapes]
masks = [torch.zeros(N, T, h, w, dtype=torch.bool, device=dev) for h, w in shapes]
poss = [sine_position(m.flatten(0, 1), C // 2).reshape(N, T, C, *m.shape[-2:]) for m in masks]


def step():
    for p in tr.parameters():
        p.grad = None                                     # (what zero_grad(set_to_none=True) leaves: no accumulation launches)
    query.grad = None
    for t in srcs:
        t.grad = None
    hs, hs_box, memory, init_ref, inter_refs, *_ = tr(srcs, masks, poss, query)
    (hs.sum() + hs_box.sum() + memory.sum() + inter_refs.sum()).backward()


def census(fn, title):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fn()
        torch.cuda.synchronize()
    ops = collections.defaultdict(lambda: [0, 0.0])
    shaped = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        ks = getattr(ev, "kernels", None) or []
        if ks and ev.device_type == torch.autograd.DeviceType.CPU:
            ops[ev.name][0] += len(ks); ops[ev.name][1] += sum(k.duration for k in ks)
            key = (ev.name, str([list(x) for x in (ev.input_shapes or []) if x])[:90])
            shaped[key][0] += len(ks); shaped[key][1] += sum(k.duration for k in ks)
    n = sum(v[0] for v in ops.values()); t = sum(v[1] for v in ops.values())
    print("=== %s: %d launches, %.2f ms of kernels" % (title, n, t / 1e3))
    for name, (c, us) in sorted(ops.items(), key=lambda kv: -kv[1][0])[:28]:
        print("%5d %9.1f us  %s" % (c, us, name[:80]))
    if os.environ.get("VNX_CENSUS_SHAPES"):
        print("--- by (operator, input shapes)")
        for (name, shp), (c, us) in sorted(shaped.items(), key=lambda kv: -kv[1][0])[:int(os.environ["VNX_CENSUS_SHAPES"])]:
            print("%5d %9.1f us  %-28s %s" % (c, us, name[:28], shp))


census(step, "encoder + decoder, forward + backward")
