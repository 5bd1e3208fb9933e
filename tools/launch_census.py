"""Which lines of this package launch the kernels of one SeqFormer training step?  torch.profiler with Python stacks:
every device kernel is attributed to the innermost frame under vnext_amd/ (or "autograd" for the backward engine) and to
its ATen operator; prints launches and device time per (frame, operator) (development tool: the glue left in the step)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T, tuning  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

dev = "cuda:0"
tuning.enable()
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
clips = T.synthetic_clips(2, 5, 360, 640, dev, seed=100, num_instances=4)
for _ in range(3):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    T.train_step(model, opt, clips)
    torch.cuda.synchronize()


def frame_of(ev):
    for fr in ev.stack or ():
        if "vnext_amd/" in fr and "torch/" not in fr:
            return fr.split("vnext_amd/")[-1].strip()
    return "(autograd / optimizer)"


by = collections.defaultdict(lambda: [0, 0.0])
ops = collections.defaultdict(lambda: [0, 0.0])
launches = 0
for ev in prof.events():
    ks = getattr(ev, "kernels", None) or []
    if not ks or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    t = sum(k.duration for k in ks)
    key = (frame_of(ev), ev.name)
    by[key][0] += len(ks); by[key][1] += t
    ops[ev.name][0] += len(ks); ops[ev.name][1] += t
    launches += len(ks)
print("kernel launches attributed: %d, device time %.1f ms" % (launches, sum(v[1] for v in by.values()) / 1e3))
print("--- by operator")
for name, (n, t) in sorted(ops.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%5d launches %9.1f us  %s" % (n, t, name[:90]))
print("--- by (frame, operator), most launches first")
for (fr, name), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][0])[:90]:
    print("%5d %9.1f us  %-60s %s" % (n, t, fr[:60], name[:50]))

print("--- glue operators by input shapes, most device time first")
shapes = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    ks = getattr(ev, "kernels", None) or []
    if not ks or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    if ev.name in ("aten::add_", "aten::sum", "aten::copy_", "aten::add", "aten::mul", "aten::cat", "aten::div", "aten::clamp_min",
                   "aten::threshold_backward", "aten::fill_", "aten::sub", "aten::bmm", "aten::clamp"):
        key = (ev.name, str(ev.input_shapes)[:90])
        shapes[key][0] += len(ks); shapes[key][1] += sum(k.duration for k in ks)
for (name, sh), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%5d %9.1f us  %-26s %s" % (n, t, name, sh))
