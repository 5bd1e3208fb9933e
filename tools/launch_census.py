"""Launches of one SeqFormer training step by the ATen operator that issued them (torch.profiler),
and by kernel name.  python tools/launch_census.py [--clips 2] [--top 60]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=2)
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--stacks", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
clips = T.synthetic_clips(a.clips, 5, 360, 640, dev, seed=100, num_instances=4)
for _ in range(3):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()


def annotate(mod, name):
    """module forward under a profiler range (forward launches only; the backward is bucketed by autograd node)"""
    fwd = mod.forward

    def wrapped(*args, **kw):
        with torch.profiler.record_function("vnx:" + name):
            return fwd(*args, **kw)
    mod.forward = wrapped


annotate(model.detr.detr.backbone, "backbone")
annotate(model.detr.detr.transformer.encoder, "encoder")
annotate(model.detr.detr.transformer.decoder, "decoder")
annotate(model.criterion, "criterion")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=a.stacks) as prof:
    with torch.profiler.record_function("vnx:forward"):
        loss_dict = model(clips)
        losses = sum(loss_dict.values())
    opt.zero_grad(set_to_none=True)
    with torch.profiler.record_function("vnx:backward"):
        losses.backward()
    with torch.profiler.record_function("vnx:clip+adamw"):
        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 0.01)
        opt.step()
    torch.cuda.synchronize()

LAUNCH = ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemsetAsync", "hipMemcpyAsync", "hipModuleLaunchKernel",
          "hipExtLaunchKernel", "hipMemcpyWithStream", "hipLaunchCooperativeKernel")
phase_count = collections.Counter()
op_count = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or ev.name not in LAUNCH:
        continue
    chain = []
    p = ev.cpu_parent
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    phases = [c for c in chain if c.startswith("vnx:")]
    phase = "/".join(reversed(phases)) if phases else "(outside)"
    phase_count[phase] += 1
    ops = [c for c in chain if not c.startswith("vnx:")]
    outer = ops[-1] if ops else "(direct)"
    if outer.startswith("autograd::engine::evaluate_function: "):
        outer = outer[len("autograd::engine::evaluate_function: "):]
    op_count[(phase, outer)] += 1
print("== launches by phase ==")
for k, c in sorted(phase_count.items(), key=lambda kv: -kv[1]):
    print(f"{c:6d}  {k}")
print("\n== launches by (phase, outermost operator / autograd node) ==")
for (ph, op), c in sorted(op_count.items(), key=lambda kv: -kv[1])[: 3 * a.top]:
    print(f"{c:6d}  {ph:34s} {op[:80]}")
kernels = collections.Counter()
ktime = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kernels[ev.name[:110]] += 1
        ktime[ev.name[:110]] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
print("TOTAL device events (kernels + memcpy/memset):", sum(kernels.values()))
print("\n== by kernel name ==")
for name, c in kernels.most_common(a.top):
    print(f"{c:6d} {ktime[name] / 1e3:9.3f} ms  {name}")
print("\n== by operator (count of calls; self device time) ==")
rows = sorted(prof.key_averages(), key=lambda r: -r.count)
for r in rows[: a.top]:
    dt = getattr(r, "self_device_time_total", None)
    if dt is None:
        dt = r.self_cuda_time_total
    print(f"{r.count:6d} {dt / 1e3:9.3f} ms  {r.key[:100]}")
