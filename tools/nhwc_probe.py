"""Backbone forward+backward in NCHW vs channels_last: time and launches (torch.profiler)."""
import os, sys, time, collections
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vnext_amd.models.seqformer import ResNet50Trunk
dev = "cuda:0"
torch.manual_seed(0)
for fmt in ("nchw", "nhwc"):
    net = ResNet50Trunk().to(dev).freeze(2)
    x = torch.randn(10, 3, 384, 640, device=dev)
    if fmt == "nhwc":
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    def step():
        outs = net(x)
        sum(o.sum() for o in outs).backward()
        for p in net.parameters():
            p.grad = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    n = sum(1 for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    outs = net(x)
    print(fmt, f"{ms:.2f} ms/step", n, "device events; out strides", [o.stride() for o in outs][:1],
          "checksum", float(sum(o.double().sum() for o in outs)))
