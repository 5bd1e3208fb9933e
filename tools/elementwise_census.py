"""Which ATen elementwise / reduction / copy ops of one SeqFormer training step move the most bytes?  torch.profiler,
grouped by operator and input shape (development tool: where the glue of the layer stack still costs)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    T.train_step(model, opt, clips)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total self device time %.1f ms" % (tot / 1e3))
for t, n, k, sh in rows[:70]:
    print("%8.1f us %4d  %-38s %s" % (t, n, k[:38], sh))
