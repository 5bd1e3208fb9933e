"""The model-level legs of bench.py on one GPU, fp32 beside bf16 autocast, with launch counts (development aid):
    python tools/model_legs.py [--legs 360,720,idol] [--top N]
360: SeqFormer-R50 training step, 2 clips of T = 5 at 360p (bench model_step); 720: the config-4 N = 1 point; idol: config 3.
--top N: the N most frequent kernel names of one bf16 step (torch.profiler), to see what the autocast path still launches."""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--legs", default="360,720,idol")
ap.add_argument("--top", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from vnext_amd import train as _T  # noqa: E402
print("channels-last:", _T.enable_channels_last(), "conv search:", _T.enable_conv_search(), file=sys.stderr)      # as bench.py's main()
legs = a.legs.split(",")
out = {}


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


if "360" in legs:
    for key, amp in (("fp32", False), ("bf16_autocast", True)):
        r = bench.model_step_leg(0, 0, 1, dev, 10, bf16=amp, count=True)
        out["seqformer_train_step_360p_" + key] = {k: r[k] for k in ("ms_per_step", "clips_per_s", "launches_per_step")}
if "720" in legs:
    r = bench.seqformer_720p_leg(dev, timed)
    r["bf16_autocast"]["graphed_trunk"] = bench.graph_leg("seqformer_720p_bf16", 6)      # (a child process)
    out["seqformer_train_step_720p"] = {k: v for k, v in r.items() if k != "config"}
if "idol" in legs or a.top:
    import vnext_amd.models  # noqa: F401
    from vnext_amd import train as T, tuning
    from vnext_amd.registry import build_model, get_idol_cfg
    tuning.enable()
    torch.manual_seed(0)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": str(dev)})).train()
    opt = T.build_optimizer(model, base_lr=1e-4)
    pair = T.synthetic_clips(1, 2, 720, 1280, dev, seed=8, num_instances=8)
    for key, amp in (("fp32", False), ("bf16_autocast", True)):
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                return T.train_step(model, opt, pair)
        for _ in range(3):
            step()
        out["idol_train_step_" + key] = {"ms_per_step": timed(step, 6), "launches_per_step": bench.count_launches(step)}
    if a.top:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        names = collections.Counter()
        dur = collections.Counter()
        for ev in prof.events():
            if ev.device_type == torch.autograd.DeviceType.CUDA:
                names[ev.name[:90]] += 1
                dur[ev.name[:90]] += ev.device_time_total if hasattr(ev, "device_time_total") else 0
        out["idol_bf16_top_kernels"] = [(n, c, round(dur[n])) for n, c in names.most_common(a.top)]
        out["idol_bf16_top_by_time_us"] = [(n, names[n], round(t)) for n, t in dur.most_common(a.top)]
        out["idol_bf16_kernel_time_total_us"] = round(sum(dur.values()))
print(json.dumps(out, indent=1))
