"""Where a SeqFormer training step spends its wall time (synchronising between phases, so the sum
exceeds the pipelined step).  python tools/step_breakdown.py [--graph]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graph", action="store_true")
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
model.graph_training = a.graph
opt = T.build_optimizer(model)
clips = T.synthetic_clips(1, 5, 360, 640, dev, seed=100, num_instances=4)
for _ in range(3):
    T.train_step(model, opt, clips)

marks = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    marks[name] = marks.get(name, 0.0) + (t - t0) * 1e3
    return t


for _ in range(a.steps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    targets = model.prepare_targets(clips)
    t = tick("prepare_targets", t)
    frames = [f for c in clips for f in c["image"]]
    if a.graph:
        hs, logits, boxes, ref0, ref_rest, feats = model._graphed_train_trunk(torch.stack(frames))
    else:
        x, srcs, hs, memory, logits, boxes, refs = model._run(clips, want_refs=True)
        feats = model._mask_features(srcs, memory)
    t = tick("trunk forward", t)
    ind = model.criterion.matcher.match_all_layers(logits, boxes, targets)
    t = tick("matching (cost + host LSAP)", t)
    losses = model(clips)
    t = tick("full forward (all of the above again + mask head + criterion)", t)
    total = sum(losses.values())
    opt.zero_grad(set_to_none=True)
    total.backward()
    t = tick("backward", t)
    params = [p for g in opt.param_groups for p in g["params"]]
    torch.nn.utils.clip_grad_norm_(params, 0.01)
    t = tick("clip_grad_norm", t)
    opt.step()
    t = tick("AdamW step", t)
for k, v in marks.items():
    print(f"{k:70s} {v / a.steps:8.2f} ms")
