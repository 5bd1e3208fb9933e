"""Host-side profile (cProfile) of a training step: python tools/profile_host.py [--idol]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T  # noqa: E402
from vnext_amd.registry import build_model, get_idol_cfg, get_seqformer_cfg  # noqa: E402

idol = "--idol" in sys.argv
dev = "cuda:0"
torch.manual_seed(0)
cfg = get_idol_cfg(**{"MODEL.DEVICE": dev}) if idol else get_seqformer_cfg(**{"MODEL.DEVICE": dev})
model = build_model(cfg).train()
opt = T.build_optimizer(model)
clips = T.synthetic_clips(1, 2 if idol else 5, 360, 640, dev, seed=100, num_instances=8 if idol else 4)
for _ in range(3):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
