"""Same-process A/B of the layer-stack fusions on the SeqFormer-R50 training step (two T=5 360p clips, one GPU):
all fused / no masked value projection / no fused FFN either (development tool; box-to-box variance is +-5 %, so only
numbers of one process compare)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T, tuning  # noqa: E402
from vnext_amd.ops import fused_ffn  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

dev = "cuda:0"
print("library gemms:", tuning.enable())
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
clips = T.synthetic_clips(2, 5, 360, 640, dev, seed=100, num_instances=4)


def timed(n=8):
    for _ in range(3):
        T.train_step(model, opt, clips)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        T.train_step(model, opt, clips)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


for rep in range(2):
    for name, ffn, ml in (("all fused", True, True), ("no masked value projection", True, False), ("neither", False, False)):
        fused_ffn.ENABLE_FFN, fused_ffn.ENABLE_MASKED_LINEAR = ffn, ml
        print("%-30s %7.2f ms/step" % (name, timed()), flush=True)
