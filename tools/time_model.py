"""Time the SeqFormer-R50 training step (and optionally inference) on one GPU.
    python tools/time_model.py [--steps 10] [--infer] [--instances 4]
Under rocprofv3 --kernel-trace --stats this gives the per-kernel breakdown of a step."""
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
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--instances", type=int, default=4)
ap.add_argument("--infer", action="store_true")
ap.add_argument("--idol", action="store_true", help="IDOL-R50: key/reference pair training step and video inference")
ap.add_argument("--size", default="360x640")
ap.add_argument("--frames", type=int, default=36)
ap.add_argument("--graph", action="store_true", help="SeqFormer: capture the training trunk (forward + backward hipGraphs)")
ap.add_argument("--bf16", action="store_true", help="run the model under torch.autocast(bfloat16)")
ap.add_argument("--phases", action="store_true", help="time forward / backward / optimizer separately")
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
if a.bf16:
    torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=not a.graph).__enter__()   # for the whole script
H_, W_ = (int(v) for v in a.size.split("x"))


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


if a.idol:
    from vnext_amd.registry import get_idol_cfg
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": dev})).train()
    opt = T.build_optimizer(model, base_lr=1e-4)
    pairs = T.synthetic_clips(1, 2, H_, W_, dev, seed=100, num_instances=8)
    for _ in range(a.warmup):
        T.train_step(model, opt, pairs)
    print(f"IDOL train step (1 key/ref pair {H_}x{W_}, 8 objects): {timed(lambda: T.train_step(model, opt, pairs), a.steps):.2f} ms/step")
    if a.infer:
        model.eval()
        g = torch.Generator(device=dev).manual_seed(1)
        video = [{"image": [torch.rand(3, H_, W_, device=dev, generator=g) * 255 for _ in range(a.frames)],
                  "height": H_, "width": W_}]
        model(video)
        ms = timed(lambda: model(video), 3)
        print(f"IDOL video inference: {ms:.1f} ms for {a.frames} frames = {a.frames / ms * 1e3:.1f} frames/s")
        with torch.no_grad():
            ms_net = timed(lambda: [model.inference_forward(video[0]["image"][s:s + 10]) for s in range(0, a.frames, 10)], 3)
        print(f"  network + candidate selection + mask head: {ms_net:.1f} ms; tracker + post-processing: {ms - ms_net:.1f} ms")
    sys.exit(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
model.graph_training = a.graph
opt = T.build_optimizer(model)
clips = T.synthetic_clips(1, 5, H_, W_, dev, seed=100, num_instances=a.instances)


for _ in range(a.warmup):
    T.train_step(model, opt, clips)
print(f"train step: {timed(lambda: T.train_step(model, opt, clips), a.steps):.2f} ms/step")
if a.phases:
    fw = timed(lambda: model(clips), a.steps)
    def fb():
        opt.zero_grad(set_to_none=True)
        sum(model(clips).values()).backward()
    fwbw = timed(fb, a.steps)
    print(f"forward (with matching + losses): {fw:.2f} ms   forward+backward: {fwbw:.2f} ms")
    with torch.no_grad():
        x, mask = model._preprocess(clips)
        t_bb = timed(lambda: model._features(x, mask), a.steps)
        print(f"backbone + input_proj forward (no grad): {t_bb:.2f} ms")
if a.infer:
    model.eval()
    for _ in range(2):
        model(clips[:1])
    print(f"inference, trunk replayed from a hipGraph: {timed(lambda: model(clips[:1]), a.steps):.2f} ms/clip (5 frames)")
    model.graph_inference = False
    for _ in range(2):
        model(clips[:1])
    print(f"inference, eager: {timed(lambda: model(clips[:1]), a.steps):.2f} ms/clip")
