"""Offline GEMM solution selection for the models' Linear / attention GEMMs on MI355X (PyTorch TunableOp over rocBLAS and
hipBLASLt): runs the fp32 legs bench.py times -- SeqFormer-R50 training (two and one T=5 360p clips per GPU, one 720p clip), SeqFormer clip
inference, IDOL video inference at 360p and 720p -- once with tuning enabled and writes the chosen solutions to a CSV that
`vnext_amd.tuning.enable()` loads at run time with tuning OFF.

    python tools/tune_gemms.py gpurun_out/tunableop_mi355x.csv      # on an MI355X; then copy to vnext_amd/tuning/

bf16 (autocast) GEMMs are not tuned HERE: online, inside the autocast step, a library candidate faulted on this stack (round 4);
tools/tune_gemms_bf16.py records their shapes and tunes them offline, one shape at a time (round 6).
The GEMMs are plain library GEMMs; nothing here touches the kernels of this library.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T  # noqa: E402
from vnext_amd.registry import build_model, get_idol_cfg, get_seqformer_cfg  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_mi355x.csv"
dev = "cuda:0"


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def tune(name, fn, warm=3, reps=5):
    for _ in range(warm):
        fn()
    before = timed(fn, reps)
    tunable.tuning_enable(True)
    t0 = time.time()
    fn(); fn()
    torch.cuda.synchronize()
    took = time.time() - t0
    tunable.tuning_enable(False)
    after = timed(fn, reps)
    print("%-34s %8.2f -> %8.2f ms   (tuning pass %.1f s, %d entries so far)" % (name, before, after, took, len(tunable.get_results())),
          flush=True)


tunable.enable(True)
tunable.tuning_enable(False)
tunable.set_max_tuning_duration(int(os.environ.get("VNX_TUNE_MS", "15")))
tunable.set_max_tuning_iterations(int(os.environ.get("VNX_TUNE_ITERS", "20")))
tunable.set_filename(out)
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
for n_clips in (2, 1):
    clips = T.synthetic_clips(n_clips, 5, 360, 640, dev, seed=100, num_instances=4)
    tune(f"seqformer train, {n_clips} clip(s)/GPU", lambda: T.train_step(model, opt, clips))
clips = T.synthetic_clips(1, 5, 720, 1280, dev, seed=104, num_instances=4)      # BASELINE config 4 at N = 1 (bench: seqformer_train_step_720p)
tune("seqformer train, 1 clip/GPU, 720p", lambda: T.train_step(model, opt, clips), warm=2, reps=3)
del opt, clips
model.eval()
model.graph_inference = False            # tune eagerly; the graph-replayed trunk then dispatches the recorded solutions
clip = T.synthetic_clips(1, 5, 360, 640, dev, seed=7, num_instances=0)
with torch.no_grad():
    tune("seqformer clip inference", lambda: model(clip), reps=5)
del model
torch.cuda.empty_cache()
model = build_model(get_idol_cfg(**{"MODEL.DEVICE": dev})).eval()
model.graph_inference = False
g = torch.Generator(device=dev).manual_seed(1)
for name, (h, w) in (("360p", (360, 640)), ("720p", (720, 1280))):
    video = [{"image": [torch.rand(3, h, w, device=dev, generator=g) * 255 for _ in range(36)], "height": h, "width": w}]
    with torch.no_grad():
        tune(f"idol video inference {name}", lambda: model(video), warm=1, reps=2)
print("entries", len(tunable.get_results()), "->", tunable.get_filename(), flush=True)
