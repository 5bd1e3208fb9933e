"""MIOpen's convolution algorithms for the models' shapes, found by measurement ONCE and kept.

PyTorch asks MIOpen for a convolution algorithm through miopenFindConvolution*Algorithm; with `torch.backends.cudnn.benchmark`
off the library answers from its heuristic, with it on it times its solvers the first time a problem is seen (55-70 s for the
problems of one SeqFormer step) and keeps the answer in its USER find-db, a text file under MIOPEN_USER_DB_PATH.  This tool
runs every model leg bench.py times -- SeqFormer training (two / one 360p clips, one 720p clip) and IDOL key / reference
training, fp32 and bf16 autocast, and the inference legs -- with the search on and MIOPEN_USER_DB_PATH pointing at <out dir>:

    python tools/record_miopen_db.py gpurun_out/miopen_db       # on an MI355X; then copy the *.ufdb.txt / *.udb.txt files
                                                                # into vnext_amd/tuning/miopen_userdb/

`vnext_amd.tuning.enable_conv_search()` hands MIOpen a private copy of the recorded directory: a recorded problem is answered
from the file (no timing at run time), an unrecorded one is searched once per process.  The convolutions are plain library
convolutions; nothing here touches this library's kernels.
"""
import os
import sys
import time

out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/miopen_db")
os.makedirs(out, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = out          # before the first convolution of the process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T, tuning  # noqa: E402
from vnext_amd.registry import build_model, get_idol_cfg, get_seqformer_cfg  # noqa: E402

torch.backends.cudnn.benchmark = True
tuning.enable()
T.enable_channels_last()
dev = "cuda:0"


def run(name, fn, n=3):
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-44s %6.1f s" % (name, time.time() - t0), flush=True)


torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
for n_clips, (h, w), seed in ((2, (360, 640), 100), (1, (360, 640), 100), (1, (720, 1280), 104)):
    clips = T.synthetic_clips(n_clips, 5, h, w, dev, seed=seed, num_instances=4)
    for amp in (False, True):
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                return T.train_step(model, opt, clips)
        run("seqformer train %d x %dp %s" % (n_clips, h, "bf16" if amp else "fp32"), step)
model.eval()
model.graph_inference = False
clip = T.synthetic_clips(1, 5, 360, 640, dev, seed=7, num_instances=0)
with torch.no_grad():
    run("seqformer clip inference", lambda: model(clip))
del model, opt
torch.cuda.empty_cache()
model = build_model(get_idol_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model, base_lr=1e-4)
pair = T.synthetic_clips(1, 2, 720, 1280, dev, seed=8, num_instances=8)
for amp in (False, True):
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            return T.train_step(model, opt, pair)
    run("idol train pair 720p %s" % ("bf16" if amp else "fp32"), step)
model.eval()
model.graph_inference = False
g = torch.Generator(device=dev).manual_seed(1)
for name, (h, w) in (("360p", (360, 640)), ("720p", (720, 1280))):
    video = [{"image": [torch.rand(3, h, w, device=dev, generator=g) * 255 for _ in range(36)], "height": h, "width": w}]
    with torch.no_grad():
        run("idol video inference %s" % name, lambda: model(video), n=1)
print(sorted((f, os.path.getsize(os.path.join(out, f))) for f in os.listdir(out)))
