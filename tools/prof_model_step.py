"""A few SeqFormer-R50 training steps (two T=5 360p clips, the bench's model_step leg without DDP) for
`rocprofv3 --kernel-trace --stats -- python tools/prof_model_step.py`; tools/summarize_model_step.py condenses the
kernel stats into profiles/rNN_model_step_top_kernels.csv (development tool)."""
import os
import sys
import time

T0 = time.time()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T  # noqa: E402
from vnext_amd.registry import build_model, get_seqformer_cfg  # noqa: E402

from vnext_amd import tuning  # noqa: E402

STEPS = int(os.environ.get("VNX_PROF_STEPS", "6"))
print("library gemms:", tuning.enable())                      # recorded rocBLAS / hipBLASLt solutions (VNX_TUNED_GEMMS=0: default)
print("channels-last trunk:", T.enable_channels_last())       # as bench.py's main(): both before the first convolution
print("conv search:", T.enable_conv_search())
if os.environ.get("VNX_CUDNN_BENCHMARK", "0") == "1":
    torch.backends.cudnn.benchmark = True                     # MIOpen: find the convolution algorithms by measurement
dev = "cuda:0"
torch.manual_seed(0)
model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
opt = T.build_optimizer(model)
clips = T.synthetic_clips(2, 5, 360, 640, dev, seed=100, num_instances=4)
if os.environ.get("VNX_PROF_AUTOCAST", "") == "bf16":         # the same step under torch.autocast(bfloat16)
    _step = T.train_step

    def _autocast_step(*a, **k):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return _step(*a, **k)
    T.train_step = _autocast_step
for _ in range(4):                 # warm-up outside the profiled region: MIOpen's kernel search runs in the first steps
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
# rocprofv3 --collection-period <VNX_PROF_DELAY>:<long>:1 starts collecting VNX_PROF_DELAY seconds after launch: wait for it
delay = float(os.environ.get("VNX_PROF_DELAY", "0"))
if delay > 0:
    left = T0 + delay + 1.5 - time.time()
    print("warm-up done after %.1f s; sleeping %.1f s" % (time.time() - T0, max(left, 0.0)))
    if left > 0:
        time.sleep(left)
for _ in range(STEPS):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
t1 = time.time()
for _ in range(STEPS):
    T.train_step(model, opt, clips)
torch.cuda.synchronize()
print("steps", STEPS, "ms/step %.2f" % ((time.time() - t1) * 1e3 / STEPS))
