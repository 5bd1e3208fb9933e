"""Where the dtype casts of a bf16-autocast training step come from: a TorchDispatchMode census of aten::_to_copy on CUDA tensors
with the innermost vnext_amd frame that issued each (casts without a Python frame are autograd undoing a forward cast).
    python tools/cast_census.py [seq|idol]        (development aid)"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.getcwd())
import vnext_amd.models
from vnext_amd import train as T, tuning
from vnext_amd.registry import build_model, get_seqformer_cfg, get_idol_cfg
from torch.utils._python_dispatch import TorchDispatchMode
dev = "cuda:0"
tuning.enable(); T.enable_channels_last()
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "seq"
if which == "idol":
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": dev})).train()
    opt = T.build_optimizer(model, base_lr=1e-4)
    clips = T.synthetic_clips(1, 2, 720, 1280, dev, seed=8, num_instances=8)
else:
    model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
    opt = T.build_optimizer(model)
    clips = T.synthetic_clips(2, 5, 360, 640, dev, seed=100, num_instances=4)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return T.train_step(model, opt, clips)
for _ in range(3): step()
torch.cuda.synchronize()
where = collections.Counter(); elems = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__
        if name.startswith("_to_copy") and isinstance(args[0], torch.Tensor) and args[0].is_cuda:
            src = args[0]
            dst = (kwargs or {}).get("dtype")
            st = [f for f in traceback.extract_stack() if "/root/repo/vnext_amd" in f.filename or "/tmp/code" in f.filename and "vnext_amd" in f.filename]
            loc = ("%s:%d" % (st[-1].filename.split("vnext_amd/")[-1], st[-1].lineno)) if st else "(autograd / no python frame)"
            key = "%s->%s  %s" % (str(src.dtype)[6:], str(dst)[6:], loc)
            where[key] += 1; elems[key] += src.numel()
        return out
with Mode():
    step()
torch.cuda.synchronize()
print("total _to_copy on cuda:", sum(where.values()))
for k, c in where.most_common(40):
    print("%4d  %8.1f MB  %s" % (c, elems[k] * 4 / 1e6, k))
