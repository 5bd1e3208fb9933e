"""IDOL training step (config 3: one key / reference pair, 720p, 8 objects), eager against the trunk replayed from hipGraphs
(IDOL.graph_training), fp32 and bf16 autocast, with launch counts:   python tools/idol_graph_probe.py   (development aid)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T, tuning  # noqa: E402
from vnext_amd.registry import build_model, get_idol_cfg  # noqa: E402
dev = "cuda:0"
tuning.enable(); T.enable_channels_last()
out = {}
for graph in (False, True):
    torch.manual_seed(0)
    model = build_model(get_idol_cfg(**{"MODEL.DEVICE": dev})).train()
    model.graph_training = graph
    opt = T.build_optimizer(model, base_lr=1e-4)
    pair = T.synthetic_clips(1, 2, 720, 1280, dev, seed=8, num_instances=8)
    for amp in (False, True):
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                return T.train_step(model, opt, pair)
        try:
            for _ in range(4):
                loss = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                loss = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100
            out["%s_%s" % ("graphed" if graph else "eager", "bf16" if amp else "fp32")] = {
                "ms_per_step": ms, "launches_per_step": bench.count_launches(step), "loss": float(loss)}
        except Exception as e:
            out["%s_%s" % ("graphed" if graph else "eager", "bf16" if amp else "fp32")] = "FAILED %s: %s" % (type(e).__name__, str(e)[:300])
    del model, opt
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
