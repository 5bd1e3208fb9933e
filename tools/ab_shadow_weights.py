"""Same process, alternating: the bf16 training steps with and without the per-layer shadow weights (ops/shadow_weights.py), and
IDOL with / without its graphed trunk.   python tools/ab_shadow_weights.py   (development aid)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import vnext_amd.models  # noqa: F401,E402
from vnext_amd import train as T, tuning  # noqa: E402
from vnext_amd.ops import shadow_weights  # noqa: E402
from vnext_amd.registry import build_model, get_idol_cfg, get_seqformer_cfg  # noqa: E402
dev = "cuda:0"
tuning.enable(); T.enable_channels_last()


def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


out = {}
legs = (("idol_720p_pair", get_idol_cfg, (1, 2, 720, 1280, dev), 1e-4),
        ("seqformer_360p_2clips", get_seqformer_cfg, (2, 5, 360, 640, dev), 2e-4),
        ("seqformer_360p_1clip", get_seqformer_cfg, (1, 5, 360, 640, dev), 2e-4),
        ("seqformer_720p_1clip", get_seqformer_cfg, (1, 5, 720, 1280, dev), 2e-4))
want = sys.argv[1].split(",") if len(sys.argv) > 1 else [l[0] for l in legs]
AMP = os.environ.get("VNX_AB_AMP", "1") == "1"
for name, cfg, clips_args, lr in legs:
    if name not in want:
        continue
    torch.manual_seed(0)
    model = build_model(cfg(**{"MODEL.DEVICE": dev})).train()
    opt = T.build_optimizer(model, base_lr=lr)
    clips = T.synthetic_clips(*clips_args, seed=8, num_instances=8 if "idol" in name else 4)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AMP):
            return T.train_step(model, opt, clips)
    variants = [("plain", False, False), ("shadow", True, False), ("shadow+graph", True, True), ("graph", False, True)]
    res = {v[0]: [] for v in variants}
    launches = {}
    # the eager variants first, all their rounds: a captured graph's private memory pool slows every eager step after it
    # (~5 ms on the SeqFormer step), which made a first version of this tool read the replay as a win on the fp32 legs
    order = [(rnd, v) for rnd in range(4) for v in variants if not v[2]] + [(rnd, v) for rnd in range(4) for v in variants if v[2]]
    for rnd, (vname, sh, gr) in order:
        if True:
            shadow_weights.ENABLED = sh
            model.graph_training = gr
            for _ in range(3):
                step()
            if rnd:
                res[vname].append(timed(step))
            elif not gr:
                launches[vname] = bench.count_launches(step)
    out[name] = {k: {"ms_per_step_runs": [round(x, 2) for x in v], "median": sorted(v)[len(v) // 2], "launches": launches.get(k)}
                 for k, v in res.items()}
    del model, opt
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
