"""SeqFormer training step, eager vs trunk captured in hipGraphs (SeqFormer.graph_training), 1 and 2 clips per GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vnext_amd.models  # noqa
from vnext_amd import train as T
from vnext_amd.registry import build_model, get_seqformer_cfg
dev = "cuda:0"
for n_clips in (2, 1):
    for graph in (False, True):
        torch.manual_seed(0)
        model = build_model(get_seqformer_cfg(**{"MODEL.DEVICE": dev})).train()
        model.graph_training = graph
        opt = T.build_optimizer(model)
        clips = T.synthetic_clips(n_clips, 5, 360, 640, dev, seed=100, num_instances=4)
        try:
            for _ in range(4):
                loss = T.train_step(model, opt, clips)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                loss = T.train_step(model, opt, clips)
            torch.cuda.synchronize()
            print(f"clips={n_clips} graph={graph}: {(time.perf_counter() - t0) * 100:.2f} ms/step  loss {float(loss):.4f}")
        except Exception as e:
            print(f"clips={n_clips} graph={graph}: FAILED {type(e).__name__}: {str(e)[:200]}")
        del model, opt
        torch.cuda.empty_cache()
