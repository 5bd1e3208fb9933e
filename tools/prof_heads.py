"""Exercises every kernel of the path outside MSDeformAttn so that `rocprofv3 --kernel-trace --stats -- python
tools/prof_heads.py` records them (nothing under profiles/ backed the bench numbers of these):
dynamic_mask_head_runs_kernel / dynamic_mask_head_bwd_kernel (inference frames at 360p / 720p, the training shape forward +
backward), reid_similarity_kernel, bisoftmax_kernel, mask_pack / mask_inter / tracker_frame_kernel (a 40-frame video),
add_dropout_layernorm_fwd / bwd + layernorm_param_grad.  Prints the same timings bench.py reports (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vnext_amd.ops.fused_norm import add_dropout_norm  # noqa: E402


def add_norm_times(device, rows=5 * 5100, steps=20):
    """the residual + dropout + LayerNorm chain of one encoder sub-layer of a T = 5 360p clip: [25 500, 256]"""
    norm = torch.nn.LayerNorm(256).to(device)
    drop = torch.nn.Dropout(0.1).train()
    x = torch.randn(rows, 256, device=device, requires_grad=True)
    r = torch.randn(rows, 256, device=device, requires_grad=True)
    go = torch.randn(rows, 256, device=device)
    for _ in range(3):
        add_dropout_norm(x, r, drop, norm).backward(go)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        add_dropout_norm(x, r, drop, norm).backward(go)
    e1.record()
    e1.synchronize()
    nbytes = rows * 256 * 4 * (4 + 4)       # forward: x, r read, y, z written; backward: grad_y, z read, grad_x, grad_r written
    return {"us_per_fwd_bwd_eager": e0.elapsed_time(e1) * 1e3 / steps, "algorithmic_bytes": nbytes, "rows": rows}


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    out = bench.head_rooflines(dev)
    out["add_dropout_layernorm_25500x256"] = add_norm_times(dev)
    print(json.dumps(out))
