"""oracle/time_reference_cpu.py -- TEST INFRASTRUCTURE ONLY (measurement script, build container only).

SURVEY.md section 8(d) asks for the CPU baseline to be the reference's own `ms_deform_attn_core_pytorch`
(projects/SeqFormer/seqformer/models/ops/functions/ms_deform_attn_func.py:42-62) driven frame by frame as the module
drives the extension (ops/modules/ms_deform_attn.py:107-120).  /root/reference exists only in the build container, not
on the GPU box where bench.py runs, so this script -- run HERE -- times, on the same cores and the same tensors,
    (1) the reference's function itself, imported from /root/reference, in the T-frame loop;
    (2) oracle/msda_torch_fallback.py: msda_core_frames  (the restatement bench.py times on the GPU box's host);
    (3) oracle/msda_torch_fallback.py: msda_grid_sample  (the folded one-call form, rounds 1-4's `cpu_baseline.value`);
checks (1) == (2) bit for bit (forward and all three gradients), and writes profiles/rNN_cpu_reference_fn.json, which
bench.py attaches to `cpu_baseline.reference_fn`.  The ratio (2) / (1) measured here is what ties the GPU box's
`cpu_baseline.value` to the reference's function.

    python oracle/time_reference_cpu.py --round 5
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.make_golden import REF_FUNC, load_reference  # noqa: E402
from oracle.msda_torch_fallback import msda_core_frames, msda_grid_sample  # noqa: E402

SHAPES = {"360p": [(48, 80), (24, 40), (12, 20), (6, 10)], "720p": [(92, 160), (46, 80), (23, 40), (12, 20)]}


def inputs(N, T, Lq, res, seed=3):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(SHAPES[res], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.randn(N, T, S, 8, 32, generator=g)
    loc = torch.rand(N, T, Lq, 8, 4, 4, 2, generator=g)
    attn = torch.softmax(torch.randn(N, T, Lq, 8, 16, generator=g), -1).view(N, T, Lq, 8, 4, 4).contiguous()
    go = torch.randn(N, T, Lq, 256, generator=g)
    return shapes, value, loc, attn, go


def reference_frames(ref, value, shapes, loc, attn):
    """the reference's function in the module's frame loop (ms_deform_attn.py:103-120)"""
    T = value.shape[1]
    frames = [value[:, t].contiguous() for t in range(T)]
    outs = [ref(frames[t], shapes, loc[:, t], attn[:, t].contiguous()).unsqueeze(1) for t in range(T)]
    return torch.cat(outs, dim=1)


def fwd_bwd(fn, value, loc, attn, go):
    leaves = [t.clone().requires_grad_(True) for t in (value, loc, attn)]
    out = fn(*leaves)
    out.backward(go)
    return out.detach(), [t.grad for t in leaves]


def median_ms(fn, value, loc, attn, go, budget_s):
    ts = []
    t_start = time.perf_counter()
    while len(ts) < 3 or (time.perf_counter() - t_start < budget_s and len(ts) < 40):
        t0 = time.perf_counter()
        fwd_bwd(fn, value, loc, attn, go)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=5)
    ap.add_argument("--budget", type=float, default=8.0)
    a = ap.parse_args()
    if not os.path.exists(REF_FUNC):
        raise SystemExit("oracle/time_reference_cpu.py needs /root/reference (build container only)")
    ref = load_reference()
    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    result = {"where": "build container (no GPU); /root/reference mounted", "host_cores": cores, "torch_threads": threads,
              "torch": torch.__version__, "reference_fn": REF_FUNC.replace("/root/reference/", ""), "cases": {}}
    for res, N, T, Lq in (("360p", 1, 5, 300), ("720p", 1, 5, 300)):
        shapes, value, loc, attn, go = inputs(N, T, Lq, res)
        sizes = [tuple(map(int, hw)) for hw in shapes]
        f_ref = lambda v, l, w: reference_frames(ref, v, shapes, l, w)          # noqa: E731
        f_port = lambda v, l, w: msda_core_frames(v, sizes, l, w)               # noqa: E731
        f_fold = lambda v, l, w: msda_grid_sample(v.flatten(0, 1), sizes, l.flatten(0, 1), w.flatten(0, 1)).view(N, T, Lq, 256)  # noqa: E731
        o1, g1 = fwd_bwd(f_ref, value, loc, attn, go)
        o2, g2 = fwd_bwd(f_port, value, loc, attn, go)
        o3, g3 = fwd_bwd(f_fold, value, loc, attn, go)
        same = bool(torch.equal(o1, o2) and all(torch.equal(x, y) for x, y in zip(g1, g2)))
        fold_err = float((o1 - o3).abs().max() / o1.abs().max())
        points = 128 * N * T * Lq
        case = {"workload": f"N={N} clip x T={T} frames, Lq={Lq}, {res}, fp32, fwd + autograd bwd, one call per frame", "points_per_step": points,
                "port_equals_reference_bit_for_bit": same, "folded_form_max_rel_err_vs_reference": fold_err}
        for key, fn in (("reference_fn_frame_loop", f_ref), ("port_frame_loop", f_port), ("port_folded_call", f_fold)):
            ms, n = median_ms(fn, value, loc, attn, go, a.budget)
            case[key] = {"ms_per_step": ms, "gpoints_per_s": points / ms / 1e6, "samples": n}
        case["port_over_reference_time"] = case["port_frame_loop"]["ms_per_step"] / case["reference_fn_frame_loop"]["ms_per_step"]
        result["cases"][res] = case
        print(res, json.dumps(case, indent=1))
    out = os.path.join(ROOT, "profiles", f"r{a.round:02d}_cpu_reference_fn.json")
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
