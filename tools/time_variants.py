"""A/B timing of the MSDA kernel variants on one GPU (development tool).

python tools/time_variants.py [--shape dec360|enc360|dec720|enc720] [--dist U|M] [--dtype f32|bf16]
Times forward (and backward) per launch with HIP events around a hipGraph that
replays `inner` launches over rotating input sets (cold: > 256 MiB of inputs in
rotation; warm: one set).
"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import MultiScaleDeformableAttention as MSDA
from vnext_amd import _lib

SHAPES = {"360": [(48, 80), (24, 40), (12, 20), (6, 10)], "720": [(92, 160), (46, 80), (23, 40), (12, 20)]}


def make_inputs(res, Lq, B, dist, dtype, seed, dev="cuda:0"):
    g = torch.Generator(device=dev).manual_seed(seed)
    shapes = torch.tensor(SHAPES[res], dtype=torch.long, device=dev)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, 8, 32, device=dev, generator=g).to(dtype)
    if dist == "U":
        loc = torch.rand(B, Lq, 8, 4, 4, 2, device=dev, generator=g)
    else:
        if Lq == S:  # encoder: the query is a pixel
            refs = []
            for (H, W) in SHAPES[res]:
                ys, xs = torch.meshgrid(torch.arange(H, device=dev) + 0.5, torch.arange(W, device=dev) + 0.5, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1) / W, ys.reshape(-1) / H], -1))
            ref = torch.cat(refs, 0).view(1, S, 1, 1, 1, 2).expand(B, S, 1, 1, 1, 2)
        else:
            ref = torch.rand(B, Lq, 1, 1, 1, 2, device=dev, generator=g)
        theta = torch.arange(8, device=dev) * (2 * 3.141592653589793 / 8)
        d = torch.stack([theta.cos(), theta.sin()], -1)
        d = d / d.abs().max(-1, keepdim=True)[0]
        k = torch.arange(1, 5, device=dev).view(1, 1, 1, 1, 4, 1)
        offs = d.view(1, 1, 8, 1, 1, 2) * k + torch.randn(B, Lq, 8, 4, 4, 2, device=dev, generator=g)
        wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().view(1, 1, 1, 4, 1, 2)
        loc = ref + offs / wh
    attn = torch.softmax(torch.randn(B, Lq, 8, 16, device=dev, generator=g), -1).view(B, Lq, 8, 4, 4)
    go = torch.randn(B, Lq, 256, device=dev, generator=g).to(dtype)
    return shapes, lsi, value, loc.contiguous(), attn.contiguous(), go


def time_graph(fn_list, reps=20):
    """fn_list: callables launched once each inside one graph; returns us per launch."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fn_list:
            f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for f in fn_list:
            f()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / len(fn_list))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


PACKED = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="dec360")
    ap.add_argument("--dist", default="U")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--B", type=int, default=5)
    ap.add_argument("--variants", default="0,1,2,3,4,5,12,13,14,15")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--bwd-only", action="store_true")
    ap.add_argument("--inner", type=int, default=24)
    ap.add_argument("--lq", type=int, default=0)
    a = ap.parse_args()
    res = a.shape[3:]
    S = sum(h * w for h, w in SHAPES[res])
    Lq = a.lq if a.lq > 0 else (300 if a.shape.startswith("dec") else S)
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    e = 4 if dtype == torch.float32 else 2
    B = a.B
    bytes_fwd = B * (e * 256 * S + 4 * 384 * Lq + e * 256 * Lq)
    bytes_bwd = B * (e * 512 * S + e * 256 * Lq + 4 * 768 * Lq)
    set_bytes = B * (e * 256 * S + 4 * 384 * Lq + 2 * e * 256 * Lq)
    nsets = max(2, -(-320 * 2**20 // set_bytes))
    nsets = min(nsets, a.inner)
    sets = [make_inputs(res, Lq, B, a.dist, dtype, 10 + i) for i in range(nsets)]
    points = 128 * B * Lq
    print(f"shape={a.shape} dist={a.dist} dtype={a.dtype} B={B} Lq={Lq} S={S} points={points} "
          f"alg bytes fwd={bytes_fwd/1e6:.1f}MB bwd={bytes_bwd/1e6:.1f}MB sets={nsets}")
    for v in [int(x) for x in a.variants.split(",")]:
        _lib.set_kernel_variant(v)
        def mk(i, bwd):
            sh, lsi, val, loc, attn, go = sets[i % nsets]
            if bwd:
                return lambda: MSDA.ms_deform_attn_backward(val, sh, lsi, loc, attn, go, 64, levels_packed=PACKED)
            return lambda: MSDA.ms_deform_attn_forward(val, sh, lsi, loc, attn, 64)
        for bwd in ([True] if a.bwd_only else [False, True] if a.bwd else [False]):
            cold = time_graph([mk(i, bwd) for i in range(a.inner)])
            warm = time_graph([mk(0, bwd) for _ in range(a.inner)])
            by = bytes_bwd if bwd else bytes_fwd
            print(f"  variant {v:2d} {'bwd' if bwd else 'fwd'}: cold {cold[0]:8.2f} us (min {cold[1]:8.2f})  "
                  f"{by/cold[0]/1e6:6.2f} TB/s {points/cold[0]/1e3:7.2f} Gpt/s | warm {warm[0]:8.2f} us "
                  f"{by/warm[0]/1e6:6.2f} TB/s", flush=True)
    _lib.set_kernel_variant(0)


if __name__ == "__main__":
    main()
