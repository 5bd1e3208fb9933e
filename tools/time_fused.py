"""Prologue fusion at the op level: (softmax + location arithmetic + op) as the reference module
composes them vs the fused kernels; forward and forward+backward, encoder and decoder shapes.
    python tools/time_fused.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vnext_amd.ops.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction, level_tensors  # noqa: E402

dev = "cuda:0"
SHAPES = [(48, 80), (24, 40), (12, 20), (6, 10)]
S = sum(h * w for h, w in SHAPES)
shapes_t, lsi = level_tensors(SHAPES, dev)


def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, B, Lq, ref_dim, ref_div in (("encoder 360p", 5, S, 2, 5), ("decoder 360p", 5, 300, 4, 1)):
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(B, S, 8, 32, device=dev, generator=g)
    off = torch.randn(B, Lq, 8, 4, 4, 2, device=dev, generator=g)
    lg = torch.randn(B, Lq, 8, 16, device=dev, generator=g)
    ref = torch.rand(B // ref_div, Lq, 4, ref_dim, device=dev, generator=g)
    gout = torch.randn(B, Lq, 256, device=dev, generator=g)

    def compose(v, o, l_, r):
        attn = torch.softmax(l_, -1).view(B, Lq, 8, 4, 4)
        rr = r.repeat_interleave(ref_div, 0) if ref_div > 1 else r
        if ref_dim == 2:
            norm = torch.stack([shapes_t[..., 1], shapes_t[..., 0]], -1)
            loc = rr[:, :, None, :, None, :] + o / norm[None, None, None, :, None, :]
        else:
            loc = rr[:, :, None, :, None, :2] + o / 4 * rr[:, :, None, :, None, 2:] * 0.5
        return MSDeformAttnFunction.apply(v, shapes_t, lsi, loc.contiguous(), attn, 64)

    def fused(v, o, l_, r):
        return MSDeformAttnFusedFunction.apply(v, shapes_t, lsi, o, l_, r)
    with torch.no_grad():
        t_c = bench(lambda: compose(value, off, lg, ref))
        t_f = bench(lambda: fused(value, off, lg, ref))
    leaves = [t.clone().requires_grad_(True) for t in (value, off, lg)]
    t_cb = bench(lambda: compose(*leaves, ref).backward(gout))
    t_fb = bench(lambda: fused(*leaves, ref).backward(gout))
    print(f"{name}: forward composed {t_c:7.1f} us, fused {t_f:7.1f} us | fwd+bwd composed {t_cb:7.1f} us, fused {t_fb:7.1f} us")
