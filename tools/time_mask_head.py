"""Forward of the dynamic mask head at the BASELINE frame sizes under forced run counts (development build:
variant 700 + r = r runs per instance, 799 = the strip kernel, 0 = the product's choice); hipGraph replay over rotating
inputs (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.time_variants import time_graph
from vnext_amd import _lib
from vnext_amd.heads import dynamic_mask_with_coords

variants = [int(v) for v in sys.argv[1:]] or [0, 799, 703, 706, 710]
for name, (H, W), n in (("360p", (48, 80), 300), ("720p", (92, 160), 300), ("360p-train", (48, 80), 120)):
    sets = []
    for i in range(8):
        g = torch.Generator(device="cuda").manual_seed(i)
        feats = torch.randn(1, 8, H, W, device="cuda", generator=g)
        ref = torch.rand(1, n, 2, device="cuda", generator=g) * torch.tensor([W * 8.0, H * 8.0], device="cuda")
        params = 0.3 * torch.randn(1, n, 169, device="cuda", generator=g)
        sets.append((feats, ref, params))
    base = None
    for v in variants:
        _lib.set_kernel_variant(v)
        with torch.no_grad():
            out = dynamic_mask_with_coords(*sets[0], [n], 8)
            if base is None:
                base = out
            same = bool(torch.equal(out, base))
            fns = [(lambda s=s: dynamic_mask_with_coords(s[0], s[1], s[2], [n], 8)) for s in sets] * 3
            med, mn = time_graph(fns)
        print(f"mask head {name} n={n} variant {v}: {med:.2f} us (min {mn:.2f})  bit-equal to the first variant: {same}", flush=True)
    _lib.set_kernel_variant(0)
