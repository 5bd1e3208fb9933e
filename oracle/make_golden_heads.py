"""oracle/make_golden_heads.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs the reference's own function bodies for the dynamic mask head
(projects/SeqFormer/seqformer/models/segmentation_condInst.py: dynamic_mask_with_coords,
mask_heads_forward, parse_dynamic_params, aligned_bilinear, compute_locations) on seeded inputs
and stores inputs + outputs in tests/golden/heads_*.npz.  The file's own imports (torchvision,
fvcore) are absent here, so the functions are cut out with `ast` (oracle/ref_extract.py) and
executed with a minimal `self`.

    python -m oracle.make_golden_heads
"""
from __future__ import annotations

import os
import types

import numpy as np
import torch

from oracle.ref_extract import extract

REF = "/root/reference/projects/SeqFormer/seqformer/models/segmentation_condInst.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    torch.set_default_dtype(torch.float64)
    fns = extract(REF, ["parse_dynamic_params", "aligned_bilinear", "compute_locations",
                        "mask_heads_forward", "dynamic_mask_with_coords"])
    # dynamic_mask_with_coords calls the module-level helpers by name: give them to its globals
    g = fns["dynamic_mask_with_coords"].__globals__
    g.update(fns)
    me = types.SimpleNamespace(dynamic_mask_channels=8, weight_nums=[80, 64, 8], bias_nums=[8, 8, 1],
                               mask_out_stride=4)
    me.mask_heads_forward = lambda *a, **k: fns["mask_heads_forward"](me, *a, **k)

    def run(name, N, H, W, num_insts, seed):
        gen = torch.Generator().manual_seed(seed)
        feats = torch.randn(N, 8, H, W, generator=gen)
        n_all = sum(num_insts)
        ref = torch.rand(1, n_all, 2, generator=gen) * torch.tensor([W * 8.0, H * 8.0])
        params = 0.3 * torch.randn(1, n_all, 169, generator=gen)
        feats.requires_grad_(True); ref.requires_grad_(True); params.requires_grad_(True)
        out = fns["dynamic_mask_with_coords"](me, feats, ref, params, num_insts=num_insts,
                                              mask_feat_stride=8, rel_coord=True)
        # gradients of the reference chain (autograd) for a seeded upstream gradient
        gout = torch.randn(out.shape, generator=gen)
        gfeats, gref, gparams = torch.autograd.grad(out, (feats, ref, params), gout)
        feats, ref, params, out = feats.detach(), ref.detach(), params.detach(), out.detach()
        path = os.path.join(OUT_DIR, f"heads_mask_{name}.npz")
        np.savez_compressed(path, feats=feats.numpy(), ref=ref[0].numpy(), params=params[0].numpy(),
                            num_insts=np.array(num_insts), out=out[0].numpy(), grad_out=gout[0].numpy(),
                            grad_feats=gfeats.numpy(), grad_ref=gref[0].numpy(), grad_params=gparams[0].numpy())
        print(f"{name:16s} {tuple(out.shape)} {os.path.getsize(path)/1024:7.1f} KiB")

    run("small", 1, 5, 7, [3], 1)
    run("two_images", 2, 6, 10, [4, 2], 2)
    run("wide", 1, 9, 70, [5], 3)        # wider than one 63-pixel strip
    run("one_pixel", 1, 1, 1, [2], 4)


if __name__ == "__main__":
    main()
