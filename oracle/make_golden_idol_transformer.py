"""oracle/make_golden_idol_transformer.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Imports the reference's projects/IDOL/idol/models/deformable_transformer.py under a synthetic
package (its relative imports satisfied with the reference's own MSDeformAttn module file driven
by the reference's pure-PyTorch op and the reference's inverse_sigmoid), runs two frames through
it in fp64 and stores state dict, inputs and outputs in tests/golden/transformer_idol.npz.

    python -m oracle.make_golden_idol_transformer
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

from oracle.make_golden_modules import load_module_class
from oracle.ref_extract import extract

REF = "/root/reference/projects/IDOL/idol"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    torch.set_default_dtype(torch.float64)
    MSDeformAttn = load_module_class(f"{REF}/models/ops/modules/ms_deform_attn.py", "_ref_idol_ops2")
    inv = extract(f"{REF}/util/misc.py", ["inverse_sigmoid"])["inverse_sigmoid"]
    pkg = "_ref_idol_pkg"
    for name in (pkg, pkg + ".models", pkg + ".models.ops", pkg + ".models.ops.modules", pkg + ".util", pkg + ".util.misc"):
        sys.modules[name] = types.ModuleType(name)
        sys.modules[name].__path__ = []
    sys.modules[pkg + ".models.ops.modules"].MSDeformAttn = MSDeformAttn
    sys.modules[pkg + ".util.misc"].inverse_sigmoid = inv
    spec = importlib.util.spec_from_file_location(pkg + ".models.deformable_transformer",
                                                  f"{REF}/models/deformable_transformer.py")
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = pkg + ".models"
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)

    torch.manual_seed(37)      # the constructors below draw their initial weights from the GLOBAL generator: seeded, so that this
    # script regenerates its fixture bit for bit (round-5 review; as make_golden_modules.py)
    gen = torch.Generator().manual_seed(37)
    C, M, L, P, N, Q = 32, 2, 4, 4, 2, 9        # L == P: the reference's sample keeper divides the POINT axis by
    # the per-level valid ratios (deformable_transformer.py:356) and only broadcasts when they are equal; M*L*P >= 30
    tr = mod.DeformableTransformer(d_model=C, nhead=M, num_encoder_layers=1, num_decoder_layers=2,
                                   dim_feedforward=48, dropout=0.0, return_intermediate_dec=True,
                                   num_frames=1, num_feature_levels=L, dec_n_points=P, enc_n_points=P).double()
    with torch.no_grad():
        for p in tr.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen))
    bbox = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(C, C), torch.nn.ReLU(), torch.nn.Linear(C, 4))
                                for _ in range(2)]).double()
    tr.decoder.bbox_embed = bbox
    tr.eval()
    sizes = [(6, 8), (3, 4), (2, 2), (2, 3)]
    srcs = [torch.randn(N, C, h, w, generator=gen) for h, w in sizes]
    poss = [torch.randn(N, C, h, w, generator=gen) for h, w in sizes]
    masks = []
    for h, w in sizes:
        m = torch.zeros(N, h, w, dtype=torch.bool)
        m[1, :, w - max(1, w // 4):] = True
        masks.append(m)
    query_embed = torch.randn(Q, 2 * C, generator=gen)
    with torch.no_grad():
        hs, memory, init_ref, inter_refs, inter_samples, _, _ = tr(srcs, masks, poss, query_embed)
    d = {f"sd.{k}": v.numpy() for k, v in tr.state_dict().items()}
    for i, (s, p, m) in enumerate(zip(srcs, poss, masks)):
        d[f"src{i}"], d[f"pos{i}"], d[f"mask{i}"] = s.numpy(), p.numpy(), m.numpy()
    d.update(query_embed=query_embed.numpy(), hs=hs.numpy(), memory=memory.numpy(), init_ref=init_ref.numpy(),
             inter_refs=inter_refs.numpy(), inter_samples=inter_samples.numpy(), cfg=np.array([C, M, L, P, 1, 2, 48]))
    path = os.path.join(OUT_DIR, "transformer_idol.npz")
    np.savez_compressed(path, **d)
    print("IDOL transformer fixture", os.path.getsize(path) // 1024, "KiB", tuple(hs.shape), tuple(inter_samples.shape))


if __name__ == "__main__":
    main()
