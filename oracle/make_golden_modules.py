"""oracle/make_golden_modules.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs in the build container only.  Imports the reference's two `MSDeformAttn` module files
  projects/IDOL/idol/models/ops/modules/ms_deform_attn.py
  projects/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py
under a synthetic package whose `functions.MSDeformAttnFunction.apply` is the reference's own
pure-PyTorch op (ops/functions/ms_deform_attn_func.py:42-62) -- the fallback wiring the
InstMove copy of the module uses (projects/InstMove/.../ops/modules/ms_deform_attn.py:116-121)
-- runs them in fp64 on CPU with seeded weights and inputs, and stores state dict, inputs and
outputs in tests/golden/module_*.npz.

    python oracle/make_golden_modules.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # `python oracle/make_golden_modules.py`
from oracle.make_golden import load_reference  # noqa: E402

REF = "/root/reference/projects"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_module_class(path, pkg):
    core = load_reference()

    class _Fn:
        @staticmethod
        def apply(value, shapes, level_start_index, loc, attn, im2col_step):
            return core(value, shapes, loc, attn)

    for name in (pkg, pkg + ".functions", pkg + ".modules"):
        sys.modules[name] = types.ModuleType(name)
        sys.modules[name].__path__ = []
    sys.modules[pkg + ".functions"].MSDeformAttnFunction = _Fn
    spec = importlib.util.spec_from_file_location(pkg + ".modules.ms_deform_attn", path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = pkg + ".modules"
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod.MSDeformAttn


def randomise(module, gen):
    """The reference initialises attention weights to zero; perturb everything so parity is not
    checked on a degenerate (uniform-attention) state."""
    with torch.no_grad():
        for p in module.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen, dtype=p.dtype))


def npz(name, **arrays):
    path = os.path.join(OUT_DIR, f"module_{name}.npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print(f"{name:24s} {os.path.getsize(path)/1024:8.1f} KiB")


def state(module, prefix="sd."):
    return {prefix + k: v for k, v in module.state_dict().items()}


def main():
    torch.set_default_dtype(torch.float64)
    # the reference modules draw their initial weights from the GLOBAL generator (xavier_uniform_ in _reset_parameters and the
    # nn.Linear constructors): seed it, or the script writes different fixtures on every run (the fixtures
    # committed until round 4 were genuine reference outputs, but not reproducible from this file)
    torch.manual_seed(20)
    gen = torch.Generator().manual_seed(21)
    shapes = torch.tensor([(6, 8), (3, 4), (2, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    C, M, L, P = 64, 2, 3, 4   # 32 channels per head: the tuned kernels' geometry
    N, T, Lq = 2, 3, 7

    # ---- IDOL ---------------------------------------------------------------------------
    cls = load_module_class(f"{REF}/IDOL/idol/models/ops/modules/ms_deform_attn.py", "_ref_idol_ops")
    m = cls(d_model=C, n_levels=L, n_heads=M, n_points=P).double()
    randomise(m, gen)
    query = torch.randn(N, Lq, C, generator=gen)
    src = torch.randn(N, S, C, generator=gen)
    mask = torch.rand(N, S, generator=gen) < 0.1
    for tag, ref in (("ref2", torch.rand(N, Lq, L, 2, generator=gen)),
                     ("ref4", torch.cat([torch.rand(N, Lq, L, 2, generator=gen),
                                         0.3 * torch.rand(N, Lq, L, 2, generator=gen)], -1))):
        out, loc, attn = m(query, ref, src, shapes, lsi, mask)
        npz(f"idol_{tag}", shapes=shapes, lsi=lsi, query=query, ref=ref, src=src, mask=mask,
            out=out, loc=loc, attn=attn, cfg=np.array([C, L, M, P]), **state(m))

    # ---- SeqFormer ------------------------------------------------------------------------
    cls = load_module_class(f"{REF}/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py",
                            "_ref_seq_ops")
    src = torch.randn(N, T, S, C, generator=gen)
    mask = torch.rand(N, T, S, generator=gen) < 0.1
    enc = cls(d_model=C, n_levels=L, n_heads=M, n_points=P, mode='encode').double()
    randomise(enc, gen)
    q_enc = torch.randn(N, T, S, C, generator=gen)
    ref_enc = torch.rand(N, S, L, 2, generator=gen)
    out = enc(q_enc, None, ref_enc, src, shapes, lsi, mask)
    npz("seq_encode", shapes=shapes, lsi=lsi, query=q_enc, ref=ref_enc, src=src, mask=mask, out=out,
        cfg=np.array([C, L, M, P]), **state(enc))

    dec = cls(d_model=C, n_levels=L, n_heads=M, n_points=P, mode='decode').double()
    randomise(dec, gen)
    q = torch.randn(N, Lq, C, generator=gen)
    for tag, qbox, ref in (
            ("first_ref2", torch.randn(N, Lq, C, generator=gen), torch.rand(N, T, Lq, L, 2, generator=gen)),
            ("later_ref4", torch.randn(N, T, Lq, C, generator=gen),
             torch.cat([torch.rand(N, T, Lq, L, 2, generator=gen),
                        0.3 * torch.rand(N, T, Lq, L, 2, generator=gen)], -1))):
        out, out_box, loc, attn = dec(q, qbox, ref, src, shapes, lsi, mask)
        npz(f"seq_decode_{tag}", shapes=shapes, lsi=lsi, query=q, query_box=qbox, ref=ref, src=src,
            mask=mask, out=out, out_box=out_box, loc=loc, attn=attn, cfg=np.array([C, L, M, P]),
            **state(dec))


if __name__ == "__main__":
    main()
