"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs ONLY in the build container, where /root/reference is mounted: it imports
the reference's own pure-PyTorch implementation
    projects/SeqFormer/seqformer/models/ops/functions/ms_deform_attn_func.py:42-62
        ms_deform_attn_core_pytorch
(the file's one hard dependency, the compiled `MultiScaleDeformableAttention`
module imported at :18, is satisfied with an empty stub) and records its
forward outputs and autograd gradients for a fixed list of seeded inputs into
tests/golden/msda_*.npz.  The GPU box has no /root/reference, so the parity
tests read these files instead.

Input conventions are those of the reference's only op test,
projects/SeqFormer/seqformer/models/ops/test.py:21-36,85 (seed 3, shapes
[(6,4),(3,2)], N,M,D=1,2,2, Lq,L,P=2,2,2, value=rand*0.01, loc=rand,
attn=rand+1e-5 normalised over (L,P); gradient channel list
30,32,64,71,1025,2048,3096), drawn from the same CPU generator in the same
order, plus a few model-shaped and edge cases of our own whose expected
outputs still come from the reference function.

    python oracle/make_golden.py            # rewrites tests/golden/msda_*.npz
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF_FUNC = ("/root/reference/projects/SeqFormer/seqformer/models/ops/functions/"
            "ms_deform_attn_func.py")
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
BIG_D = 1024  # above this only a digest of grad_value is stored (fixture size)


def load_reference():
    sys.modules.setdefault("MultiScaleDeformableAttention",
                           types.ModuleType("MultiScaleDeformableAttention"))
    spec = importlib.util.spec_from_file_location("_vnext_ref_msda_func", REF_FUNC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ms_deform_attn_core_pytorch


def lsi_of(shapes: torch.Tensor) -> torch.Tensor:
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def input_digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_reference(ref, value, shapes, loc, attn, grad_out=None, want_grad=True):
    """fp64 forward (+ autograd backward with upstream grad_out)."""
    v = value.double().clone().requires_grad_(want_grad)
    s = loc.double().clone().requires_grad_(want_grad)
    a = attn.double().clone().requires_grad_(want_grad)
    out = ref(v, shapes, s, a)
    res = {"out_f64": out.detach().numpy()}
    res["out_f32"] = ref(value.float(), shapes, loc.float(), attn.float()).numpy()
    if want_grad:
        if grad_out is None:
            grad_out = torch.ones_like(out)
        out.backward(grad_out.double())
        res["grad_out"] = grad_out.double().numpy()
        res["grad_value"] = v.grad.numpy()
        res["grad_loc"] = s.grad.numpy()
        res["grad_attn"] = a.grad.numpy()
    return res


def save(name, shapes, value, loc, attn, res, store_inputs=True, recipe=""):
    d = {"shapes": shapes.numpy(), "lsi": lsi_of(shapes).numpy(), "recipe": np.array(recipe)}
    d["digest"] = np.array(input_digest(value.numpy(), loc.numpy(), attn.numpy()))
    if store_inputs:
        d.update(value=value.numpy(), loc=loc.numpy(), attn=attn.numpy())
    gv = res.get("grad_value")
    if gv is not None and gv.shape[-1] > BIG_D:
        res = dict(res)
        del res["grad_value"]
        res["grad_value_head"] = gv[..., :40].copy()
        res["grad_value_tail"] = gv[..., -40:].copy()
        res["grad_value_rowsum"] = gv.sum(-1)
    d.update(res)
    path = os.path.join(OUT_DIR, f"msda_{name}.npz")
    np.savez_compressed(path, **d)
    print(f"{name:28s} {os.path.getsize(path)/1024:8.1f} KiB")


TESTPY_CHANNELS = [30, 32, 64, 71, 1025, 2048, 3096]  # test.py:85


def testpy_draws():
    """Yield (name, shapes, value, loc, attn, grad_out|None): the exact draws of
    the reference's test.py in program order (needs torch only, not the
    reference; the tests re-run it to rebuild the inputs of the wide-channel
    cases, whose inputs are not stored, and verify them against the digest)."""
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    state = torch.random.get_rng_state()
    torch.manual_seed(3)

    def draw(channels):
        value = torch.rand(N, S, M, channels) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, attn

    try:
        # check_forward_equal_with_pytorch_double, then _float (test.py:31-60)
        for tag in ("fwd_double", "fwd_float"):
            value, loc, attn = draw(D)
            yield f"testpy_{tag}", shapes, value, loc, attn, None
        # check_gradient_numerical for each channel count (test.py:63-86)
        for channels in TESTPY_CHANNELS:
            value, loc, attn = draw(channels)
            g = torch.Generator().manual_seed(1000 + channels)
            grad_out = torch.randn(N, Lq, M * channels, generator=g, dtype=torch.float64)
            yield f"testpy_grad_d{channels}", shapes, value, loc, attn, grad_out
    finally:
        torch.random.set_rng_state(state)


def reference_test_py_cases(ref):
    for name, shapes, value, loc, attn, grad_out in testpy_draws():
        res = run_reference(ref, value, shapes, loc, attn, grad_out, want_grad=grad_out is not None)
        save(name, shapes, value, loc, attn, res, store_inputs=value.shape[-1] <= BIG_D,
             recipe="reference test.py draw order, torch.manual_seed(3); "
                    "grad_out=randn(Generator seed 1000+D)")


def model_like(seed, B, M, D, shapes_list, Lq, P, spread=1.0):
    """Reference points + per-head directional offsets (module init bias,
    ops/modules/ms_deform_attn.py:65-73) + noise; a few samples leave [0,1]."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes_list, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, generator=g)
    ref_pts = torch.rand(B, Lq, 1, 1, 1, 2, generator=g) * 1.1 - 0.05
    theta = torch.arange(M, dtype=torch.float32) * (2.0 * np.pi / M)
    direction = torch.stack([theta.cos(), theta.sin()], -1)
    direction = direction / direction.abs().max(-1, keepdim=True)[0]
    k = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, 1, 1, P, 1)
    offs = direction.view(1, 1, M, 1, 1, 2) * k + spread * torch.randn(B, Lq, M, L, P, 2, generator=g)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().view(1, 1, 1, L, 1, 2)
    loc = ref_pts + offs / wh
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    grad_out = torch.randn(B, Lq, M * D, generator=g, dtype=torch.float64)
    return shapes, value, loc, attn, grad_out


def extra_cases(ref):
    # the production head geometry (M=8, D=32, L=4, P=4) on small maps
    shapes, value, loc, attn, go = model_like(11, 2, 8, 32, [(8, 12), (4, 6), (2, 3), (1, 2)], 37, 4)
    save("model_d32", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="model_like(seed 11)")
    # ragged sizes: odd heads / channels / levels / points, Lq not a multiple of 8
    shapes, value, loc, attn, go = model_like(12, 3, 3, 20, [(7, 5), (4, 3), (1, 1)], 13, 3, spread=2.0)
    save("ragged", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="model_like(seed 12)")
    # D=64 heads (the reference's second shared-memory branch, cuh:1162-1183)
    shapes, value, loc, attn, go = model_like(13, 1, 4, 64, [(9, 11), (5, 6)], 21, 4)
    save("model_d64", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="model_like(seed 13)")
    # one level, one point, one head, one channel
    shapes, value, loc, attn, go = model_like(14, 1, 1, 1, [(2, 2)], 1, 1)
    loc = torch.tensor([0.37, 0.61]).view(1, 1, 1, 1, 1, 2)
    save("minimal", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="model_like(seed 14), loc=(0.37,0.61)")
    # every sample outside the maps: output and all gradients are exactly zero
    shapes, value, loc, attn, go = model_like(15, 1, 2, 8, [(4, 4), (2, 2)], 5, 2)
    loc = loc.abs() + 1.5
    save("all_outside", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="model_like(seed 15), loc=|loc|+1.5")
    # borders: samples within one pixel outside each edge, where 1-3 taps are missing
    # (cuh:55-78).  Forward only plus gradients: coordinates avoid exact integers.
    g = torch.Generator().manual_seed(16)
    shapes = torch.as_tensor([(5, 7), (3, 4)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    B, M, D, Lq, L, P = 1, 2, 4, 16, 2, 4
    value = torch.randn(B, S, M, D, generator=g)
    edge = torch.rand(B, Lq, M, L, P, 2, generator=g)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().view(1, 1, 1, L, 1, 2)
    # pixel coordinate in (-1, 0) or (size-1, size): loc = (pix + 0.5) / size
    lowside = torch.rand(B, Lq, M, L, P, 2, generator=g) < 0.5
    pix = torch.where(lowside, -0.95 + 0.9 * edge, wh - 1 + 0.05 + 0.9 * edge)
    loc = (pix + 0.5) / wh
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    go = torch.randn(B, Lq, M * D, generator=g, dtype=torch.float64)
    save("borders", shapes, value, loc, attn, run_reference(ref, value, shapes, loc, attn, go),
         recipe="border ring, seed 16")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.set_num_threads(1)
    ref = load_reference()
    reference_test_py_cases(ref)
    extra_cases(ref)


if __name__ == "__main__":
    main()
