"""MSDeformAttn modules against outputs of the reference's own module
files (fixtures: oracle/make_golden_modules.py).  The CPU tests exercise the host logic --
frame folding, location arithmetic, return conventions -- with the op swapped for the oracle
(tests only); the GPU tests run the real thing end to end."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import msda_oracle as O
from vnext_amd.ops.modules import MSDeformAttnIDOL, MSDeformAttnSeqFormer
from vnext_amd.ops.functions import ms_deform_attn_func as func_mod

CASES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "module_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"module_{name}.npz")))


def build(name, g, device, dtype):
    C, L, M, P = (int(x) for x in g["cfg"])
    if name.startswith("idol"):
        m = MSDeformAttnIDOL(C, L, M, P)
    else:
        m = MSDeformAttnSeqFormer(C, L, M, P, mode="encode" if "encode" in name else "decode")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    assert set(sd) == set(m.state_dict()), "parameter names must match the reference module"
    m = m.to(dtype=torch.float64)  # before loading: the fixtures are fp64
    m.load_state_dict(sd)
    return m.to(device=device, dtype=dtype)


def run(name, m, g, device, dtype):
    t = lambda k, dt=dtype: torch.from_numpy(g[k]).to(device=device, dtype=dt)  # noqa: E731
    shapes, lsi = torch.from_numpy(g["shapes"]).to(device), torch.from_numpy(g["lsi"]).to(device)
    mask = torch.from_numpy(g["mask"]).to(device)
    if name.startswith("idol"):
        out, loc, attn = m(t("query"), t("ref"), t("src"), shapes, lsi, mask)
        return {"out": out, "loc": loc, "attn": attn}
    if "encode" in name:
        return {"out": m(t("query"), None, t("ref"), t("src"), shapes, lsi, mask)}
    out, out_box, loc, attn = m(t("query"), t("query_box"), t("ref"), t("src"), shapes, lsi, mask)
    return {"out": out, "out_box": out_box, "loc": loc, "attn": attn}


class _OracleOp:
    """Stands in for the HIP op on CPU (tests only)."""

    @staticmethod
    def ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step):
        out = O.msda_forward(value.detach().numpy(), shapes.numpy(), lsi.numpy(), loc.detach().numpy(),
                             attn.detach().numpy())
        return torch.from_numpy(out)


def test_fixture_set():
    assert {"idol_ref2", "idol_ref4", "seq_encode", "seq_decode_first_ref2", "seq_decode_later_ref4"} <= set(CASES)


@pytest.mark.parametrize("name", CASES)
def test_host_logic_matches_reference_module_cpu(name, monkeypatch):
    monkeypatch.setattr(func_mod, "MSDA", _OracleOp)
    g = load(name)
    m = build(name, g, "cpu", torch.float64)
    with torch.no_grad():
        got = run(name, m, g, "cpu", torch.float64)
    for k, v in got.items():
        assert tuple(v.shape) == g[k].shape, f"{k}: {tuple(v.shape)} vs {g[k].shape}"
        np.testing.assert_allclose(v.numpy(), g[k], rtol=1e-9, atol=1e-11, err_msg=k)


def test_seqformer_folds_frames_into_one_launch(monkeypatch):
    calls = []

    class Counting(_OracleOp):
        @staticmethod
        def ms_deform_attn_forward(value, *a):
            calls.append(tuple(value.shape))
            return _OracleOp.ms_deform_attn_forward(value, *a)

    monkeypatch.setattr(func_mod, "MSDA", Counting)
    g = load("seq_encode")
    m = build("seq_encode", g, "cpu", torch.float64)
    with torch.no_grad():
        run("seq_encode", m, g, "cpu", torch.float64)
    N, T = g["query"].shape[:2]
    assert len(calls) == 1 and calls[0][0] == N * T  # the reference launches T times (:107-120)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
@pytest.mark.parametrize("name", CASES)
def test_modules_on_gpu(name, dtype, tol):
    g = load(name)
    m = build(name, g, "cuda:0", dtype)
    with torch.no_grad():
        got = run(name, m, g, "cuda:0", dtype)
    for k, v in got.items():
        scale = max(1e-30, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(v.double().cpu().numpy(), g[k], rtol=0, atol=tol * scale, err_msg=k)


@pytest.mark.gpu
def test_module_backward_on_gpu_matches_cpu_autograd(monkeypatch):
    """Gradients through the whole module: GPU op + autograd vs the same module on CPU whose op
    is the grid_sample statement (differentiable; oracle/msda_torch_fallback.py)."""
    from oracle.msda_torch_fallback import msda_grid_sample
    g = load("seq_decode_later_ref4")
    m_gpu = build("seq_decode_later_ref4", g, "cuda:0", torch.float32)
    t = lambda k: torch.from_numpy(g[k]).float()  # noqa: E731
    shapes, lsi, mask = (torch.from_numpy(g[k]) for k in ("shapes", "lsi", "mask"))
    src_gpu = t("src").cuda().requires_grad_(True)
    out, out_box, _, _ = m_gpu(t("query").cuda(), t("query_box").cuda(), t("ref").cuda(), src_gpu,
                               shapes.cuda(), lsi.cuda(), mask.cuda())
    (out.sum() + 2 * out_box.sum()).backward()

    class Fn:
        @staticmethod
        def apply(value, shapes_, lsi_, loc, attn, step):
            return msda_grid_sample(value, shapes_, loc, attn)

    from vnext_amd.ops.modules import ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", Fn)
    m_cpu = build("seq_decode_later_ref4", g, "cpu", torch.float64)
    src_cpu = torch.from_numpy(g["src"]).requires_grad_(True)
    d = lambda k: torch.from_numpy(g[k])  # noqa: E731
    out_c, box_c, _, _ = m_cpu(d("query"), d("query_box"), d("ref"), src_cpu, shapes, lsi, mask)
    (out_c.sum() + 2 * box_c.sum()).backward()
    ref = src_cpu.grad.numpy()
    np.testing.assert_allclose(src_gpu.grad.double().cpu().numpy(), ref, rtol=0,
                               atol=5e-5 * float(np.abs(ref).max()))
    for (n1, p1), (n2, p2) in zip(m_gpu.named_parameters(), m_cpu.named_parameters()):
        assert n1 == n2
        if p2.grad is None:
            continue
        r = p2.grad.numpy()
        np.testing.assert_allclose(p1.grad.double().cpu().numpy(), r, rtol=0,
                                   atol=1e-4 * max(1e-12, float(np.abs(r).max())), err_msg=n1)
