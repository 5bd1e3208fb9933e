"""GPU parity tests of the MSDeformAttn HIP path (run with -m gpu on an MI355X).

Everything goes through the product boundary: MultiScaleDeformableAttention ->
vnext_amd.msda_ext -> ctypes -> libvnext_hip.so.  The checker is the CPU oracle
(oracle/), itself pinned to the reference by tests/test_oracle.py, plus the
golden vectors generated from the reference (tests/golden).

Tolerances: fp64 1e-10 relative (order of summation differs from the reference's
fp32 atomics anyway); fp32 well inside the reference's own rtol 1e-2 / atol 1e-3
(ops/test.py:56) -- we assert 1e-5 of the output scale; bf16/fp16 within 1e-2 of
the scale, the north-star bound.
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import make_golden
from oracle import msda_oracle as O

pytestmark = pytest.mark.gpu

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from vnext_amd import _lib  # noqa: E402
from vnext_amd.ops.functions import MSDeformAttnFunction  # noqa: E402

DEV = "cuda:0"
NAMES = golden_names()
GRAD_NAMES = [n for n in NAMES if "fwd" not in n]


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def scale_of(x):
    return max(1e-30, float(np.abs(x).max()))


@pytest.fixture(autouse=True)
def _auto_variant():
    _lib.set_kernel_variant(0)
    yield
    _lib.set_kernel_variant(0)


def run_fwd(g, dtype, loc_dtype=None, variant=0):
    _lib.set_kernel_variant(variant)
    out = MSDA.ms_deform_attn_forward(dev(g["value"], dtype), dev(g["shapes"]), dev(g["lsi"]),
                                      dev(g["loc"], loc_dtype or dtype),
                                      dev(g["attn"], loc_dtype or dtype), 64)
    torch.cuda.synchronize()
    return out.double().cpu().numpy()


def run_bwd(g, dtype, loc_dtype=None, variant=0):
    _lib.set_kernel_variant(variant)
    gv, gl, ga = MSDA.ms_deform_attn_backward(
        dev(g["value"], dtype), dev(g["shapes"]), dev(g["lsi"]), dev(g["loc"], loc_dtype or dtype),
        dev(g["attn"], loc_dtype or dtype), dev(g["grad_out"], dtype), 64)
    torch.cuda.synchronize()
    return gv.double().cpu().numpy(), gl.double().cpu().numpy(), ga.double().cpu().numpy()


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", NAMES)
def test_forward_golden_f64(name):
    g = load_golden(name)
    out = run_fwd(g, torch.float64)
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-10, atol=1e-13 * scale_of(g["out_f64"]) + 1e-300)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", NAMES)
def test_forward_golden_f32(name, variant):
    g = load_golden(name)
    out = run_fwd(g, torch.float32, variant=variant)
    s = scale_of(g["out_f64"])
    np.testing.assert_allclose(out, g["out_f64"], rtol=0, atol=1e-5 * s)
    # and the reference's own criterion (ops/test.py:56)
    assert np.allclose(out, g["out_f32"], rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("name", GRAD_NAMES)
def test_backward_golden_f64(name):
    g = load_golden(name)
    gv, gl, ga = run_bwd(g, torch.float64)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-9, atol=1e-12 * scale_of(g["grad_loc"]))
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-9, atol=1e-12 * scale_of(g["grad_attn"]))
    if "grad_value" in g:
        np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-9, atol=1e-12 * scale_of(g["grad_value"]))
    else:
        s = scale_of(g["grad_value_head"])
        np.testing.assert_allclose(gv[..., :40], g["grad_value_head"], rtol=1e-9, atol=1e-12 * s)
        np.testing.assert_allclose(gv[..., -40:], g["grad_value_tail"], rtol=1e-9, atol=1e-12 * s)
        np.testing.assert_allclose(gv.sum(-1), g["grad_value_rowsum"], rtol=1e-8, atol=1e-10 * s)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", [n for n in GRAD_NAMES if "d2048" not in n and "d3096" not in n])
def test_backward_golden_f32(name, variant):
    g = load_golden(name)
    if "grad_value" not in g:
        pytest.skip("digest-only case is covered in fp64")
    gv, gl, ga = run_bwd(g, torch.float32, variant=variant)
    np.testing.assert_allclose(gv, g["grad_value"], rtol=0, atol=2e-5 * scale_of(g["grad_value"]))
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=0, atol=2e-5 * scale_of(g["grad_loc"]))
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=0, atol=2e-5 * scale_of(g["grad_attn"]))


# ------------------------------------------------- the reference's own test, ported
def _testpy_case(name):
    for nm, shapes, value, loc, attn, go in make_golden.testpy_draws():
        if nm == name:
            lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
            return shapes.to(DEV), lsi.to(DEV), value.to(DEV), loc.to(DEV), attn.to(DEV)
    raise KeyError(name)


def test_reference_check_forward_equal_with_pytorch_double():
    """ops/test.py:31-44 with the oracle standing in for ms_deform_attn_core_pytorch."""
    shapes, lsi, value, loc, attn = _testpy_case("testpy_fwd_double")
    out = MSDeformAttnFunction.apply(value.double(), shapes, lsi, loc.double(), attn.double(), 2)
    g = load_golden("testpy_fwd_double")
    assert torch.allclose(out.cpu(), torch.from_numpy(g["out_f64"]))


def test_reference_check_forward_equal_with_pytorch_float():
    """ops/test.py:47-60: rtol=1e-2, atol=1e-3."""
    shapes, lsi, value, loc, attn = _testpy_case("testpy_fwd_float")
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 2)
    g = load_golden("testpy_fwd_float")
    assert torch.allclose(out.cpu(), torch.from_numpy(g["out_f32"]), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("channels", [30, 32, 64, 71])
def test_reference_check_gradient_numerical(channels):
    """ops/test.py:63-78: torch.autograd.gradcheck in double through the Function."""
    shapes, lsi, value, loc, attn = _testpy_case(f"testpy_grad_d{channels}")
    value = value.double().requires_grad_(True)
    loc = loc.double().requires_grad_(True)
    attn = attn.double().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, attn, 2))


def test_gradcheck_wide_channels_sampled():
    """D=1025 (the reference's odd wide branch): numerical gradient on a sample of inputs."""
    shapes, lsi, value, loc, attn = _testpy_case("testpy_grad_d1025")
    value, loc, attn = value.double(), loc.double(), attn.double()
    g = torch.Generator().manual_seed(0)
    go = torch.randn(1, 2, 2 * 1025, generator=g, dtype=torch.float64).to(DEV)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 2)

    def f(v, s, a):
        return float((MSDA.ms_deform_attn_forward(v, shapes, lsi, s, a, 2) * go).sum())

    eps = 1e-6
    for which, (arr, grad) in enumerate(((value, gv), (loc, gl), (attn, ga))):
        flat = arr.reshape(-1)
        for i in torch.randint(0, flat.numel(), (8,), generator=g).tolist():
            args = [value.clone(), loc.clone(), attn.clone()]
            args[which].view(-1)[i] += eps
            up = f(*args)
            args[which].view(-1)[i] -= 2 * eps
            dn = f(*args)
            num = (up - dn) / (2 * eps)
            assert abs(num - float(grad.reshape(-1)[i])) <= 1e-6 + 1e-5 * abs(num)


# --------------------------------------------------------- seeded cases vs the oracle
def make_case(seed, B, M, D, shapes_list, Lq, P, model_like=True, spread=1.0):
    shapes, value, loc, attn, go = make_golden.model_like(seed, B, M, D, shapes_list, Lq, P, spread)
    if not model_like:
        g = torch.Generator().manual_seed(seed + 1)
        loc = torch.rand(loc.shape, generator=g)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return {"shapes": shapes.numpy(), "lsi": lsi.numpy(), "value": value.numpy(),
            "loc": loc.numpy(), "attn": attn.numpy(), "grad_out": go.numpy()}


PYRAMID = [(24, 40), (12, 20), (6, 10), (3, 5)]
FWD_VARIANTS = [0, 1, 2, 3, 4, 5, 12, 13, 14, 15]


@pytest.mark.parametrize("variant", FWD_VARIANTS)
@pytest.mark.parametrize("Lq,uniform", [(300, True), (37, False), (1, False), (1275, False)])
def test_forward_every_kernel_variant_f32(variant, Lq, uniform):
    g = make_case(100 + Lq, 3, 8, 32, PYRAMID, Lq, 4, model_like=not uniform)
    ref = O.msda_forward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                         g["loc"].astype(np.float64), g["attn"].astype(np.float64), nthreads=4)
    out = run_fwd(g, torch.float32, variant=variant)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5 * scale_of(ref))


@pytest.mark.parametrize("M,L,P", [(8, 4, 4), (4, 2, 3), (1, 1, 1), (16, 5, 8), (3, 3, 5)])
def test_forward_d32_other_geometries(M, L, P):
    shapes = [(9, 14), (7, 5), (4, 4), (2, 3), (1, 2)][:L]
    g = make_case(7 + M, 2, M, 32, shapes, 53, P)
    ref = O.msda_forward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                         g["loc"].astype(np.float64), g["attn"].astype(np.float64))
    for variant in (0, 1, 2, 5):
        out = run_fwd(g, torch.float32, variant=variant)
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5 * scale_of(ref))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("loc_fp32", [True, False])
@pytest.mark.parametrize("variant", [0, 1])
def test_forward_16bit(dtype, loc_fp32, variant):
    g = make_case(21, 2, 8, 32, PYRAMID, 150, 4)
    # quantise the inputs first so the comparison isolates the kernel's arithmetic
    vq = torch.from_numpy(g["value"]).to(dtype).double().numpy()
    ldt = torch.float32 if loc_fp32 else dtype
    lq = torch.from_numpy(g["loc"]).to(ldt).double().numpy()
    aq = torch.from_numpy(g["attn"]).to(ldt).double().numpy()
    ref = O.msda_forward(vq, g["shapes"], g["lsi"], lq, aq)
    out = run_fwd(g, dtype, loc_dtype=ldt, variant=variant)
    # output rounding to a 16-bit type: 2^-8 relative for bf16; north star asks 1e-2
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-2 * scale_of(ref))
    if loc_fp32:
        tight = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
        assert np.abs(out - ref).max() <= tight * scale_of(ref)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5, 12, 15, 300, 303, 420, 425])   # 420/425: the other grad_value kernels
@pytest.mark.parametrize("Lq,uniform", [(300, True), (37, False), (1, False)])
def test_backward_f32_vs_oracle(variant, Lq, uniform):
    g = make_case(200 + Lq, 3, 8, 32, PYRAMID, Lq, 4, model_like=not uniform)
    rv, rl, ra = O.msda_backward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                                 g["loc"].astype(np.float64), g["attn"].astype(np.float64),
                                 g["grad_out"], nthreads=4)
    gv, gl, ga = run_bwd(g, torch.float32, variant=variant)
    np.testing.assert_allclose(gv, rv, rtol=0, atol=2e-5 * scale_of(rv))
    np.testing.assert_allclose(gl, rl, rtol=0, atol=2e-5 * scale_of(rl))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=2e-5 * scale_of(ra))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("variant", [0, 1])
def test_backward_16bit(dtype, variant):
    g = make_case(31, 2, 8, 32, PYRAMID, 150, 4)
    vq = torch.from_numpy(g["value"]).to(dtype).double().numpy()
    goq = torch.from_numpy(g["grad_out"]).to(dtype).double().numpy()
    rv, rl, ra = O.msda_backward(vq, g["shapes"], g["lsi"], g["loc"].astype(np.float64),
                                 g["attn"].astype(np.float64), goq, nthreads=4)
    gq = dict(g, grad_out=goq)
    gv, gl, ga = run_bwd(gq, dtype, loc_dtype=torch.float32, variant=variant)
    np.testing.assert_allclose(gv, rv, rtol=0, atol=1e-2 * scale_of(rv))
    np.testing.assert_allclose(gl, rl, rtol=0, atol=1e-4 * scale_of(rl))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=1e-4 * scale_of(ra))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_levels_packed_promise_changes_nothing(dtype):
    """The packed-levels promise only removes launches; values are the same kernels' output."""
    g = make_case(33, 2, 8, 32, PYRAMID, 77, 4)
    ldt = torch.float32
    args = (dev(g["value"], dtype), dev(g["shapes"]), dev(g["lsi"]), dev(g["loc"], ldt),
            dev(g["attn"], ldt), dev(g["grad_out"], dtype), 64)
    a = MSDA.ms_deform_attn_backward(*args)
    b = MSDA.ms_deform_attn_backward(*args, levels_packed=True)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for x, y in zip(a, b):
        s = float(x.float().abs().max())
        assert float((x.float() - y.float()).abs().max()) <= tol * s


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_unpacked_levels_take_the_general_path(dtype):
    """Levels stored out of order with gaps: the reference semantics (any
    level_start_index) must survive; the owner-computes kernels stand down on the device."""
    g = make_case(35, 2, 8, 32, [(6, 9), (4, 5), (2, 3)], 41, 4)
    sizes = [54, 20, 6]
    lsi = np.array([40, 100, 3], dtype=np.int64)       # level 2 first, gaps everywhere
    S = 160
    rng = np.random.default_rng(0)
    value = rng.standard_normal((2, S, 8, 32)).astype(np.float32)
    g = dict(g, lsi=lsi, value=value)
    vq = torch.from_numpy(value).to(dtype).double().numpy()
    goq = torch.from_numpy(g["grad_out"]).to(dtype).double().numpy()
    ref = O.msda_forward(vq, g["shapes"], lsi, g["loc"].astype(np.float64), g["attn"].astype(np.float64))
    rv, rl, ra = O.msda_backward(vq, g["shapes"], lsi, g["loc"].astype(np.float64),
                                 g["attn"].astype(np.float64), goq)
    out = run_fwd(g, dtype, loc_dtype=torch.float32)
    gv, gl, ga = run_bwd(dict(g, grad_out=goq), dtype, loc_dtype=torch.float32)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(out, ref, rtol=0, atol=tol * scale_of(ref))
    np.testing.assert_allclose(gv, rv, rtol=0, atol=tol * scale_of(rv))
    np.testing.assert_allclose(gl, rl, rtol=0, atol=tol * scale_of(rl))
    np.testing.assert_allclose(ga, ra, rtol=0, atol=tol * scale_of(ra))
    used = np.zeros(S, bool)
    for s0, n in zip(lsi, sizes):
        used[s0:s0 + n] = True
    assert np.all(gv[:, ~used] == 0)


@pytest.mark.parametrize("units_min", [1, 2, 8, 16])
def test_backward_owner_units_split(units_min):
    """grad_value does not depend on how the levels are cut into owner units."""
    g = make_case(37, 2, 8, 32, [(31, 33), (16, 17), (8, 9), (1, 1)], 90, 4, spread=3.0)
    rv, rl, ra = O.msda_backward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                                 g["loc"].astype(np.float64), g["attn"].astype(np.float64),
                                 g["grad_out"], nthreads=4)
    gv, gl, ga = run_bwd(g, torch.float32, variant=200 + units_min)
    np.testing.assert_allclose(gv, rv, rtol=0, atol=2e-5 * scale_of(rv))
    np.testing.assert_allclose(gl, rl, rtol=0, atol=2e-5 * scale_of(rl))


# ------------------------------------------------------------------ edge cases
def test_empty_batch_and_empty_queries():
    shapes = torch.tensor([[4, 4], [2, 2]], device=DEV)
    lsi = torch.tensor([0, 16], device=DEV)
    for B, Lq in ((0, 5), (2, 0)):
        v = torch.randn(B, 20, 2, 32, device=DEV)
        loc = torch.rand(B, Lq, 2, 2, 3, 2, device=DEV)
        attn = torch.rand(B, Lq, 2, 2, 3, device=DEV)
        out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
        assert out.shape == (B, Lq, 64)
        gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, torch.zeros_like(out), 64)
        assert gv.shape == v.shape and gl.shape == loc.shape and ga.shape == attn.shape
        assert float(gv.abs().sum()) == 0.0


def test_outputs_need_no_prefill_and_inputs_are_untouched():
    g = make_case(41, 2, 8, 32, PYRAMID, 64, 4)
    v, s, i = dev(g["value"]), dev(g["shapes"]), dev(g["lsi"])
    loc, attn = dev(g["loc"]), dev(g["attn"])
    copies = [t.clone() for t in (v, s, i, loc, attn)]
    # poison the caching allocator's free blocks so stale data would show
    junk = torch.full((4 << 20,), float("nan"), device=DEV)
    del junk
    out1 = MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 64)
    junk = torch.full((4 << 20,), float("nan"), device=DEV)
    del junk
    go = dev(g["grad_out"], torch.float32)
    gv1, gl1, ga1 = MSDA.ms_deform_attn_backward(v, s, i, loc, attn, go, 64)
    assert torch.isfinite(out1).all() and torch.isfinite(gv1).all()
    assert torch.isfinite(gl1).all() and torch.isfinite(ga1).all()
    for t, c in zip((v, s, i, loc, attn), copies):
        assert torch.equal(t, c)


def test_nan_location_contributes_nothing():
    """A NaN coordinate fails the reference's range test (cuh:288) and is skipped."""
    g = make_case(43, 1, 8, 32, PYRAMID, 16, 4)
    g["loc"][0, 3, 2, 1, 2, 0] = np.nan
    clean = dict(g, attn=g["attn"].copy())
    clean["attn"][0, 3, 2, 1, 2] = 0.0
    clean["loc"] = np.nan_to_num(g["loc"], nan=0.5)
    for variant in (0, 1):
        out = run_fwd(g, torch.float32, variant=variant)
        ref = run_fwd(clean, torch.float32, variant=variant)
        assert np.isfinite(out).all()
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6 * scale_of(ref))


def test_argument_checks_match_the_reference():
    g = make_case(45, 3, 8, 32, PYRAMID, 8, 4)
    v, s, i = dev(g["value"]), dev(g["shapes"]), dev(g["lsi"])
    loc, attn = dev(g["loc"]), dev(g["attn"])
    with pytest.raises(RuntimeError, match=r"batch\(3\) must divide im2col_step\(2\)"):
        MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 2)  # ms_deform_attn_cuda.cu:52
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        MSDA.ms_deform_attn_forward(v.transpose(1, 2).contiguous().transpose(1, 2), s, i, loc, attn, 64)
    with pytest.raises(RuntimeError, match="spatial_shapes must be a CUDA tensor"):
        MSDA.ms_deform_attn_forward(v, s.cpu(), i, loc, attn, 64)
    with pytest.raises(RuntimeError, match="int64"):
        MSDA.ms_deform_attn_forward(v, s.int(), i, loc, attn, 64)
    out = MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 3)
    assert out.shape == (3, 8, 256)


def test_runs_on_the_callers_stream_and_in_a_graph():
    g = make_case(47, 2, 8, 32, PYRAMID, 96, 4)
    v, s, i = dev(g["value"]), dev(g["shapes"]), dev(g["lsi"])
    loc, attn = dev(g["loc"]), dev(g["attn"])
    expect = MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 64)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 64)
    side.synchronize()
    assert torch.equal(out, expect)
    # no allocation / sync inside the C call: capturable
    static_out = torch.empty_like(expect)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out.copy_(MSDA.ms_deform_attn_forward(v, s, i, loc, attn, 64))
    static_out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, expect)


def test_autograd_function_contract():
    """(gv, None, None, gloc, gattn, None) and once-differentiable, func.py:30-39."""
    g = make_case(49, 2, 8, 32, PYRAMID, 20, 4)
    v = dev(g["value"]).requires_grad_(True)
    loc = dev(g["loc"]).requires_grad_(True)
    attn = dev(g["attn"]).requires_grad_(True)
    s, i = dev(g["shapes"]), dev(g["lsi"])
    out = MSDeformAttnFunction.apply(v, s, i, loc, attn, 64)
    assert out.shape == (2, 20, 256)
    go = dev(g["grad_out"], torch.float32)
    # non-contiguous upstream gradient is accepted (func.py:34)
    (out * go).sum().backward()
    rv, rl, ra = O.msda_backward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                                 g["loc"].astype(np.float64), g["attn"].astype(np.float64),
                                 g["grad_out"])
    np.testing.assert_allclose(v.grad.cpu().numpy(), rv, rtol=0, atol=2e-5 * scale_of(rv))
    np.testing.assert_allclose(loc.grad.cpu().numpy(), rl, rtol=0, atol=2e-5 * scale_of(rl))
    np.testing.assert_allclose(attn.grad.cpu().numpy(), ra, rtol=0, atol=2e-5 * scale_of(ra))
    assert s.grad is None and i.grad is None


# ------------------------------------------- BASELINE sizes: size-independent properties
SHAPES_360P = [(48, 80), (24, 40), (12, 20), (6, 10)]    # S = 5100
SHAPES_720P = [(92, 160), (46, 80), (23, 40), (12, 20)]  # S = 19560


def _full_case(shapes_list, B, Lq, seed, uniform):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    shapes = torch.tensor(shapes_list, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, 8, 32, generator=gen)
    if uniform:
        loc = torch.rand(B, Lq, 8, 4, 4, 2, generator=gen)
    else:
        ref = torch.rand(B, Lq, 1, 1, 1, 2, generator=gen)
        wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float().view(1, 1, 1, 4, 1, 2)
        loc = ref + 3.0 * torch.randn(B, Lq, 8, 4, 4, 2, generator=gen) / wh
    attn = torch.softmax(torch.randn(B, Lq, 8, 16, generator=gen), -1).view(B, Lq, 8, 4, 4)
    return shapes, lsi, value, loc, attn


@pytest.mark.parametrize("shapes_list,Lq,uniform", [
    (SHAPES_360P, 300, True),        # headline decoder shape
    (SHAPES_360P, 5100, False),      # encoder shape, 360p
    (SHAPES_720P, 300, False),       # decoder shape, 720p
])
def test_full_size_forward_rows_and_linearity(shapes_list, Lq, uniform):
    B = 5
    shapes, lsi, value, loc, attn = _full_case(shapes_list, B, Lq, 3, uniform)
    dv, ds, di, dl, da = (t.to(DEV) for t in (value, shapes, lsi, loc, attn))
    out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    # (1) every query is independent: a random subset of rows against the oracle
    rows = torch.randperm(Lq, generator=torch.Generator().manual_seed(1))[:48].sort().values
    ref = O.msda_forward(value.double().numpy(), shapes.numpy(), lsi.numpy(),
                         loc[:, rows].double().numpy(), attn[:, rows].double().numpy(), nthreads=8)
    got = out[:, rows.to(DEV)].double().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5 * scale_of(ref))
    # (2) linear in value: f(2a - 3b) = 2 f(a) - 3 f(b)
    other = torch.randn(value.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    fo = MSDA.ms_deform_attn_forward(other, ds, di, dl, da, 64)
    mix = MSDA.ms_deform_attn_forward(2 * dv - 3 * other, ds, di, dl, da, 64)
    lin = 2 * out - 3 * fo
    assert float((mix - lin).abs().max()) <= 2e-5 * float(lin.abs().max())
    # (3) constant maps: output = constant * (attention mass of the taps that exist)
    ones = torch.ones_like(dv)
    mass = MSDA.ms_deform_attn_forward(ones, ds, di, dl, da, 64)
    assert float(mass.max()) <= 1.0 + 1e-5 and float(mass.min()) >= -1e-6
    # (4) all kernel variants agree with each other at this size
    for variant in (1, 2, 3, 4, 5):
        _lib.set_kernel_variant(variant)
        alt = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
        assert float((alt - out).abs().max()) <= 2e-5 * float(out.abs().max())
    _lib.set_kernel_variant(0)


@pytest.mark.parametrize("shapes_list,Lq,uniform", [
    (SHAPES_360P, 300, True),
    (SHAPES_360P, 5100, False),
])
def test_full_size_backward_properties(shapes_list, Lq, uniform):
    B = 5
    shapes, lsi, value, loc, attn = _full_case(shapes_list, B, Lq, 5, uniform)
    dv, ds, di, dl, da = (t.to(DEV) for t in (value, shapes, lsi, loc, attn))
    go = torch.randn(B, Lq, 256, generator=torch.Generator().manual_seed(6))
    dgo = go.to(DEV)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dv, ds, di, dl, da, dgo, 64)
    # (1) loc / attn gradients are per-query: subset of rows against the oracle
    rows = torch.randperm(Lq, generator=torch.Generator().manual_seed(1))[:32].sort().values
    rv, rl, ra = O.msda_backward(value.double().numpy(), shapes.numpy(), lsi.numpy(),
                                 loc[:, rows].double().numpy(), attn[:, rows].double().numpy(),
                                 go[:, rows].double().numpy(), nthreads=8)
    np.testing.assert_allclose(gl[:, rows.to(DEV)].double().cpu().numpy(), rl, rtol=0, atol=2e-5 * scale_of(rl))
    np.testing.assert_allclose(ga[:, rows.to(DEV)].double().cpu().numpy(), ra, rtol=0, atol=2e-5 * scale_of(ra))
    # (2) grad_value is additive over queries: zero all other rows' upstream gradient
    masked = torch.zeros_like(dgo)
    masked[:, rows.to(DEV)] = dgo[:, rows.to(DEV)]
    gv_sub, _, _ = MSDA.ms_deform_attn_backward(dv, ds, di, dl, da, masked, 64)
    np.testing.assert_allclose(gv_sub.double().cpu().numpy(), rv, rtol=0, atol=2e-5 * scale_of(rv))
    # (3) adjoint identity: <f(v), g> = <v, grad_value(g)>
    out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    lhs = float((out.double() * dgo.double()).sum())
    rhs = float((dv.double() * gv.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # (4) the attention gradient of a sample is <g, its bilinear tap>: with attn := 1-hot it
    #     reproduces the forward, sum_k attn_k * grad_attn_k = <out, g> per (b, q, m)
    per_head = (out.view(B, Lq, 8, 32).double() * dgo.view(B, Lq, 8, 32).double()).sum(-1)
    recon = (da.double() * ga.double()).sum((-1, -2))
    assert float((per_head - recon).abs().max()) <= 1e-4 * float(per_head.abs().max())
