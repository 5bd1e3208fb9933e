"""CPU: the oracle (oracle/msda_oracle.c) against every golden vector produced by
the reference's own function (oracle/make_golden.py), and against an independent
numpy statement.  This is what pins the checker the GPU parity tests rely on."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import make_golden
from oracle import msda_oracle as O

NAMES = golden_names()


def test_golden_set_is_complete():
    expected = {"testpy_fwd_double", "testpy_fwd_float", "model_d32", "model_d64", "ragged",
                "minimal", "all_outside", "borders"} | {f"testpy_grad_d{c}" for c in make_golden.TESTPY_CHANNELS}
    assert expected <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference_f64(name):
    g = load_golden(name)
    out = O.msda_forward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                         g["loc"].astype(np.float64), g["attn"].astype(np.float64))
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference_f32(name):
    g = load_golden(name)
    out = O.msda_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    assert out.dtype == np.float32
    scale = max(1e-30, float(np.abs(g["out_f64"]).max()))
    # the reference's own fp32 tolerance is rtol 1e-2 / atol 1e-3 (ops/test.py:56); we hold 2e-6 of scale
    np.testing.assert_allclose(out, g["out_f32"], rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(out, g["out_f64"], rtol=0, atol=2e-6 * scale)


@pytest.mark.parametrize("name", [n for n in NAMES if "fwd" not in n])
@pytest.mark.parametrize("threads", [1, 4])
def test_backward_matches_reference_autograd(name, threads):
    g = load_golden(name)
    gv, gl, ga = O.msda_backward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                                 g["loc"].astype(np.float64), g["attn"].astype(np.float64),
                                 g["grad_out"], nthreads=threads)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-10, atol=1e-13)
    if "grad_value" in g:
        np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-10, atol=1e-14)
    else:  # wide-channel cases store a digest of grad_value
        np.testing.assert_allclose(gv[..., :40], g["grad_value_head"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(gv[..., -40:], g["grad_value_tail"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(gv.sum(-1), g["grad_value_rowsum"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["testpy_fwd_double", "borders", "ragged", "minimal", "all_outside"])
def test_independent_numpy_statement_agrees(name):
    g = load_golden(name)
    ref = O.msda_numpy_small(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    out = O.msda_forward(g["value"].astype(np.float64), g["shapes"], g["lsi"],
                         g["loc"].astype(np.float64), g["attn"].astype(np.float64))
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-15)


def test_backward_is_the_gradient_of_forward():
    """Central differences on the oracle itself (fp64), as the reference's gradcheck does."""
    rng = np.random.default_rng(5)
    shapes = np.array([[5, 4], [3, 2]], dtype=np.int64)
    lsi = O.level_start_index(shapes)
    B, M, D, Lq, L, P = 2, 2, 3, 3, 2, 2
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((B, S, M, D))
    loc = rng.random((B, Lq, M, L, P, 2)) * 1.2 - 0.1
    attn = rng.random((B, Lq, M, L, P))
    go = rng.standard_normal((B, Lq, M * D))
    gv, gl, ga = O.msda_backward(value, shapes, lsi, loc, attn, go)

    def f(v, s, a):
        return float((O.msda_forward(v, shapes, lsi, s, a) * go).sum())

    eps = 1e-6
    for arr, grad, which in ((value, gv, 0), (loc, gl, 1), (attn, ga, 2)):
        idx = [tuple(rng.integers(0, n) for n in arr.shape) for _ in range(12)]
        for i in idx:
            args = [value.copy(), loc.copy(), attn.copy()]
            args[which][i] += eps
            up = f(*args)
            args[which][i] -= 2 * eps
            dn = f(*args)
            np.testing.assert_allclose((up - dn) / (2 * eps), grad[i], rtol=2e-5, atol=1e-7)


def test_threads_do_not_change_results():
    g = load_golden("model_d32")
    a = O.msda_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"], nthreads=1)
    b = O.msda_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"], nthreads=4)
    assert np.array_equal(a, b)
