"""Parity at the batch the reference actually presents ("What's missing" item 3): B = 10.

* SeqFormer trains two clips x five frames per GPU (projects/SeqFormer/configs/base_ytvis.yaml:18,34: IMS_PER_BATCH 16
  on 8 GPUs, SAMPLING_FRAME_NUM 5), i.e. the decoder call of a training step is B = 10, Lq = 300 once T is folded into the
  batch.  At that size the record-fed grad_value kernel runs its multi-round template instance
  (`msda_bwd_gv_sel_kernel<float, false>`: plain loads, several rounds of workgroups) instead of the one-round, `nt`-load
  instance the B = 5 headline takes -- a different kernel as far as parity is concerned.  Forward + all three gradients,
  ALL rows, against the oracle, with uniform (the reference test's convention, ops/test.py:34) and model-like locations.
* IDOL infers 720p videos in chunks of ten frames (projects/IDOL/idol/idol.py:252-262, BATCH_INFER_LEN 10): the encoder
  call there is B = 10, Lq = S = 19 560.  Forward, fp32 and bf16 (BASELINE config 5 / config 3's dtype), ALL rows.
"""
import numpy as np
import pytest
import torch

from oracle import msda_oracle as O
from test_parity_gaps import S360, case, off_the_pixel_grid, scale
from test_parity_r3 import S720, encoder_case

DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("uniform", [True, False], ids=["U", "M"])
def test_decoder_360p_batch10_all_rows_all_gradients_fp32(uniform):
    import MultiScaleDeformableAttention as MSDA
    sh, lsi, value, loc, attn, go = case(S360, 10, 300, seed=91 + int(uniform), uniform=uniform)
    dv, ds, di, dl, da, dg = (t.to(DEV) for t in (value, sh, lsi, loc, attn, go))
    out = MSDA.ms_deform_attn_forward(dv, ds, di, dl, da, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dv, ds, di, dl, da, dg, 64)
    torch.cuda.synchronize()
    args = (value.double().numpy(), sh.numpy(), lsi.numpy(), loc.double().numpy(), attn.double().numpy())
    want = O.msda_forward(*args, nthreads=8)
    rv, rl, ra = O.msda_backward(*args, go.double().numpy(), nthreads=8)
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=0, atol=1e-5 * scale(want))
    np.testing.assert_allclose(gv.double().cpu().numpy(), rv, rtol=0, atol=2e-5 * scale(rv))
    ok = off_the_pixel_grid(loc, sh)
    assert ok.mean() > 0.999
    np.testing.assert_allclose(gl.double().cpu().numpy() * ok, rl * ok, rtol=0, atol=2e-5 * scale(rl))
    np.testing.assert_allclose(ga.double().cpu().numpy(), ra, rtol=0, atol=2e-5 * scale(ra))
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-2, atol=1e-3)          # the reference's own bar, ops/test.py:56
    # every batch element is an independent problem (cuh:255-263): the B = 10 call equals ten B = 1 calls, bit for bit
    # in the forward and in grad_loc / grad_attn (grad_value's sums are order-dependent in the last bits)
    b = 7
    one = MSDA.ms_deform_attn_forward(dv[b:b + 1].contiguous(), ds, di, dl[b:b + 1].contiguous(),
                                      da[b:b + 1].contiguous(), 64)
    assert torch.equal(one[0], out[b])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_encoder_720p_batch10_forward_all_rows(dtype):
    """IDOL's ten-frame inference chunk at 720p: 25 M output elements against the fp64 oracle on the same (rounded)
    inputs; fp32 at 3e-5 of scale (the far queries of this generator carry the fp32 rounding of x W - 0.5, see
    test_parity_r3), bf16 within the north star's 1e-2."""
    import MultiScaleDeformableAttention as MSDA
    sh, lsi, value, loc, attn, _go = encoder_case(S720, 10, seed=59)
    v_in = value.to(dtype)
    out = MSDA.ms_deform_attn_forward(v_in.to(DEV), sh.to(DEV), lsi.to(DEV), loc.to(DEV), attn.to(DEV), 64)
    torch.cuda.synchronize()
    assert out.dtype == dtype and tuple(out.shape) == (10, int(sh.prod(1).sum()), 256)
    want = O.msda_forward(v_in.double().numpy(), sh.numpy(), lsi.numpy(), loc.double().numpy(), attn.double().numpy(),
                          nthreads=16)
    tol = 3e-5 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=0, atol=tol * scale(want))
    if dtype == torch.float32:
        assert np.allclose(out.cpu().numpy(), want, rtol=1e-2, atol=1e-3)
