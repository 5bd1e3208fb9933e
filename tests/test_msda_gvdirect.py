"""grad_value of calls below 1 024 queries (the decoders'): the self-decoding kernel of vnext_amd/csrc/msda_d32_gvdirect.hip,
launched after the grad_loc kernel on the caller's stream or, with VNX_MSDA_FORK, beside it on the library's side stream
(capi.hip: side_lane).  Everything through
the C ABI against the fp64 C oracle: the BASELINE decoder shapes, ragged query counts incl. several passes (> 304
queries), levels of a handful of pixels (a row spread over several 8-lane groups), every sample on one spot (one row's
segment is the whole list), other level / point counts, 16-bit values, unpacked levels (the general path must
take over on the device), the fork under stream capture, on a non-default stream, from two host threads, and against
the single-stream form and the record-fed kernels of rounds 1-4.
Reference semantics: ms_deform_im2col_cuda.cuh:87-159 (scatter), :253-298 (decode)."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from oracle import msda_oracle as O  # noqa: E402
from vnext_amd import _lib  # noqa: E402
from test_msda_gvtiles import check, oracle, pixel_queries, run, scale  # noqa: E402

DEV = "cuda:0"
P360 = [(48, 80), (24, 40), (12, 20), (6, 10)]
P720 = [(92, 160), (46, 80), (23, 40), (12, 20)]
TINY = [(5, 7), (3, 4), (2, 2), (1, 1)]


def uniform_case(shapes, B, Lq, seed, M=8, P=4, spread=1.2, centre=None):
    """the reference test's convention (ops/test.py:33-36): locations uniform, here over a square a little larger than the map"""
    g = torch.Generator().manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int(sh.prod(1).sum())
    lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, M, 32, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * spread - (spread - 1) / 2
    if centre is not None:
        loc = centre + 0.004 * torch.rand(B, Lq, M, L, P, 2, generator=g)
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P).contiguous()
    go = torch.randn(B, Lq, M * 32, generator=g)
    return sh, lsi, value, loc.contiguous(), attn, go


@pytest.mark.parametrize("shapes,B,Lq", [(P360, 5, 300), (P360, 10, 300), (P720, 2, 300), (P360, 1, 100)])
@pytest.mark.parametrize("dist", ["U", "M"])
def test_baseline_decoder_shapes(shapes, B, Lq, dist):
    case = uniform_case(shapes, B, Lq, seed=B + Lq) if dist == "U" else pixel_queries(shapes, B, Lq, seed=B + Lq, far=0.05)
    check(run(case, 0), oracle(case), case)


@pytest.mark.parametrize("Lq", [1, 2, 7, 63, 64, 127, 303, 304, 305, 319, 320, 321, 608, 640, 700, 1023])
def test_ragged_query_counts_and_several_passes(Lq):
    case = uniform_case([(24, 40), (12, 20), (6, 10), (3, 5)], 2, Lq, seed=Lq)
    check(run(case, 0), oracle(case), case)


@pytest.mark.parametrize("shapes", [TINY, [(1, 1)] * 4, [(2, 33), (1, 70), (65, 1), (3, 3)], [(20, 16)], [(19, 17), (300, 2)]])
def test_small_and_odd_levels(shapes):
    """levels of a few pixels: more units than 320-row pieces, a row on 2 / 4 / 8 groups; one-pixel, one-row, one-column levels"""
    case = uniform_case(shapes, 3, 90, seed=len(shapes) * 7 + shapes[0][1])
    check(run(case, 0), oracle(case), case)


@pytest.mark.parametrize("Lq", [300, 1000])
def test_every_sample_on_one_spot(Lq):
    """4 x points x queries taps on four pixels of every level: the sorted list holds the worst case, one row's segment is
    every tap of the pass"""
    case = uniform_case(P360, 2, Lq, seed=3, centre=0.37)
    check(run(case, 0), oracle(case), case, tol=4e-5)


@pytest.mark.parametrize("L,P", [(3, 2), (2, 8), (1, 3), (4, 1), (5, 5), (2, 16)])
def test_other_level_and_point_counts(L, P):
    shapes = [(30, 44), (15, 22), (8, 11), (4, 6), (2, 3)][:L]
    case = uniform_case(shapes, 2, 150, seed=10 * L + P, M=4, P=P)
    check(run(case, 0), oracle(case), case)


@pytest.mark.parametrize("L,P", [(5, 4), (4, 8), (3, 2)])
@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16])
def test_many_queries_with_other_level_and_point_counts(L, P, vdt):
    """from 1 024 queries up a call the tile-fed kernel is not built for (L * P != 16 or P != 4) must NOT take the self-decoding
    kernel (one pass per 304 queries, 16-bit rows rounded once per pass): the record-fed path accumulates in fp32 (capi.hip:
    use_direct)"""
    shapes = [(30, 44), (15, 22), (8, 11), (4, 6), (2, 3)][:L]
    case = uniform_case(shapes, 2, 1500, seed=100 * L + P, M=4, P=P)
    got = run(case, 0, vdt)
    want = oracle(case, vdt)
    if vdt == torch.float32:
        check(got, want, case)
    else:
        np.testing.assert_allclose(got[0], want[0], rtol=0, atol=8e-3 * scale(want[0]))
        np.testing.assert_allclose(got[2], want[2], rtol=0, atol=8e-3 * scale(want[2]))


@pytest.mark.parametrize("vdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("loc16", [False, True])
def test_sixteen_bit_values(vdt, loc16):
    case = uniform_case(P360, 2, 300, seed=21)
    ldt = vdt if loc16 else torch.float32
    got = run(case, 0, vdt, ldt)
    want = oracle(case, vdt, ldt)
    tol = 8e-3 if vdt == torch.bfloat16 else 2e-3
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=tol * scale(want[0]))
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=(3e-2 if loc16 else tol) * scale(want[2]))


def test_unpacked_levels_fall_to_the_general_path():
    """level_start_index with a gap (and the promise not given): the fast kernels must do nothing on the device and the
    general kernels everything -- and with packed levels, promise not given, the fast ones everything."""
    sh, lsi, value, loc, attn, go = uniform_case([(12, 20), (6, 10), (3, 5)], 2, 200, seed=5, M=4)
    check(run((sh, lsi, value, loc, attn, go), 0, packed=False), oracle((sh, lsi, value, loc, attn, go)), (sh, lsi, value, loc, attn, go))
    gap = 13
    S = value.shape[1]
    lsi2 = lsi.clone(); lsi2[1:] += gap
    value2 = torch.randn(2, S + gap, 4, 32)
    case2 = (sh, lsi2, value2, loc, attn, go)
    got = run(case2, 0, packed=False)
    want = oracle(case2)
    check(got, want, case2)
    assert np.all(got[0][:, 240:240 + gap] == 0)          # the pixels between the levels receive nothing


def raw_backward(case, flags, stream=None):
    """the C ABI called directly (flags are not part of the reference signature)"""
    sh, lsi, value, loc, attn, go = [t.to(DEV) for t in case]
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn)
    lib = _lib.lib()
    n = lib.vnx_msda_backward_workspace_bytes(0, 0, B, S, M, D, L, Lq, P, flags)
    assert n == 0, "a call below 1 024 queries with packed levels needs no workspace (include/vnext_hip.h)"
    s = stream if stream is not None else torch.cuda.current_stream()
    st = lib.vnx_msda_backward(0, 0, value.data_ptr(), sh.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), go.data_ptr(),
                               gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, L, Lq, P, flags, None, 0, s.cuda_stream)
    _lib.check(st)
    return gv, gl, ga


def test_fork_flag_and_record_kernels_agree():
    case = uniform_case(P360, 5, 300, seed=8)
    forked = [t.cpu().numpy() for t in raw_backward(case, _lib.MSDA_LEVELS_PACKED | _lib.MSDA_FORK)]
    torch.cuda.synchronize()
    serial = [t.cpu().numpy() for t in raw_backward(case, _lib.MSDA_LEVELS_PACKED)]
    torch.cuda.synchronize()
    assert np.array_equal(forked[1], serial[1]) and np.array_equal(forked[2], serial[2])      # grad_loc / grad_attn: same kernel
    np.testing.assert_allclose(forked[0], serial[0], rtol=0, atol=3e-6 * scale(serial[0]))   # grad_value: order of a row's taps
    records = run(case, 430)                                                                  # rounds 1-4: records + per-unit selection
    np.testing.assert_allclose(forked[0], records[0], rtol=0, atol=3e-6 * scale(records[0]))
    assert np.array_equal(forked[1], records[1]) and np.array_equal(forked[2], records[2])
    dev_fork = run(case, 441)                                                                 # development build: the fork by variant
    np.testing.assert_allclose(forked[0], dev_fork[0], rtol=0, atol=3e-6 * scale(dev_fork[0]))


def test_fork_under_stream_capture_and_replay():
    """torch.cuda.graph captures in global mode: the side stream must already exist or be created in relaxed mode, the
    capture must end with the side stream joined, and replays on fresh inputs must give fresh results"""
    case = uniform_case(P360, 5, 300, seed=31)
    sh, lsi, value, loc, attn, go = [t.to(DEV) for t in case]
    B, S, M, D = value.shape
    Lq = loc.shape[1]
    gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn)
    lib = _lib.lib()

    def call():
        st = lib.vnx_msda_backward(0, 0, value.data_ptr(), sh.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(),
                                   go.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, 4, Lq, 4,
                                   _lib.MSDA_LEVELS_PACKED | _lib.MSDA_FORK, None, 0, torch.cuda.current_stream().cuda_stream)
        _lib.check(st)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
        call()
    for seed in (32, 33):
        fresh = uniform_case(P360, 5, 300, seed=seed)
        for dst, src in zip((value, loc, attn, go), fresh[2:]):
            dst.copy_(src)
        gv.fill_(float("nan")); gl.fill_(float("nan")); ga.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        check([t.double().cpu().numpy() for t in (gv, gl, ga)], oracle(fresh), fresh)


def test_first_call_of_a_process_under_capture():
    """the same in a fresh process, where the capture's first call is the one that creates the side stream"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--capture-first"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "capture-first ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_on_a_side_stream_back_to_back():
    """twenty calls in a row on a non-default stream, inputs alternating, results consumed on that stream right away: the
    join must order every call's grad_value before the next operation of the caller's stream"""
    cases = [uniform_case(P360, 2, 300, seed=40 + i) for i in range(2)]
    wants = [oracle(c)[0] for c in cases]
    stream = torch.cuda.Stream()
    sums = []
    with torch.cuda.stream(stream):
        dev_cases = [[t.to(DEV) for t in c] for c in cases]
        stream.synchronize()
        for i in range(20):
            gv, gl, ga = raw_backward(dev_cases[i % 2], _lib.MSDA_LEVELS_PACKED | (_lib.MSDA_FORK if i % 4 < 2 else 0), stream)
            sums.append(gv.double().abs().sum())        # consumed on `stream` with no host synchronisation in between
        stream.synchronize()
    for i, s in enumerate(sums):
        want = np.abs(wants[i % 2]).sum()
        assert abs(float(s) - want) <= 1e-5 * want


def test_two_host_threads():
    """one side lane per host thread: two threads calling at once, each on its own stream, do not share events"""
    cases = [uniform_case(P360, 2, 300, seed=50 + i) for i in range(2)]
    wants = [oracle(c) for c in cases]
    out = [None, None]

    def work(i):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            dev = [t.to(DEV) for t in cases[i]]
            for _ in range(10):
                res = raw_backward(dev, _lib.MSDA_LEVELS_PACKED | _lib.MSDA_FORK, s)
            s.synchronize()
            out[i] = [t.double().cpu().numpy() for t in res]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(2):
        check(out[i], wants[i], cases[i])


def test_autograd_function_takes_the_path():
    """MSDeformAttnFunction.backward on a decoder-shape call (what a model's decoder layer does)"""
    from vnext_amd.ops.functions import MSDeformAttnFunction
    case = uniform_case(P360, 2, 300, seed=61)
    sh, lsi, value, loc, attn, go = case
    leaves = [value.to(DEV).requires_grad_(True), loc.to(DEV).requires_grad_(True), attn.to(DEV).requires_grad_(True)]
    out = MSDeformAttnFunction.apply(leaves[0], sh.to(DEV), lsi.to(DEV), leaves[1], leaves[2], 64)
    out.backward(go.to(DEV))
    torch.cuda.synchronize()
    check([t.grad.double().cpu().numpy() for t in leaves], oracle(case), case)


if __name__ == "__main__" and "--capture-first" in sys.argv:
    test_fork_under_stream_capture_and_replay()
    print("capture-first ok")
