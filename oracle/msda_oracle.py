"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy front-end of oracle/libmsda_oracle.so (the C restatement in
msda_oracle.c).  Importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; vnext_amd/ never imports it.

It also carries `msda_numpy_small`, a second, independent pure-numpy statement
of the same arithmetic (loops, small cases only) that the tests use to
cross-check the C code, so the checker itself has two legs besides the golden
vectors generated from the reference (oracle/make_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.msda_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().msda_oracle_max_threads())


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attn):
    dt = np.float64 if value.dtype == np.float64 else np.float32
    value = np.ascontiguousarray(value, dtype=dt)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    assert attn.shape == (B, Lq, M, L, P)
    assert shapes.shape == (L, 2) and lsi.shape == (L,)
    return dt, value, shapes, lsi, loc, attn, (B, S, M, D, L, Lq, P)


def msda_forward(value, shapes, lsi, loc, attn, nthreads: int = 1) -> np.ndarray:
    """[B,S,M,D],[L,2],[L],[B,Lq,M,L,P,2],[B,Lq,M,L,P] -> [B,Lq,M*D]"""
    dt, value, shapes, lsi, loc, attn, dims = _prep(value, shapes, lsi, loc, attn)
    B, S, M, D, L, Lq, P = dims
    out = np.empty((B, Lq, M * D), dtype=dt)
    fn = lib().msda_oracle_fwd_f64 if dt == np.float64 else lib().msda_oracle_fwd_f32
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attn),
       *(ctypes.c_int(x) for x in dims), _p(out), ctypes.c_int(nthreads))
    return out


def msda_backward(value, shapes, lsi, loc, attn, grad_out, nthreads: int = 1):
    """-> (grad_value, grad_loc, grad_attn) with the input shapes."""
    dt, value, shapes, lsi, loc, attn, dims = _prep(value, shapes, lsi, loc, attn)
    grad_out = np.ascontiguousarray(grad_out, dtype=dt)
    B, S, M, D, L, Lq, P = dims
    assert grad_out.size == B * Lq * M * D
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = lib().msda_oracle_bwd_f64 if dt == np.float64 else lib().msda_oracle_bwd_f32
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attn), _p(grad_out),
       *(ctypes.c_int(x) for x in dims), _p(gv), _p(gl), _p(ga), ctypes.c_int(nthreads))
    return gv, gl, ga


def level_start_index(shapes) -> np.ndarray:
    """[0, H0*W0, H0*W0+H1*W1, ...]  (reference test.py:24)"""
    shapes = np.asarray(shapes, dtype=np.int64)
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def msda_numpy_small(value, shapes, lsi, loc, attn) -> np.ndarray:
    """Independent float64 loop statement of the forward (small inputs only).

    Written from the op's definition -- bilinear interpolation of a zero-padded
    map sampled at pixel coordinates (x*W-0.5, y*H-0.5), i.e. grid_sample with
    align_corners=False (reference ms_deform_attn_func.py:48,56-57) -- rather
    than from the CUDA corner rules, so an error in one statement does not
    repeat in the other.
    """
    value = np.asarray(value, dtype=np.float64)
    loc = np.asarray(loc, dtype=np.float64)
    attn = np.asarray(attn, dtype=np.float64)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((B, Lq, M, D))
    for l in range(L):
        H, W = int(shapes[l][0]), int(shapes[l][1])
        fmap = value[:, int(lsi[l]):int(lsi[l]) + H * W].reshape(B, H, W, M, D)
        padded = np.zeros((B, H + 2, W + 2, M, D))
        padded[:, 1:-1, 1:-1] = fmap
        for b in range(B):
            for q in range(Lq):
                for m in range(M):
                    for k in range(P):
                        x = loc[b, q, m, l, k, 0] * W - 0.5
                        y = loc[b, q, m, l, k, 1] * H - 0.5
                        if not (-1 < x < W and -1 < y < H):
                            continue
                        x0, y0 = int(np.floor(x)), int(np.floor(y))
                        fx, fy = x - x0, y - y0
                        px, py = x0 + 1, y0 + 1  # index into the padded map
                        tap = ((1 - fy) * (1 - fx) * padded[b, py, px, m]
                               + (1 - fy) * fx * padded[b, py, px + 1, m]
                               + fy * (1 - fx) * padded[b, py + 1, px, m]
                               + fy * fx * padded[b, py + 1, px + 1, m])
                        out[b, q, m] += attn[b, q, m, l, k] * tap
    return out.reshape(B, Lq, M * D)
