"""y = LayerNorm(x + dropout(r)) in one HIP pass (vnext_amd/csrc/add_norm.hip).

The reference closes every sub-layer of its deformable transformer with
`x = x + self.dropoutN(x2); x = self.normN(x)` (projects/SeqFormer/seqformer/models/deformable_transformer.py:201-236,
286-385; IDOL's copy is identical): dropout, add and layer_norm are three launches forward and four backward, 30 times
per training step.  `add_dropout_norm(x, r, dropout_module, norm_module)` is that expression; on the GPU, for fp32 rows
of 256 channels (both models' hidden size), it is one launch forward and two backward.  Everywhere else (CPU, autocast,
other widths) it IS the reference expression, evaluated by torch.

Dropout: the kernel keeps element e iff hash(seed, e) >= p * 2^32 and the backward recomputes the mask from the same
seed, so no mask is stored.  The seed of a call is `torch.initial_seed()` mixed with a per-process call counter: runs
are repeatable after `torch.manual_seed`, masks differ from call to call, but they are not torch's Philox stream (a
dropout mask has no reference value to be equal to).

hipGraph capture: a host integer is baked into a captured graph, so replays of a captured TRAINING step
would drop the same elements every step, where the `nn.Dropout` this replaces is graph-safe (its Philox offset advances
per replay).  Under capture with p > 0 the kernels therefore also read a per-device STEP SEED from device memory
(`seed_device` of the C ABI): `step_scope(device)` -- entered by the owner of the captured callable at the top of its
forward, `_TrainTrunk.forward` here -- bumps that word with a captured `add_`, once per replay; the baked host seeds
only tell the call sites apart.  What the kernels read is not the bumped word itself but a SNAPSHOT of it taken when the
scope is entered (one captured 8-byte copy per scope entry): the backward of a site reads the snapshot its forward
read, so a second `step_scope` entry between a forward and its backward -- two trunk forwards before one backward,
`(loss1 + loss2).backward()`, a second graphed shape key -- cannot change the mask under it.  The step-seed
word is created eagerly, never inside a capture (its fill would be captured and every replay would reset the seed).  A
capture that never entered a `step_scope` falls back to the reference expression `norm(x + dropout(r))` (three
graph-safe ATen launches).
"""
from __future__ import annotations

import torch

from .. import _lib

CHANNELS = 256
_calls = 0
_step_seed = {}          # device -> int64[1] tensor, bumped once per training step inside step_scope()
_scopes = []             # the step_scope objects entered and not yet left (innermost last)


def _device_key(device):
    device = torch.device(device)
    return (device.type, device.index if device.index is not None else torch.cuda.current_device())


def step_seed_tensor(device) -> torch.Tensor:
    device = torch.device(device)
    key = _device_key(device)
    t = _step_seed.get(key)
    if t is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(
                "vnext_amd.ops.fused_norm: the per-device step seed must exist before a hipGraph capture starts (its "
                "fill would be captured and every replay would reset the seed): run one eager warm-up step, or call "
                "fused_norm.step_seed_tensor(device) once, before capturing")
        t = _step_seed[key] = torch.full((1,), torch.initial_seed() & 0x7FFFFFFFFFFF, dtype=torch.int64, device=device)
    return t


class step_scope:
    """`with step_scope(device): ...` around the part of a training step that may be captured into a hipGraph: bumps
    the device's step seed (a captured op when capturing, so every replay bumps it again) and marks the fused
    dropout sites inside as safe to fuse under capture."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.snapshot = None     # int64[1]: the step seed as this entry left it; what the sites inside hand to the kernels

    def __enter__(self):
        if self.device.type == "cuda":
            t = step_seed_tensor(self.device)
            t.add_(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFF)
            # under capture: a captured copy into the graph's private pool, rewritten by every replay; the autograd
            # contexts of the sites inside keep it alive until their backward has run
            self.snapshot = t.clone()
        _scopes.append(self)
        return self

    def __exit__(self, *exc):
        _scopes.remove(self)
        return False


def _scope_snapshot(device):
    """The seed snapshot of the innermost open step_scope of this device, or None."""
    key = _device_key(device)
    for sc in reversed(_scopes):
        if sc.snapshot is not None and _device_key(sc.device) == key:
            return sc.snapshot
    return None


def _next_seed() -> int:
    global _calls
    _calls += 1
    # 63 bits: torch.profiler(record_shapes=True) converts Function.apply's integer arguments to int64
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _calls * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF


class _AddDropoutLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, p, eps, seed, seed_tensor, r_bias=None):
        lib = _lib.lib()
        x, r = x.contiguous(), r.contiguous()
        rows = x.numel() // CHANNELS
        y, z = torch.empty_like(x), torch.empty_like(x)
        ctx.branch_dtype = r.dtype          # fp32, or bf16 under autocast (the Linear / attention output): the kernel reads it as it is
        stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.vnx_add_dropout_layernorm_forward(
                _branch_code(r.dtype), x.data_ptr(), r.data_ptr(), r_bias.data_ptr() if r_bias is not None else None,
                gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), z.data_ptr(),
                stats.data_ptr(), rows, CHANNELS, float(p), float(eps), int(seed),
                seed_tensor.data_ptr() if seed_tensor is not None else None, _lib.current_stream(x)))
        ctx.save_for_backward(z, stats, gamma)
        ctx.p, ctx.seed, ctx.seed_tensor, ctx.has_bias = float(p), int(seed), seed_tensor, r_bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_y):
        lib = _lib.lib()
        z, stats, gamma = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        if grad_y.dtype != torch.float32:
            grad_y = grad_y.float()
        rows = z.numel() // CHANNELS
        grad_x, grad_r = torch.empty_like(z), torch.empty_like(z, dtype=ctx.branch_dtype)
        grad_gamma, grad_beta = torch.empty_like(gamma), torch.empty_like(gamma)
        grad_bias = torch.empty_like(gamma) if ctx.has_bias else None      # the folded Linear bias: column sums of grad_r
        partial = torch.empty(lib.vnx_add_dropout_layernorm_partial_bytes(), dtype=torch.uint8, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(lib.vnx_add_dropout_layernorm_backward(
                _branch_code(ctx.branch_dtype), grad_y.data_ptr(), z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), grad_x.data_ptr(),
                grad_r.data_ptr(), grad_gamma.data_ptr(), grad_beta.data_ptr(),
                grad_bias.data_ptr() if grad_bias is not None else None, partial.data_ptr(), rows, CHANNELS,
                ctx.p, ctx.seed, ctx.seed_tensor.data_ptr() if ctx.seed_tensor is not None else None,
                _lib.current_stream(z)))
        return grad_x, grad_r, grad_gamma, grad_beta, None, None, None, None, grad_bias


def _branch_code(dtype):
    return {torch.bfloat16: _lib.VNX_BF16, torch.float16: _lib.VNX_F16}.get(dtype, _lib.VNX_F32)


def fused_applies(x, r, norm) -> bool:
    """fp32 residual stream and LayerNorm; the branch fp32 or -- what a Linear emits under torch.autocast -- bf16 / f16 (Detectron2's
    AMP trainer autocasts to float16).
    The result is fp32 either way, as the eager chain's (the sum promotes, autocast runs layer_norm in fp32)."""
    return (x.is_cuda and x.dtype == torch.float32 and r.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.shape == r.shape
            and x.shape[-1] == CHANNELS and tuple(norm.normalized_shape) == (CHANNELS,)
            and norm.weight is not None and norm.bias is not None and norm.weight.dtype == torch.float32)


def dropout_site(x, dropout):
    """(p, seed_tensor, ok) for a fused dropout site on x's device: p = the module's drop probability (0 in eval mode);
    under hipGraph capture with p > 0 the seed snapshot of the enclosing step_scope, and ok = False when there is none
    (the caller must then fall back to the graph-safe nn.Dropout)."""
    p = dropout.p if dropout.training else 0.0
    if p >= 1.0:
        return p, None, False
    if p > 0.0 and torch.cuda.is_current_stream_capturing():
        snap = _scope_snapshot(x.device)
        return p, snap, snap is not None
    return p, None, True


def add_dropout_norm(x, r, dropout, norm, seed=None, r_bias=None):
    """`norm(x + dropout(r))` for an nn.Dropout and an nn.LayerNorm (see the module docstring).  r_bias: the bias of the
    Linear that produced r, when the caller ran that GEMM without it (`norm(x + dropout(r + r_bias))`): its gradient then
    comes out of this op's backward instead of a separate reduction launch."""
    def reference():
        return norm(x + dropout(r if r_bias is None else r + r_bias))
    if not fused_applies(x, r, norm) or (r_bias is not None and (r_bias.dtype != torch.float32 or r_bias.numel() != CHANNELS)):
        return reference()
    p, seed_tensor, ok = dropout_site(x, dropout)
    if not ok:
        return reference()
    return _AddDropoutLayerNorm.apply(x, r, norm.weight, norm.bias, p, norm.eps, _next_seed() if seed is None else seed,
                                      seed_tensor, r_bias.contiguous() if r_bias is not None else None)
