"""A transformer layer's GEMM weights in the autocast dtype by ONE multi-tensor launch (round 6, BASELINE config 3 is bf16).

`torch.autocast` casts every fp32 weight (and bias) at the GEMM that consumes it and autograd casts every weight gradient back:
per Linear two small launches forward and two backward -- 474 `fp32 -> bf16` and 345 `bf16 -> fp32` copies in an IDOL bf16 step
of 3 350 launches, whose 33 ms of kernels take 50 ms because every launch also costs the command processor a few microseconds.
A deformable-transformer layer owns 6 (encoder) to 12 (SeqFormer decoder) Linears; this module casts all of a layer's GEMM
weights in one `torch._foreach_copy_` when the layer is entered, swaps them in for the duration of the layer's forward, and
returns their gradients to the fp32 parameters through one multi-tensor cast when the layer's backward is done:

* numerically what autocast does (the same fp32 -> bf16 rounding of the same weights, the same bf16 weight gradients cast back);
* per LAYER, not per model: a layer's weight gradients reach the fp32 parameters -- and DDP's bucket hooks -- when that layer's
  backward has run, so the all-reduce still overlaps the rest of the backward (a whole-model shadow copy would deliver every
  gradient at the end);
* only tensors that reach a library GEMM through `F.linear` under autocast: the weights of the layer's `nn.Linear`s, and the biases
  of the two plain projections of a deformable-attention module.  Every other bias is read as fp32 by a fused pass of this
  package (add_norm.hip's r_bias, ffn_act.hip), `in_proj_weight` of the self-attention by the fp32 kernel of self_attn.hip,
  LayerNorm parameters by add_norm.hip: they stay what they are.

`install(layer)` registers the two hooks; the transformer layers call it in `__init__`.  VNX_SHADOW_WEIGHTS=0 switches it off.
Outside autocast, on the CPU, or when a parameter is not fp32, the hooks do nothing.  The swap lives for the duration of ONE
layer's forward on the calling thread (restored by an `always_call` hook, also when the forward raises): code that walks the
module tree from another thread in that window -- a checkpoint writer calling `state_dict()` mid-forward -- would see the 16-bit
copies of that layer; Detectron2's trainers checkpoint between iterations.

Measured (round 6, one MI355X, same process, alternating): launches per bf16 step 3 659 -> 3 419 (SeqFormer), 3 346 -> 3 166
(IDOL); step time unchanged within noise (51.4 / 50.8 ms IDOL, 61.8 / 62.1 SeqFormer): the casts were 4-us kernels behind a
host-bound step.  Kept for the launch count.
"""
from __future__ import annotations

import os

import torch
from torch import nn

ENABLED = os.environ.get("VNX_SHADOW_WEIGHTS", "1") != "0"
# biases that reach a GEMM epilogue through a plain `linear(x)` call (ops/modules/ms_deform_attn.py:_raw_offsets_and_logits)
_PLAIN_BIAS_OWNERS = ("sampling_offsets", "attention_weights")


class _CastGroup(torch.autograd.Function):
    """params -> their copies in `dtype`, one multi-tensor launch; backward: the copies' gradients -> fp32, one launch."""

    @staticmethod
    def forward(ctx, dtype, *params):
        ctx.set_materialize_grads(False)
        out = [torch.empty_like(p, dtype=dtype) for p in params]
        torch._foreach_copy_(out, list(params))
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        live = [i for i, g in enumerate(grads) if g is not None]
        res = [None] * len(grads)
        if live:
            out = [torch.empty_like(grads[i], dtype=torch.float32) for i in live]
            torch._foreach_copy_(out, [grads[i] for i in live])
            for i, g in zip(live, out):
                res[i] = g
        return (None, *res)


def _entries(layer):
    found = []
    for mod_name, mod in layer.named_modules():
        if isinstance(mod, nn.Linear):
            found.append((mod, "weight"))
            if mod.bias is not None and mod_name.rsplit(".", 1)[-1] in _PLAIN_BIAS_OWNERS:
                found.append((mod, "bias"))
    return found


def _enter(layer, args):
    if not ENABLED or not torch.is_autocast_enabled():
        return
    entries = _entries(layer)
    params = [m._parameters[n] for m, n in entries]
    if not params or not all(isinstance(p, nn.Parameter) and p.is_cuda and p.dtype == torch.float32 for p in params):
        return          # (also: a layer entered again from inside itself, already swapped)
    adt = torch.get_autocast_dtype("cuda")
    if adt not in (torch.bfloat16, torch.float16):
        return
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        copies = _CastGroup.apply(adt, *params)
    else:
        with torch.no_grad():
            copies = [torch.empty_like(p, dtype=adt) for p in params]
            torch._foreach_copy_(copies, params)
    for (m, n), c in zip(entries, copies):
        m._parameters[n] = c
    layer._vnx_shadowed = (entries, params)


def _leave(layer, args, output):
    state = layer.__dict__.pop("_vnx_shadowed", None)
    if state is not None:
        for (m, n), p in zip(*state):
            m._parameters[n] = p


def install(layer: nn.Module) -> None:
    """Register the swap on `layer` (a transformer encoder / decoder layer)."""
    layer.register_forward_pre_hook(_enter)
    layer.register_forward_hook(_leave, always_call=True)
