"""`MSDeformAttn` modules of both projects, on the MI355X op.

Two classes, because the reference has two (same name, different forward):

* `MSDeformAttnIDOL`  <- projects/IDOL/idol/models/ops/modules/ms_deform_attn.py:30-116
  `forward(query, reference_points, input_flatten, spatial_shapes, level_start_index,
  padding_mask) -> (output, sampling_locations, attention_weights)`
* `MSDeformAttnSeqFormer` <- projects/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py:32-217
  `mode='encode'|'decode'`, clip tensors `[N, T, ...]`, extra `output_proj_box`.

Parameter names, shapes and initialisation are the reference's (checkpoints load
unchanged).  What differs is the mapping onto the machine: the
reference loops over the T frames of a clip in Python and launches the op once per frame
(`:107-120, :153-167, :198-211`); here the frame axis is folded into the op's batch axis, so
a layer is ONE launch whatever T is, and no per-frame `contiguous()` / `cat` copies are made.
The arithmetic that builds `sampling_locations` keeps the reference's expression order so
fp32 results match bit for bit up to the op.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from ... import msda_ext
from ..fused_ffn import autocast_once, linear_masked
from ..functions import MSDeformAttnFunction, MSDeformAttnFusedFunction, check_flattened_length


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class _MSDeformAttnBase(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-two head dimension (32 for the tuned kernels) is faster.")
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        # set by a transformer layer that adds the output projections' biases itself, in the pass that follows
        # (`add_dropout_norm(..., r_bias=...)`: the bias gradient then falls out of the LayerNorm backward instead of a
        # reduction launch per projection): the module then returns `output_proj` / `output_proj_box` WITHOUT bias
        self.defer_output_bias = False

    def _project_output(self, x, proj):
        return F.linear(x, proj.weight) if self.defer_output_bias else proj(x)

    def _reset_parameters(self):
        # reference :62-75 (IDOL) / :65-80 (SeqFormer): zero offset weights, bias = the head's unit
        # direction (max-norm normalised) times (k+1); uniform attention; xavier projections
        constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    # -- fused prologue -----------------------------------------------------------------------
    # When a caller does not read the sampling_locations / attention_weights the reference module
    # returns (`return_samples = False`; the transformers in vnext_amd/models set it), the softmax and
    # the location arithmetic run inside the sampling kernels (MSDeformAttnFusedFunction) and the two
    # tensors are never materialised; the module then returns None in their place.
    return_samples = True
    fused_prologue = True

    def _raw_offsets_and_logits(self, query):
        lead = query.shape[:-1]
        query = autocast_once(query)      # (feeds two Linears: one cast under autocast instead of two)
        offsets = self.sampling_offsets(query).view(*lead, self.n_heads, self.n_levels, self.n_points, 2)
        logits = self.attention_weights(query).view(*lead, self.n_heads, self.n_levels * self.n_points)
        return offsets, logits

    def _try_fused(self, value, offsets, logits, reference, spatial_shapes, level_start_index):
        """value [B,S,M,D], offsets [B,Lq,M,L,P,2], logits [B,Lq,M,L*P], reference [B or B/T,Lq,L,2|4]
        -> sampled [B, Lq, C] or None when the fused kernels do not take this case."""
        if self.return_samples or not self.fused_prologue:
            return None
        if reference.dtype != offsets.dtype and not (offsets.dtype == torch.bfloat16 and value.dtype == torch.bfloat16
                                                     and reference.dtype == torch.float32):
            # autocast: the Linears emit 16-bit tensors, the reference points stay fp32 -- rounding positions to 8 mantissa bits
            # would cost more than half a pixel at 720p.  bf16: the kernels read the offsets / logits as they are beside fp32
            # reference points (VNX_MSDA_REF_F32, round 6: until then both were promoted here, two casts forward and two
            # backward per call, 78 MB each on an encoder call); other combinations are promoted.
            offsets, logits, reference = offsets.float(), logits.float(), reference.float()
        reference = reference.contiguous()
        if not msda_ext.fused_supported(value, spatial_shapes, offsets, logits, reference, level_start_index):
            return None
        return MSDeformAttnFusedFunction.apply(value.contiguous(), spatial_shapes, level_start_index,
                                               offsets.contiguous(), logits.contiguous(), reference)

    # -- shared pieces ------------------------------------------------------------------------
    def _project_value(self, input_flatten, input_padding_mask):
        # value = self.value_proj(input_flatten); value.masked_fill(input_padding_mask[..., None], 0)  (SeqFormer :94-96):
        # one GEMM + one in-place pass on the GPU (vnext_amd/ops/fused_ffn.py), the expression itself elsewhere
        return linear_masked(input_flatten, self.value_proj, input_padding_mask)

    def _offsets_and_weights(self, query):
        lead = query.shape[:-1]
        query = autocast_once(query)
        offsets = self.sampling_offsets(query).view(*lead, self.n_heads, self.n_levels, self.n_points, 2)
        weights = self.attention_weights(query).view(*lead, self.n_heads, self.n_levels * self.n_points)
        weights = F.softmax(weights, -1).view(*lead, self.n_heads, self.n_levels, self.n_points)
        return offsets, weights

    def _locations(self, reference_points, offsets, spatial_shapes):
        """reference_points [..., Lq, L, 2|4] and offsets [..., Lq, M, L, P, 2] broadcast over
        their leading axes (the reference's expressions, IDOL :102-108)."""
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            return reference_points[..., :, None, :, None, :] + offsets / offset_normalizer[None, None, None, :, None, :]
        if reference_points.shape[-1] == 4:
            return reference_points[..., :, None, :, None, :2] \
                + offsets / self.n_points * reference_points[..., :, None, :, None, 2:] * 0.5
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))


class MSDeformAttnIDOL(_MSDeformAttnBase):
    """IDOL's module (no frame axis)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__(d_model, n_levels, n_heads, n_points)
        self._reset_parameters()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        check_flattened_length(input_spatial_shapes, Len_in)
        value = self._project_value(input_flatten, input_padding_mask)
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        if not self.return_samples and self.fused_prologue:
            offsets, logits = self._raw_offsets_and_logits(query)
            output = self._try_fused(value, offsets, logits, reference_points, input_spatial_shapes,
                                     input_level_start_index)
            if output is not None:
                return self._project_output(output, self.output_proj), None, None
        sampling_offsets, attention_weights = self._offsets_and_weights(query)
        sampling_locations = self._locations(reference_points, sampling_offsets, input_spatial_shapes)
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                            sampling_locations.contiguous(), attention_weights,
                                            self.im2col_step)
        return self._project_output(output, self.output_proj), sampling_locations, attention_weights


class MSDeformAttnSeqFormer(_MSDeformAttnBase):
    """SeqFormer's module: clip tensors [N, T, ...]; T is folded into the op batch."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, mode='encode'):
        super().__init__(d_model, n_levels, n_heads, n_points)
        self.mode = mode
        # present in the reference whatever the mode, and NOT re-initialised by
        # _reset_parameters (SeqFormer :63,77-80) -- kept so state dicts round-trip
        self.output_proj_box = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def forward(self, query, query_box, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        if self.mode == 'encode':
            return self.encode_forward(query, reference_points, input_flatten, input_spatial_shapes,
                                       input_level_start_index, input_padding_mask)
        elif self.mode == 'decode':
            return self.decode_forward(query, query_box, reference_points, input_flatten,
                                       input_spatial_shapes, input_level_start_index, input_padding_mask)

    def _apply_folded(self, value, locations, weights, shapes, level_start_index):
        """value [N,T,S,M,D], locations [N,T,Lq,M,L,P,2], weights [N,T,Lq,M,L,P] -> [N,T,Lq,C]"""
        N, T = value.shape[:2]
        out = MSDeformAttnFunction.apply(
            value.reshape(N * T, *value.shape[2:]), shapes, level_start_index,
            locations.reshape(N * T, *locations.shape[2:]).contiguous(),
            weights.reshape(N * T, *weights.shape[2:]).contiguous(), self.im2col_step)
        return out.view(N, T, out.shape[1], out.shape[2])

    def encode_forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                       input_level_start_index, input_padding_mask=None):
        N, nf, Len_q, _ = query.shape
        N, nf, Len_in, _ = input_flatten.shape
        check_flattened_length(input_spatial_shapes, Len_in)
        value = self._project_value(input_flatten, input_padding_mask)
        value = value.view(N, nf, Len_in, self.n_heads, self.d_model // self.n_heads)
        if reference_points.shape[-1] != 2:
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'.format(
                reference_points.shape[-1]))
        if self.fused_prologue:      # encode_forward never returned locations / weights
            offsets, logits = self._raw_offsets_and_logits(query)
            keep, self.return_samples = self.return_samples, False
            sampled = self._try_fused(value.reshape(N * nf, *value.shape[2:]), offsets.reshape(N * nf, *offsets.shape[2:]),
                                      logits.reshape(N * nf, *logits.shape[2:]), reference_points,
                                      input_spatial_shapes, input_level_start_index)
            self.return_samples = keep
            if sampled is not None:
                return self._project_output(sampled.view(N, nf, Len_q, -1), self.output_proj)
        sampling_offsets, attention_weights = self._offsets_and_weights(query)  # [N,nf,Lq,M,L,P(,2)]
        # the encoder's reference points are shared by the frames: [N, Lq, L, 2] (SeqFormer :107-112)
        locations = self._locations(reference_points[:, None], sampling_offsets, input_spatial_shapes)
        sampled = self._apply_folded(value, locations, attention_weights, input_spatial_shapes,
                                     input_level_start_index)
        return self._project_output(sampled, self.output_proj)

    def decode_forward(self, query, query_box, reference_points, input_flatten, input_spatial_shapes,
                       input_level_start_index, input_padding_mask=None):
        N, nf, Len_in, _ = input_flatten.shape
        check_flattened_length(input_spatial_shapes, Len_in)
        value = self._project_value(input_flatten, input_padding_mask)
        value = value.view(N, nf, Len_in, self.n_heads, self.d_model // self.n_heads)
        if query_box.dim() == 4 and not self.return_samples and self.fused_prologue:
            offsets, logits = self._raw_offsets_and_logits(query_box)          # per-frame box queries
            sampled = self._try_fused(value.reshape(N * nf, *value.shape[2:]), offsets.reshape(N * nf, *offsets.shape[2:]),
                                      logits.reshape(N * nf, *logits.shape[2:]),
                                      reference_points.reshape(N * nf, *reference_points.shape[2:]),
                                      input_spatial_shapes, input_level_start_index)
            if sampled is not None:
                sampled = sampled.view(N, nf, sampled.shape[1], -1)
                return self._project_output(sampled, self.output_proj), self._project_output(sampled, self.output_proj_box), None, None
        sampling_offsets, attention_weights = self._offsets_and_weights(query_box)
        if query_box.dim() == 3:
            # first decoder layer: one set of offsets / weights per query, shared by the frames
            # (SeqFormer :128-171); reference_points are per frame [N, nf, Lq, L, 2|4]
            locations = self._locations(reference_points, sampling_offsets[:, None], input_spatial_shapes)
            weights = attention_weights[:, None].expand(N, nf, *attention_weights.shape[1:])
        else:
            assert query_box.dim() == 4  # [N, nf, Lq, C]: per-frame box queries (SeqFormer :172-217)
            locations = self._locations(reference_points, sampling_offsets, input_spatial_shapes)
            weights = attention_weights
        sampled = self._apply_folded(value, locations, weights, input_spatial_shapes, input_level_start_index)
        output = self._project_output(sampled, self.output_proj)
        output_box = self._project_output(sampled, self.output_proj_box)
        # the reference returns the LAST frame's locations (the loop variable, :171,217)
        return output, output_box, locations[:, -1], attention_weights
